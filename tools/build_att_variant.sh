#!/bin/bash
# build libswc_<tag>.so with extra flags for swc_attention16.hip only: tools/build_att_variant.sh <tag> <flags...>
# (e.g. -DATT_PIPE=0, -DATT_ABL=4; select the result with SWC_LIB=<path>; the product library is never overwritten)
set -e
cd "$(dirname "$0")/../simwhisper_codec_amd"
tag=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -fno-slp-vectorize "$@" -I ../include -I csrc -c csrc/swc_attention16.hip -o /tmp/att16_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libswc_$tag.so build/swc_api.o build/swc_gemm.o build/swc_attention.o /tmp/att16_$tag.o build/swc_pointwise.o build/swc_convnext.o build/swc_mlp.o build/swc_convnext64.o
echo built libswc_$tag.so
