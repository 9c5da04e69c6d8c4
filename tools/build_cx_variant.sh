#!/bin/bash
# build libswc_<tag>.so with extra flags for swc_convnext.hip only (timing ablations -DCX_ABL=mask, tuning constants):
#   tools/build_cx_variant.sh <tag> <flags...>      select with SWC_LIB=simwhisper_codec_amd/libswc_<tag>.so
set -e
cd "$(dirname "$0")/../simwhisper_codec_amd"
tag=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp "$@" -I ../include -I csrc -c csrc/swc_convnext.hip -o /tmp/cx_$tag.o
objs=""
for o in build/*.o; do [ "$o" = "build/swc_convnext.o" ] || objs="$objs $o"; done   # every other object of the product build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libswc_$tag.so $objs /tmp/cx_$tag.o
echo built libswc_$tag.so
