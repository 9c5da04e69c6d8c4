#!/bin/bash
# build libswc_<tag>.so with extra flags for swc_attention16.hip only: tools/build_att_variant.sh <tag> <flags...>
# (e.g. -DATT_PIPE=0, -DATT_ABL=4; select the result with SWC_LIB=<path>; the product library is never overwritten)
set -e
cd "$(dirname "$0")/../simwhisper_codec_amd"
tag=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -fno-slp-vectorize "$@" -I ../include -I csrc -c csrc/swc_attention16.hip -o /tmp/att16_$tag.o
objs=""
for o in build/*.o; do [ "$o" = "build/swc_attention16.o" ] || objs="$objs $o"; done   # every other object of the product build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libswc_$tag.so $objs /tmp/att16_$tag.o
echo built libswc_$tag.so
