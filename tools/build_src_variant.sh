#!/bin/bash
# build libswc_<tag>.so with extra flags for ONE source file (timing ablations, tuning constants); every other object comes from
# the product build under simwhisper_codec_amd/build/:
#   tools/build_src_variant.sh <tag> <file.hip> <flags...>      select with SWC_LIB=simwhisper_codec_amd/libswc_<tag>.so
set -e
cd "$(dirname "$0")/../simwhisper_codec_amd"
tag=$1; src=$2; shift 2
base=${src%.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp "$@" -I ../include -I csrc -c csrc/$src -o /tmp/${base}_$tag.o
objs=""
for o in build/*.o; do [ "$o" = "build/$base.o" ] || objs="$objs $o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libswc_$tag.so $objs /tmp/${base}_$tag.o
echo built libswc_$tag.so
