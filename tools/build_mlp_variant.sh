#!/bin/bash
# build libswc_<tag>.so with extra flags for swc_mlp.hip only (timing ablations -DML_ABL=mask, tuning constants -DML_PF=):
#   tools/build_mlp_variant.sh <tag> <flags...>      select with SWC_LIB=simwhisper_codec_amd/libswc_<tag>.so
set -e
cd "$(dirname "$0")/../simwhisper_codec_amd"
tag=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp "$@" -I ../include -I csrc -c ${MLP_SRC:-csrc/swc_mlp.hip} -o /tmp/mlp_$tag.o
objs=""
for o in build/*.o; do [ "$o" = "build/swc_mlp.o" ] || objs="$objs $o"; done   # every other object of the product build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libswc_$tag.so $objs /tmp/mlp_$tag.o
echo built libswc_$tag.so
