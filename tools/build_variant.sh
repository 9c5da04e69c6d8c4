#!/bin/bash
# build libswc_<tag>.so with extra flags for swc_gemm.hip only: tools/build_variant.sh <tag> <flags...>
# (always with -DSWC_TUNING: only such builds read the SWC_GEMM_* geometry overrides; select one with SWC_LIB=<path>,
# the product library libswc_hip.so is never overwritten)
set -e
cd "$(dirname "$0")/../simwhisper_codec_amd"
tag=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -mllvm -amdgpu-sched-strategy=max-ilp -DSWC_TUNING "$@" -I ../include -I csrc -c csrc/swc_gemm.hip -o /tmp/gemm_$tag.o
objs=""
for o in build/*.o; do [ "$o" = "build/swc_gemm.o" ] || objs="$objs $o"; done   # every other object of the product build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libswc_$tag.so $objs /tmp/gemm_$tag.o
echo built libswc_$tag.so
