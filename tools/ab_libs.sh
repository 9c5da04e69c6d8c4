#!/bin/bash
# same-box comparison of alternative swc_gemm builds (tools/build_variant.sh <tag>; SWC_LIB): per-shape TFLOP/s, two rounds
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for rep in 1 2; do
  for tag in "$@"; do
    lib=$R/simwhisper_codec_amd/libswc_$tag.so
    echo "== $tag"
    SWC_LIB=$lib python3 tools/bench_gemm.py bf16 2>&1 | grep TFLOP | awk '{printf "bf16 %s %s | ", $2, $(NF-1)} END {print ""}'
    SWC_LIB=$lib python3 tools/bench_gemm.py f16s 2>&1 | grep TFLOP | awk '{printf "f16s %s %s | ", $2, $(NF-1)} END {print ""}'
  done
done
