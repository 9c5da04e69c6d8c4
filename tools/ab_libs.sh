#!/bin/bash
# same-box comparison of alternative library builds (SWC_LIB): tools/ab_libs.sh tag1 tag2 ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for rep in 1 2; do
  for tag in "$@"; do
    lib=$R/simwhisper_codec_amd/libswc_$tag.so
    echo "== $tag"
    SWC_LIB=$lib python tools/bench_gemm.py bf16 2>&1 | grep "total\|pw1\|fc2  \|qkv" | awk '{printf "%s %s %s | ", $2, $(NF-3), $(NF-1)} END {print ""}'
    SWC_LIB=$lib python tools/bench_gemm.py f16s 2>&1 | grep "total\|fc1\|fc2  " | awk '{printf "f16s %s %s %s | ", $2, $(NF-3), $(NF-1)} END {print ""}'
  done
done
