#!/bin/bash
# same-box timing of swc_proj_ln for several library builds: tools/ab_projln.sh <tag> <tag> ...   (tag "hip" = the product)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for rep in 1 2; do
  for tag in "$@"; do
    echo "== $tag"; SWC_LIB=$R/simwhisper_codec_amd/libswc_$tag.so python tools/bench_projln.py 2>/dev/null | grep -E "proj_ln|gemm "
  done
done
