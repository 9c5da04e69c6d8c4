#!/bin/bash
# same-box timing of swc_convnext_block / swc_convnext_mlp for several library builds: tools/ab_convnext.sh <tag> <tag> ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for rep in 1 2; do
  for tag in "$@"; do
    echo "== $tag"; SWC_LIB=$R/simwhisper_codec_amd/libswc_$tag.so python tools/bench_convnext.py 2>/dev/null | grep -E "^fused|^block"
  done
done
