#!/bin/bash
# B = 8 x 10 s (configs[1], bf16): the small-grid tile choice of swc_gemm on a tuning build (tools/build_variant.sh tune):
# SWC_GEMM_SMALL=1 (shipped: 64 x 128 tiles when the 128 x 128 tiling leaves CU slots empty) against 0 (128 x 128 always)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"
for rep in 1 2; do
  for small in 1 T256; do
    echo "== SWC_GEMM_SMALL=$small"
    SWC_GEMM_SMALL=$([ "$small" = T256 ] && echo 1 || echo $small) SWC_GEMM_TILE=$([ "$small" = T256 ] && echo 256 || echo 0) SWC_LIB=$R/simwhisper_codec_amd/libswc_tune.so python bench.py --steps 20 --warmup 5 --batch 8 --precision bf16 --cpu-baseline off --no-dist --no-inflight --other-configs off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
  done
done
