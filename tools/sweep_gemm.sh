#!/bin/bash
# per-shape sweep of the tile-order band and the rows-per-tile choice (same box, one process per setting)
# needs a tuning build: tools/build_variant.sh tune
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
export SWC_LIB=${SWC_LIB:-$R/simwhisper_codec_amd/libswc_tune.so}
for kind in bf16 f16s; do
  for band in 0 1 2 4 8; do
    for mt in 0 8 6 4; do
      echo "== $kind band=$band mt=$mt"
      SWC_GEMM_BAND=$band SWC_GEMM_MT=$mt python tools/bench_gemm.py $kind 2>&1 | grep TFLOP | awk '{printf "%s %s %s %s %s us\n", $2, $4, $6, $8, $9}'
    done
  done
done
