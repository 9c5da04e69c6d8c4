#!/bin/bash
# tile-order band A/B; needs a tuning build: tools/build_variant.sh tune && SWC_LIB=simwhisper_codec_amd/libswc_tune.so
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
export SWC_LIB=${SWC_LIB:-$R/simwhisper_codec_amd/libswc_tune.so}
for rep in 1 2 3; do
  for band in 1 4; do
    echo "== band=$band"
    SWC_GEMM_BAND=$band python tools/bench_gemm.py bf16 2>&1 | grep "out_proj\|fc2\|pwconv2\|head\|idft" | awk '{printf "%s %s us\n", $2, $9}'
    SWC_GEMM_BAND=$band python tools/bench_gemm.py f16s 2>&1 | grep "out_proj\|fc2" | awk '{printf "f16s %s %s us\n", $2, $9}'
  done
done
