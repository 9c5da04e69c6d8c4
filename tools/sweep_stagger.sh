#!/bin/bash
# same-box sweep of the start stagger of swc_gemm (tuning build libswc_tun2.so, SWC_GEMM_STAGGER = delay unit in cycles):
# per-shape TFLOP/s, then the whole step.  usage: bash tools/sweep_stagger.sh "0 1000 2000 4000" [lib tag]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
tag=${2:-tun2}
lib=$R/simwhisper_codec_amd/libswc_$tag.so
for rep in 1 2; do
  for st in $1; do
    echo "== stagger $st"
    SWC_LIB=$lib SWC_GEMM_STAGGER=$st python3 tools/bench_gemm.py f16s 2>&1 | grep TFLOP | awk '{printf "f16s %s %s | ", $2, $(NF-1)} END {print ""}'
    SWC_LIB=$lib SWC_GEMM_STAGGER=$st python3 tools/bench_gemm.py bf16 2>&1 | grep TFLOP | awk '{printf "bf16 %s %s | ", $2, $(NF-1)} END {print ""}'
  done
done
for rep in 1 2; do
  for st in $1; do
    echo "== step, stagger $st"
    SWC_LIB=$lib SWC_GEMM_STAGGER=$st python3 bench.py --steps 10 --warmup 3 --cpu-baseline off --no-dist --no-inflight --other-configs off 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d.get('roofline',{})
print(d['value'], d['ms_per_step'], r.get('kernel'), r.get('achieved'), {k:v['TFLOP/s'] for k,v in r.get('other',{}).items()})"
  done
done
