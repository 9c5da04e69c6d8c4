#!/bin/bash
# One evidence pass on the GPU box (from the repo root): rocprofv3 kernel stats of the bench command, PMC HBM traffic,
# MFMA utilisation, per-stage device times, HBM micro-bench.  usage: bash tools/profile_round.sh <tag> [commit]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-r04}
export SWC_COMMIT=${2:-unknown} SWC_PROFILE_TAG=$tag
out=$R/gpurun_out/$tag
mkdir -p $out
cd "$R"
python3 bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
echo "bench done"; tail -c 600 $out/bench.json
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 $R/bench.py --steps 13 --warmup 3 --cpu-baseline off --no-dist --no-timer --no-inflight --other-configs off > $out/bench_under_rocprof.json 2> $out/rocprof.err )
echo "rocprof stats done"
python3 tools/prof_check.py $out > $out/kernel_time_vs_step.txt 2>&1; cat $out/kernel_time_vs_step.txt
bash tools/pmc_traffic.sh $out/pmc_traffic > $out/pmc_traffic.log 2>&1
echo "pmc traffic done"
bash tools/pmc_mfma.sh $out/pmc_mfma > $out/pmc_mfma.log 2>&1
echo "pmc mfma done"
SWC_TRACE=time python3 tools/stage_times.py > $out/stage_times.txt 2>&1
echo "stage times done"; cat $out/stage_times.txt
python3 tools/bench_pointwise.py > $out/hbm_pointwise.txt 2> $out/hbm_pointwise.err
echo "pointwise done"
# presets that the bench line's own `other_configs` (configs[1], [2], [4], fp32) does not cover, on the same box (one line each)
python3 -c "
import json
d=json.loads(open('$out/bench.json').readline())
for k, v in (d.get('other_configs') or {}).items():
    r=(v.get('roofline') or {})
    print(json.dumps({'config': k, 'value': v.get('value'), 'ms_per_step': v.get('ms_per_step'), 'dominant': r.get('kernel'), 'achieved': r.get('achieved'), 'frac': r.get('frac'), 'other': {a: b['TFLOP/s'] for a, b in (r.get('other') or {}).items()}, 'error': v.get('error')}))
" > $out/other_configs.jsonl
for spec in "--precision bf16" "--precision fp8_fc1" "--precision f16s"; do
  python3 bench.py --steps 6 --warmup 2 --cpu-baseline off --no-dist --no-inflight --other-configs off $spec 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d.get('roofline',{})
print(json.dumps({'args': '$spec', 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'dominant': r.get('kernel'), 'achieved': r.get('achieved'), 'frac': r.get('frac'), 'other': {k: v['TFLOP/s'] for k, v in r.get('other', {}).items()}}))" >> $out/other_configs.jsonl
done
cat $out/other_configs.jsonl
SWC_TRACE=time python3 tools/stage_times.py 32 30 > $out/stage_times_32x30.txt 2>&1; cat $out/stage_times_32x30.txt
