#!/bin/bash
# One evidence pass on the GPU box (from the repo root): rocprofv3 kernel stats of the bench command, PMC HBM traffic,
# MFMA utilisation, per-stage device times, HBM micro-bench.  usage: bash tools/profile_round.sh <tag> [commit]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-r02}
export SWC_COMMIT=${2:-unknown} SWC_PROFILE_TAG=$tag
out=$R/gpurun_out/$tag
mkdir -p $out
cd "$R"
python3 bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
echo "bench done"; tail -c 600 $out/bench.json
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 $R/bench.py --steps 13 --warmup 3 --cpu-baseline off --no-dist --no-timer > $out/bench_under_rocprof.json 2> $out/rocprof.err )
echo "rocprof stats done"
bash tools/pmc_traffic.sh $out/pmc_traffic > $out/pmc_traffic.log 2>&1
echo "pmc traffic done"
bash tools/pmc_mfma.sh $out/pmc_mfma > $out/pmc_mfma.log 2>&1
echo "pmc mfma done"
SWC_TRACE=time python3 tools/stage_times.py > $out/stage_times.txt 2>&1
echo "stage times done"; cat $out/stage_times.txt
python3 tools/bench_pointwise.py > $out/hbm_pointwise.txt 2> $out/hbm_pointwise.err
echo "pointwise done"
