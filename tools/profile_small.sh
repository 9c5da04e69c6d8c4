#!/bin/bash
# Kernel stats + stage times of the small-batch config (BASELINE.json configs[1], B=8 x 10 s).  usage: bash tools/profile_small.sh <tag> [precision]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-r02_small}; prec=${2:-mixed}
out=$R/gpurun_out/$tag
mkdir -p $out
cd "$R"
python3 bench.py --steps 20 --warmup 5 --batch 8 --cpu-baseline off --no-dist --precision $prec > $out/bench_$prec.json 2> $out/bench_$prec.err
tail -c 900 $out/bench_$prec.json; echo
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$prec -o s -- python3 $R/bench.py --steps 13 --warmup 3 --batch 8 --precision $prec --cpu-baseline off --no-dist --no-timer > $out/under_rocprof_$prec.json 2> $out/rocprof_$prec.err )
python3 tools/prof_summary.py $out/stats_$prec/s_kernel_stats.csv 16 40 > $out/summary_$prec.txt 2>&1
