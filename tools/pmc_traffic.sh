#!/bin/bash
# HBM traffic of the GEMM families from two separate --pmc passes of the bench command (MI355X_MICROARCH.md, HBM
# section: FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains beside --kernel-trace).
# usage (on the GPU box, from the repo root): bash tools/pmc_traffic.sh <outdir>
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=${1:-$R/gpurun_out/pmc_traffic}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-baseline off --no-timer --no-dist --no-inflight --other-configs off > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-baseline off --no-timer --no-dist --no-inflight --other-configs off > $out/write.log 2>&1
cd $R
python3 tools/pmc_traffic.py $out > $out/traffic.json
cat $out/traffic.json
