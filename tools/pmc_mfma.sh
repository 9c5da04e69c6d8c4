#!/bin/bash
# MFMA pipe utilisation per kernel family from one --pmc pass of the bench command (no trace domains beside
# --kernel-trace).  usage (GPU box, repo root): bash tools/pmc_mfma.sh <outdir>
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=${1:-$R/gpurun_out/pmc_mfma}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d $out/p -o m -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-baseline off --no-timer --no-dist --no-inflight --other-configs off > $out/run.log 2>&1
cd $R
python3 tools/pmc_mfma.py $out > $out/mfma_util.json
cat $out/mfma_util.json
