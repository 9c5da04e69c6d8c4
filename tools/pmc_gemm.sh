#!/bin/bash
# PMC passes over one GEMM shape: tools/pmc_gemm.sh <tag> M N K kind  (env SWC_GEMM_* pass through)
set -e
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $R/gpurun_out/pmc_$tag/a -o a -- python3 $R/tools/one_gemm.py "$@" >> $R/gpurun_out/pmc_$tag/run.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL --output-format csv -d $R/gpurun_out/pmc_$tag/b -o b -- python3 $R/tools/one_gemm.py "$@" >> $R/gpurun_out/pmc_$tag/run.log 2>&1
echo "pmc $tag done"
