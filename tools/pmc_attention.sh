#!/bin/bash
# PMC passes over swc_attention16 (tools/bench_attention.py): tools/pmc_attention.sh <tag>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-att}
out=$R/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $out/a -o a -- python3 $R/tools/bench_attention.py >> $out/run.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $out/b -o b -- python3 $R/tools/bench_attention.py >> $out/run.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_LEVEL_WAVES SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY --output-format csv -d $out/c -o c -- python3 $R/tools/bench_attention.py >> $out/run.log 2>&1
cd $R
python3 - "$out" <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        fam = "attn16<bf16>" if "attn16_kernel<1" in k else ("attn16<f16s>" if "attn16_kernel<2" in k else None)
        if fam:
            acc[fam][r["Counter_Name"]].append(float(r["Counter_Value"]))
for fam, d in acc.items():
    print("==", fam)
    for c, v in sorted(d.items()):
        print(f"  {c:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
PY
