#!/usr/bin/env python3
"""SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs) per kernel family (tools/pmc_mfma.sh).
SQ_VALU_MFMA_BUSY_CYCLES sums, over the chip's SIMDs, the cycles their MFMA pipe was busy (16 per 16x16x32 16-bit
MFMA: checked against the instruction count of a 32768 x 4096 x 4096 GEMM); GRBM_GUI_ACTIVE is reported summed over
the 8 XCDs (its value / 8 over the dispatch's duration gives the 2.0-2.1 GHz the chip holds: printed as
implied_clock_ghz); MI355X has 256 CUs x 4 SIMDs."""
import collections
import csv
import glob
import json
import sys

root = sys.argv[1]
FAM = [("gemm_kernel<0", "gemm_f32"), ("gemm_kernel<1", "gemm_bf16"), ("gemm_kernel<2", "gemm_f16s"),
       ("attn16_kernel<1>", "attention_bf16"), ("attn16_kernel<2>", "attention_f16s"),
       ("convnext_mlp_kernel", "convnext_bf16"), ("mlp_block_kernel", "mlp_bf16")]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for f in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k, fam in FAM:
            if k in r["Kernel_Name"]:
                acc[fam][r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    n[fam] += 1
                    acc[fam]["duration_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
out = {}
for fam, c in acc.items():
    busy, active = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    out[fam] = {"mfma_busy_cycles": busy, "gui_active_cycles_per_xcd": active, "launches": n[fam],
                "implied_clock_ghz": round(active / c["duration_ns"], 3) if c.get("duration_ns") else None,
                "mfma_pipe_utilisation": round(busy / (active * 1024), 4) if active else None}
    # busy cycles: 16 per 16x16x32 and 32 per 32x32x16 16-bit MFMA, summed over SIMDs
print(json.dumps(out, indent=1))
