#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE counter CSVs (tools/pmc_traffic.sh) -> HBM bytes per launch of each swc_gemm family.
Units and corrections as MI355X_MICROARCH.md prescribes: both counters are in KiB; on gfx950 FETCH_SIZE tallies the
128-byte requests of wide coalesced reads at 64 bytes, so reads are doubled; WRITE_SIZE is exact."""
import collections
import csv
import glob
import json
import sys

root = sys.argv[1]
FAM = {"gemm_kernel<0": "gemm_f32", "gemm_kernel<1": "gemm_bf16", "gemm_kernel<2": "gemm_f16s", "gemm_kernel<3": "gemm_fp8",
       "convnext_mlp_kernel": "convnext_bf16", "mlp_block_kernel": "mlp_bf16", "dwconv7_ln_kernel": "dwconv7_ln", "attn16_kernel<1>": "attention_bf16",
       "attn16_kernel<2>": "attention_f16s", "layernorm_kernel": "layernorm", "snake_aa_kernel": "snake_aa",
       "istft_ola_kernel": "istft_ola", "mel_frames_kernel": "mel_frames"}


def collect(sub, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for k, fam in FAM.items():
                if k in r["Kernel_Name"]:
                    acc[fam].append(float(r["Counter_Value"]))
    return acc


rd, wr = collect("fetch", "FETCH_SIZE"), collect("write", "WRITE_SIZE")
out = {}
for fam in sorted(set(rd) | set(wr)):
    r = sum(rd[fam]) / max(1, len(rd[fam])) * 1024 * 2
    w = sum(wr[fam]) / max(1, len(wr[fam])) * 1024
    out[fam] = {"hbm_read_bytes_per_launch": r, "hbm_write_bytes_per_launch": w, "hbm_bytes_per_launch": r + w,
                "launches_sampled": len(rd[fam]),
                "note": "FETCH_SIZE x 1024 x 2 (gfx950 half-count correction for wide coalesced reads) + WRITE_SIZE x 1024; "
                        "separate --pmc passes of `bench.py --steps 2 --warmup 1` (tools/pmc_traffic.sh)"}
import os, subprocess
try:
    out["_commit"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                    text=True).stdout.strip() or os.environ.get("SWC_COMMIT")
except Exception:
    out["_commit"] = os.environ.get("SWC_COMMIT")
out["_profile"] = os.environ.get("SWC_PROFILE_TAG", os.path.basename(os.path.normpath(root)))
print(json.dumps(out, indent=1))
