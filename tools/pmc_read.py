#!/usr/bin/env python3
"""Average PMC counters per dispatch of kernels matching a substring: pmc_read.py <dir> <substr>"""
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k][1:] if len(acc[k]) > 1 else acc[k]
    print(f"{k:32s} {sum(v)/len(v):16.0f}  (n={len(v)})")
