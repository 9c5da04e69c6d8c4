#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats csv: top kernels and per-step time."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total device time {tot/1e6:.2f} ms over {steps:g} steps = {tot/1e6/steps:.2f} ms/step")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 22]:
    print(f'{r["Name"][:84]:84s} calls={int(r["Calls"])/steps:7.1f}/step  {float(r["TotalDurationNs"])/1e6/steps:8.3f} ms/step  avg={float(r["AverageNs"])/1e3:8.1f} us  {float(r["Percentage"]):5.1f}%')
