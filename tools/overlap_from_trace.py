#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace csv: how much of the busy time has 1, 2, ... kernels executing at once (batches in
flight on several streams overlap their launches).  usage: overlap_from_trace.py <kernel_trace.csv> [skip_fraction]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
lo = t0 + int((t1 - t0) * float(sys.argv[2] if len(sys.argv) > 2 else 0.5))  # skip warm-up / weight packing
depth, last, hist = 0, None, {}
for t, d in ev:
    if last is not None and t > lo:
        a = max(last, lo)
        hist[depth] = hist.get(depth, 0) + (t - a)
    depth += d
    last = t
tot = sum(hist.values())
print(f"{len(rows)} kernel launches; window {tot/1e6:.1f} ms")
for k in sorted(hist):
    print(f"  {k} kernel(s) executing: {hist[k]/1e6:8.2f} ms  {100.0*hist[k]/tot:5.1f} %")
