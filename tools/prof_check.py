#!/usr/bin/env python3
"""Consistency of an evidence set (tools/profile_round.sh): the kernel time per step summed from the rocprofv3 stats CSV
must agree with the ms_per_step the same (profiled) command printed — more kernel time than wall time means launches of
two batches overlapped in the trace (the round-2 set was polluted that way) and the per-kernel averages are not serial ones.
usage: prof_check.py gpurun_out/<tag>"""
import csv, glob, json, sys
out = sys.argv[1]
line = json.loads(open(f"{out}/bench_under_rocprof.json").readline())
steps, warm = line["steps"], line["warmup"]
f = sorted(glob.glob(f"{out}/stats/**/*kernel_stats.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
# the command runs `warm` + `steps` metric steps (independent shards only: --no-dist --no-inflight --other-configs off) and,
# before them, one utterance alone for the answer check
n = steps + warm
per_step = tot / n
ms = line["independent_shards"]["ms_per_step"]
print(f"{f}\nkernel time {tot:.1f} ms over {n} steps (+ one single-utterance call) = {per_step:.3f} ms/step; "
      f"the same run printed ms_per_step = {ms:.3f}; ratio {per_step / ms:.3f} (must be <= 1.05)")
for r in rows[:12]:
    print(f'  {r["Name"][:90]:90s} {int(r["Calls"]) / n:6.1f}/step  avg {float(r["AverageNs"]) / 1e3:8.1f} us  {float(r["TotalDurationNs"]) / 1e6 / n:7.3f} ms/step')
