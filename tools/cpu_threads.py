#!/usr/bin/env python3
"""CPU baseline against the host thread count ("N = physical cores; report N"): the oracle's encode+decode of
B = 8 x 10 s (BASELINE.json configs[1]'s shape) at several torch thread counts up to every core visible to this process.
bench.py uses every visible core unless this sweep shows fewer is faster.  usage: python tools/cpu_threads.py [B=8] > profiles/rNN_cpu_threads.txt"""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, yaml
from bench import bench_inputs, cpu_allotment, cpu_model
from oracle.ref_cpu import Oracle
from simwhisper_codec_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
gp = yaml.safe_load(open(os.path.join(ROOT, "config", "SimWhisperCodec.yaml")))["generator_params"]
ora = Oracle(gp, synth.synth_state_dict(gp))
wavs = bench_inputs(B, 160000)
visible, quota = cpu_allotment()
print(f"cpu: {cpu_model()}; os.cpu_count() = {os.cpu_count()}; affinity mask (sched_getaffinity) = {visible} logical CPUs; "
      f"cgroup cpu quota = {quota} CPUs; bench.py uses min(mask, quota) = {min(visible, int(quota) if quota else visible)} threads")
use = min(visible, int(quota) if quota else visible)
counts = sorted({c for c in (use // 4, use // 2, use, 2 * use) if 1 <= c <= visible})
for n in counts:
    torch.set_num_threads(n)
    ts = []
    for it in range(3):
        t0 = time.perf_counter()
        ora.decode(ora.encode(wavs)["codes_list"])
        if it:
            ts.append(time.perf_counter() - t0)
    print(f"threads {n:3d}: {B * 10 / statistics.median(ts):7.3f} audio-s/s   (passes {' '.join(f'{t:.2f}' for t in ts)} s)", flush=True)
