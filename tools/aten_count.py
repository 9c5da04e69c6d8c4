#!/usr/bin/env python3
"""ATen / copy kernels that one steady-state encode+decode step still launches beside the swc_* kernels (torch profiler)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, yaml
from torch.profiler import profile, ProfilerActivity
from simwhisper_codec_amd import synth
from simwhisper_codec_amd.codec import AudioCodec
from bench import bench_inputs
gp = yaml.safe_load(open(os.path.join(ROOT, "config", "SimWhisperCodec.yaml")))["generator_params"]
m = AudioCodec(gp, precision="mixed"); m.load_state_dict(synth.synth_state_dict(gp)); m = m.to("cuda:0").eval()
wavs = [w.cuda() for w in bench_inputs(32, 160000)]
for _ in range(3): m.decode(m.encode(wavs)["codes_list"])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    m.decode(m.encode(wavs)["codes_list"]); torch.cuda.synchronize()
rows = []
for e in prof.key_averages():
    if e.device_type is not None and "swc" not in e.key and e.count:
        pass
ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
from collections import Counter
c = Counter(); t = Counter()
for e in ev:
    n = e.name
    if "anonymous namespace" in n and "at::native" not in n: continue
    c[n[:90]] += 1; t[n[:90]] += e.device_time_total if hasattr(e, "device_time_total") else 0
for k, v in c.most_common(): print(v, round(t[k], 1), k)
# which aten ops launched them
ops_ = Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and e.cpu_parent is None:
        ops_[e.name] += 1
print(ops_.most_common(25))
