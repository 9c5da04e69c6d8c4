#!/usr/bin/env python3
"""Same-process A/B of `AudioCodec.conv1_split_f16` (conv1 on the split-f16 MFMA path with the mel bins padded 80 -> 96,
against conv1 on the exact-f32 path): encode-only and whole-step time at the metric shape, the two models alternating,
and the codes of both compared.

usage: python tools/ab_conv1.py [B=32] [seconds=10] [rounds=6]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, yaml
from simwhisper_codec_amd import synth
from simwhisper_codec_amd.codec import AudioCodec
from bench import bench_inputs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
R = int(sys.argv[3]) if len(sys.argv) > 3 else 6
gp = yaml.safe_load(open(os.path.join(ROOT, "config", "SimWhisperCodec.yaml")))["generator_params"]
sd = synth.synth_state_dict(gp)


def make(flag):
    m = AudioCodec(gp, precision="mixed")
    m.conv1_split_f16 = flag
    m.load_state_dict(sd, strict=True)
    return m.to("cuda:0").eval()


models = {"split-f16 conv1": make(True), "exact-f32 conv1": make(False)}
wavs = [w.cuda() for w in bench_inputs(B, int(secs * 16000))]
codes = {}
for k, m in models.items():
    for _ in range(3):
        c = m.encode(wavs)["codes_list"]
        m.decode(c)
    codes[k] = torch.stack([x.long() for x in c])
a, b = codes.values()
print(f"codes: {int((a != b).sum())} of {a.numel()} differ between the two conv1 paths")


def timed(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


res = {k: {"encode": [], "step": []} for k in models}
for r in range(R):
    for k, m in models.items():
        res[k]["encode"].append(timed(lambda: m.encode(wavs)))
        res[k]["step"].append(timed(lambda: m.decode(m.encode(wavs)["codes_list"])))
for k, v in res.items():
    for w in ("encode", "step"):
        xs = sorted(v[w])
        print(f"{k:18s} {w:7s} median {xs[len(xs) // 2]:7.3f} ms   min {xs[0]:7.3f}   ({' '.join(f'{x:.3f}' for x in v[w])})")
