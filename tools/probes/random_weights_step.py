#!/usr/bin/env python3
"""Step time of the metric shape with RANDOM weights (N(0, 1 / fan_in) matrices, unit norms, small biases) instead of the closed-form
synthetic checkpoint: trained weights look like the former, and the clock the chip holds depends on the operands' bit patterns
(profiles/r03_convnext_mfma16.txt).  Timing only: nothing here is compared with the oracle.
usage: python tools/probes/random_weights_step.py [synthetic|random] [steps=20]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, yaml
from simwhisper_codec_amd import synth
from simwhisper_codec_amd.codec import AudioCodec
from bench import bench_inputs

kind = sys.argv[1] if len(sys.argv) > 1 else "random"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
gp = yaml.safe_load(open(os.path.join(ROOT, "config", "SimWhisperCodec.yaml")))["generator_params"]
sd = synth.synth_state_dict(gp)
if kind == "random":
    g = torch.Generator().manual_seed(7)
    new = {}
    for k, v in sd.items():
        if "filter" in k or "window" in k or "basis" in k:
            new[k] = v.clone()                       # fixed buffers (anti-alias filters, ISTFT window)
        elif v.dtype.is_floating_point and v.dim() >= 2:
            fan_in = v[0].numel()
            new[k] = torch.randn(v.shape, generator=g) * fan_in ** -0.5
        elif v.dtype.is_floating_point and v.dim() == 1 and (k.endswith("weight") or k.endswith("weight_g") or "alpha" in k or "beta" in k or "gamma" in k):
            new[k] = v.clone()                       # norm scales, snake parameters, layer scales: keep the synthetic ones
        elif v.dtype.is_floating_point:
            new[k] = torch.randn(v.shape, generator=g) * 0.02
        else:
            new[k] = v.clone()
    sd = new
m = AudioCodec(gp, precision="mixed")
m.saturation_policy = "off"   # random weights may clip the split-f16 range: timing only, no fall-back to f32
m.load_state_dict(sd, strict=True)
m = m.to("cuda:0").eval()
wavs = [w.cuda() for w in bench_inputs(32, 160000)]
for _ in range(4):
    out = m.decode(m.encode(wavs)["codes_list"])["syn_wav_list"]
torch.cuda.synchronize()
finite = all(bool(torch.isfinite(w).all()) for w in out)
t0 = time.perf_counter()
for _ in range(steps):
    m.decode(m.encode(wavs)["codes_list"])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"{kind:9s} weights: {dt * 1e3:7.3f} ms per step = {320 / dt:8.1f} audio-s/s   (outputs finite: {finite})", flush=True)
