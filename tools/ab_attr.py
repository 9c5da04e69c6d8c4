#!/usr/bin/env python3
"""Same-process A/B of one AudioCodec tunable (a class attribute read at call time): encode / decode / whole-step time at
the metric shape, the settings alternating in interleaved rounds, and the outputs of the settings compared.

usage: python tools/ab_attr.py <attribute> <value> <value> [...] [--precision mixed] [--B 32] [--seconds 10] [--rounds 6]
   e.g. python tools/ab_attr.py fused_layer_mlp_min_rows 10240 1099511627776
"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, yaml
from simwhisper_codec_amd import synth
from simwhisper_codec_amd.codec import AudioCodec
from bench import bench_inputs

ap = argparse.ArgumentParser()
ap.add_argument("attr"); ap.add_argument("values", nargs="+")
ap.add_argument("--precision", default="mixed"); ap.add_argument("--B", type=int, default=32)
ap.add_argument("--seconds", type=float, default=10.0); ap.add_argument("--rounds", type=int, default=6)
a = ap.parse_args()
gp = yaml.safe_load(open(os.path.join(ROOT, "config", "SimWhisperCodec.yaml")))["generator_params"]
m = AudioCodec(gp, precision=a.precision); m.load_state_dict(synth.synth_state_dict(gp), strict=True); m = m.to("cuda:0").eval()
wavs = [w.cuda() for w in bench_inputs(a.B, int(a.seconds * 16000))]


def conv(v):
    try:
        return int(v)
    except ValueError:
        try:
            return float(v)
        except ValueError:
            return {"True": True, "False": False, "None": None}.get(v, v)


vals = [conv(v) for v in a.values]


def timed(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


outs, res = {}, {v: {"encode": [], "decode": [], "step": []} for v in vals}
PACK_TIME = ("layer_fusion", "conv1_split_f16", "convnext_f16", "layer_tail_f16", "vocos_head_split_f16", "decoder_io_split_f16", "upsample_split_f16")


def setv(v):
    for name in a.attr.split(","):   # several attributes at once: "vocos_head_split_f16,decoder_io_split_f16,upsample_split_f16"
        setattr(m, name, v)
        if name in PACK_TIME:   # read at pack time: re-pack
            m._pk = None


for v in vals:
    setv(v)
    for _ in range(3):
        c = m.encode(wavs)["codes_list"]; w = m.decode(c)["syn_wav_list"]
    outs[v] = (torch.stack([x.long() for x in c]), torch.stack(list(w)))
c0, w0 = outs[vals[0]]
for v in vals[1:]:
    c1, w1 = outs[v]
    print(f"{a.attr}={v} vs {vals[0]}: {int((c0 != c1).sum())} of {c0.numel()} codes differ, waveform max |d| / peak "
          f"{float((w0 - w1).abs().max() / w0.abs().max()):.3e}")
for r in range(a.rounds):
    for v in vals:
        setv(v)
        c = m.encode(wavs)["codes_list"]
        res[v]["encode"].append(timed(lambda: m.encode(wavs)))
        res[v]["decode"].append(timed(lambda: m.decode(c)))
        res[v]["step"].append(timed(lambda: m.decode(m.encode(wavs)["codes_list"])))
for v in vals:
    for k in ("encode", "decode", "step"):
        xs = sorted(res[v][k])
        print(f"{a.attr}={str(v):>14s} {k:7s} median {xs[len(xs) // 2]:7.3f} ms   min {xs[0]:7.3f}   ({' '.join(f'{x:.3f}' for x in res[v][k])})")
