#!/usr/bin/env python3
"""Throughput of encode()+decode() with 1..4 batches in flight (pipeline.InFlight) at a given batch size.
usage: inflight_depth.py [batch] [seconds] [precision]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, yaml
from simwhisper_codec_amd import synth
from simwhisper_codec_amd.codec import AudioCodec
from simwhisper_codec_amd.pipeline import InFlight
from bench import bench_inputs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
prec = sys.argv[3] if len(sys.argv) > 3 else "mixed"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
gp = yaml.safe_load(open(os.path.join(ROOT, "config", "SimWhisperCodec.yaml")))["generator_params"]
m = AudioCodec(gp, precision=prec); m.load_state_dict(synth.synth_state_dict(gp), strict=True); m = m.to("cuda:0").eval()
wavs = [w.cuda() for w in bench_inputs(B, int(secs * 16000))]


def step(model, w):
    return model.decode(model.encode(w, overlap_seconds=10)["codes_list"], overlap_seconds=10)


for _ in range(3):
    step(m, wavs)
for depth in (1, 2, 3, 4):
    with InFlight(m, depth) as pipe:
        pipe.map(step, [wavs] * (2 * depth))
        best = 1e9
        for rnd in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            pipe.map(step, [wavs] * 24)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 24)
        print(f"B={B} x {secs:g}s {prec}: {depth} in flight: {best*1e3:7.3f} ms/step  {B*secs/best:8.1f} audio-s/s", flush=True)
