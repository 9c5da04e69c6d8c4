#!/usr/bin/env python3
"""Experiment: the whole step as ONE chain over 32 utterances vs TWO concurrent chains over 16 utterances each (two host
threads, two streams): does kernel-level concurrency fill the tails / epilogues / memory phases of the single chain?"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, yaml
from simwhisper_codec_amd import synth
from simwhisper_codec_amd.codec import AudioCodec
from bench import bench_inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
gp = yaml.safe_load(open(os.path.join(ROOT, "config", "SimWhisperCodec.yaml")))["generator_params"]
sd = synth.synth_state_dict(gp)
models = []
for _ in range(2):
    m = AudioCodec(gp, precision="mixed"); m.load_state_dict(sd); m = m.to("cuda:0").eval(); m.saturation_policy = "off"
    models.append(m)
wavs = [w.cuda() for w in bench_inputs(32, 160000)]
N = 10


def chain(m, ws, stream, n):
    with torch.cuda.stream(stream):
        for _ in range(n):
            m.decode(m.encode(ws)["codes_list"])


s0 = torch.cuda.current_stream()
for m in models:
    chain(m, wavs[:16], s0, 2)
chain(models[0], wavs, s0, 3)
torch.cuda.synchronize()
for rnd in range(3):
    t0 = time.perf_counter(); chain(models[0], wavs, s0, N); torch.cuda.synchronize(); one = (time.perf_counter() - t0) / N
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    ta = threading.Thread(target=chain, args=(models[0], wavs[:16], sa, N)); tb = threading.Thread(target=chain, args=(models[1], wavs[16:], sb, N))
    torch.cuda.synchronize(); t0 = time.perf_counter(); ta.start(); tb.start(); ta.join(); tb.join(); torch.cuda.synchronize()
    two = (time.perf_counter() - t0) / N
    t0 = time.perf_counter(); chain(models[0], wavs[:16], s0, N); torch.cuda.synchronize(); half = (time.perf_counter() - t0) / N
    print(f"round {rnd}: one chain B=32 {one*1e3:.2f} ms | two concurrent chains of B=16 {two*1e3:.2f} ms per pair | one chain B=16 {half*1e3:.2f} ms", flush=True)
