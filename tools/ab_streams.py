#!/usr/bin/env python3
"""Whole-step time with the ConvNeXt blocks as one chain (vocos_streams = 1) vs two out-of-phase half-batch chains on two
streams (vocos_streams = 2), same process, interleaved rounds.  usage: ab_streams.py [phase_us ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, yaml
from simwhisper_codec_amd import synth
from simwhisper_codec_amd.codec import AudioCodec
from bench import bench_inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
gp = yaml.safe_load(open(os.path.join(ROOT, "config", "SimWhisperCodec.yaml")))["generator_params"]
m = AudioCodec(gp, precision="mixed"); m.load_state_dict(synth.synth_state_dict(gp)); m = m.to("cuda:0").eval()
wavs = [w.cuda() for w in bench_inputs(32, 160000)]
codes = m.encode(wavs)["codes_list"]
phases = [int(v) for v in sys.argv[1:]] or [130]
variants = [(1, 0)] + [(2, p) for p in phases]
for rnd in range(3):
    for streams, ph in variants:
        m.vocos_streams, m.vocos_phase_us = streams, ph
        for _ in range(3):
            m.decode(codes)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            m.decode(codes)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f"round {rnd} streams={streams} phase_us={ph:4d} decode {dt * 1e3:7.3f} ms", flush=True)
