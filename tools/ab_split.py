#!/usr/bin/env python3
"""Decode time of 32 x 30 s (window-0 Vocos call: 32 rows x 2080 frames = 520 tiles of 128 frames on 256 CUs) for several
two-chain splits of the ConvNeXt blocks (utterances in the first chain; 0 = one chain)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, yaml
from simwhisper_codec_amd import synth
from simwhisper_codec_amd.codec import AudioCodec
from bench import bench_inputs
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
gp = yaml.safe_load(open(os.path.join(ROOT, "config", "SimWhisperCodec.yaml")))["generator_params"]
m = AudioCodec(gp, precision="mixed"); m.load_state_dict(synth.synth_state_dict(gp)); m = m.to("cuda:0").eval()
wavs = [w.cuda() for w in bench_inputs(32, 480000)]
codes = m.encode(wavs)["codes_list"]
for rnd in range(2):
    for h in [int(v) for v in sys.argv[1:]] or [0, 31, 28, 24, 16]:
        m.vocos_split_override = h
        for _ in range(2):
            m.decode(codes)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            m.decode(codes)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f"round {rnd} first chain = {h:2d} utterances: decode {dt*1e3:7.2f} ms", flush=True)
