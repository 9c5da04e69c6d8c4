#!/usr/bin/env python3
"""What the range guard costs per step: saturation_policy "fallback" (one counter read-back per encode) vs "off"."""
import sys, time, torch, yaml, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simwhisper_codec_amd import synth
from simwhisper_codec_amd.codec import AudioCodec
from bench import bench_inputs
gp = yaml.safe_load(open(os.path.join(ROOT, "config", "SimWhisperCodec.yaml")))["generator_params"]
sd = synth.synth_state_dict(gp)
for prec in ("mixed", "fp8"):
    m = AudioCodec(gp, precision=prec); m.load_state_dict(sd); m = m.to("cuda:0").eval()
    for B in (32, 8):
        wavs = [w.cuda() for w in bench_inputs(B, 160000)]
        for pol in ("fallback", "off", "fallback", "off"):
            m.saturation_policy = pol
            for _ in range(3): m.decode(m.encode(wavs)["codes_list"])
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): m.decode(m.encode(wavs)["codes_list"])
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
            print(prec, B, pol, f"{dt*1e3:.3f} ms")
