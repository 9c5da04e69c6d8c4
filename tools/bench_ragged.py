#!/usr/bin/env python3
"""Ragged batch (32 utterances of 2..28 s): one padded call per stage vs length-bucketed calls (AudioCodec.length_bucketing)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, yaml
from simwhisper_codec_amd import synth, spec
from simwhisper_codec_amd.codec import AudioCodec, length_groups
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
gp = yaml.safe_load(open(os.path.join(ROOT, "config", "SimWhisperCodec.yaml")))["generator_params"]
m = AudioCodec(gp, precision="mixed"); m.load_state_dict(synth.synth_state_dict(gp)); m = m.to("cuda:0").eval()
g = torch.Generator().manual_seed(3)
lens = [int(16000 * (2 + 26 * torch.rand(1, generator=g).item())) for _ in range(32)]   # 2 .. 28 s, random order
wavs = [0.1 * torch.randn(n, generator=g).cuda() for n in lens]
audio = sum(lens) / 16000
srt = sorted(lens, reverse=True)
print("encode groups (tokens):", [(a, b, spec.token_len(srt[a])) for a, b in length_groups([spec.token_len(v) for v in srt])])
need = [min(srt[0] // 1280, v // 1280 + 64) for v in srt]
print("decode groups (code frames):", [(a, b, need[a]) for a, b in length_groups([4 * v for v in need])])
for pack, bucket in ((False, False), (False, True), (True, False), (True, True)) * 2:
    m.varlen_packing, m.length_bucketing = pack, bucket
    ovh = 5000
    for _ in range(2):
        c = m.encode(wavs)["codes_list"]; m.decode(c)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(6):
        c = m.encode(wavs)["codes_list"]
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 6
    t0 = time.perf_counter()
    for _ in range(6):
        m.decode(c)
    torch.cuda.synchronize(); td = (time.perf_counter() - t0) / 6
    print(f"packing={pack} bucketing={bucket}: encode {te*1e3:6.2f} ms  decode {td*1e3:6.2f} ms  = {audio/(te+td):6.0f} audio-s/s ({audio:.0f} s of audio)", flush=True)
