#!/usr/bin/env python3
"""Per-stage device time of one bench step (SWC_TRACE=time: event pairs around every stage of the path, trace.py)."""
import os, sys
os.environ.setdefault("SWC_TRACE", "time")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, yaml
from simwhisper_codec_amd import synth, trace
from simwhisper_codec_amd.codec import AudioCodec
from bench import bench_inputs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
gp = yaml.safe_load(open(os.path.join(ROOT, "config", "SimWhisperCodec.yaml")))["generator_params"]
m = AudioCodec(gp, precision=sys.argv[3] if len(sys.argv) > 3 else "mixed")
m.load_state_dict(synth.synth_state_dict(gp), strict=True)
m = m.to("cuda:0").eval()
wavs = [w.cuda() for w in bench_inputs(B, int(secs * 16000))]
for _ in range(3):
    m.decode(m.encode(wavs)["codes_list"])
trace.report()
N = 10
for _ in range(N):
    m.decode(m.encode(wavs)["codes_list"])
rep = trace.report()
tot = sum(v["ms"] for v in rep.values()) / N
print(f"B={B} x {secs:g}s  stage device time per step (sum {tot:.2f} ms)")
for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"]):
    print(f"  {k:12s} {v['ms'] / N:8.3f} ms  {100 * v['ms'] / N / tot:5.1f} %  ({v['calls'] // N} call(s)/step)")
