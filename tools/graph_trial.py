#!/usr/bin/env python3
"""Trial: capture encode+decode of fixed shapes in a HIP graph; compare step time with eager launches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, yaml
from simwhisper_codec_amd import synth
from simwhisper_codec_amd.codec import AudioCodec
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
gp = yaml.safe_load(open(os.path.join(ROOT, "config", "SimWhisperCodec.yaml")))["generator_params"]
m = AudioCodec(gp, precision="mixed"); m.load_state_dict(synth.synth_state_dict(gp), strict=True); m = m.to("cuda").eval()
wavs = [synth.synth_audio(160000, index=i).cuda() for i in range(B)]
def step(ws):
    enc = m.encode(ws); return enc, m.decode(enc["codes_list"])
for _ in range(3): step(wavs)
torch.cuda.synchronize()
def timeit(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
t_eager = timeit(lambda: step(wavs))
static = [w.clone() for w in wavs]
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): step(static)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    enc_g, dec_g = step(static)
torch.cuda.synchronize()
ref_enc, ref_dec = step(wavs)
for a, b in zip(static, wavs): a.copy_(b)
g.replay(); torch.cuda.synchronize()
ok = all(torch.equal(a, b) for a, b in zip(enc_g["codes_list"], ref_enc["codes_list"])) and \
     all(torch.equal(a, b) for a, b in zip(dec_g["syn_wav_list"], ref_dec["syn_wav_list"]))
t_graph = timeit(g.replay)
print(f"B={B}: eager {1e3*t_eager:.2f} ms/step, graph {1e3*t_graph:.2f} ms/step, identical outputs: {ok}")
