#!/usr/bin/env python3
"""Host-side cost of one encode+decode step at a small batch (launch-bound): cProfile top entries."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, yaml
from simwhisper_codec_amd import synth
from simwhisper_codec_amd.codec import AudioCodec
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
gp = yaml.safe_load(open(os.path.join(ROOT, "config", "SimWhisperCodec.yaml")))["generator_params"]
m = AudioCodec(gp, precision="mixed"); m.load_state_dict(synth.synth_state_dict(gp), strict=True); m = m.to("cuda").eval()
wavs = [synth.synth_audio(160000, index=i).cuda() for i in range(B)]
def step():
    return m.decode(m.encode(wavs)["codes_list"])
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
t_host = (time.perf_counter() - t0) / 10
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / 10
print(f"B={B}: host enqueue {1e3*t_host:.2f} ms/step, wall {1e3*t_all:.2f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
