import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, yaml
from simwhisper_codec_amd import synth
from simwhisper_codec_amd.codec import AudioCodec
from oracle.ref_cpu import Oracle
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
gp = yaml.safe_load(open(os.path.join(ROOT, "config", "SimWhisperCodec.yaml")))["generator_params"]
sd = synth.synth_state_dict(gp)
m = AudioCodec(gp, precision="mixed"); m.load_state_dict(sd, strict=True); m = m.to("cuda").eval()
# 1. one long utterance (210 s: 10 windows of 30 s every 20 s) + a short one in the same batch
wavs = [synth.synth_audio(210 * 16000 + 777, index=900, kind="speech"), synth.synth_audio(3 * 16000, index=901, kind="speech")]
t0 = time.time(); enc = m.encode([w.cuda() for w in wavs]); dec = m.decode(enc["codes_list"]); torch.cuda.synchronize()
print(f"long-form: codes {[tuple(c.shape) for c in enc['codes_list']]}, wav {[int(w.numel()) for w in dec['syn_wav_list']]}, {time.time()-t0:.2f} s (first call)")
torch.set_num_threads(16)
o = Oracle(gp, sd)
t0 = time.time(); want = o.encode(wavs, trim=True)["codes_list"]; print(f"oracle encode {time.time()-t0:.0f} s")
for a, b in zip(enc["codes_list"], want):
    print("  code mismatches", int((a.cpu().long() != b.long()).sum()), "of", b.numel())
ww = o.decode(want)["syn_wav_list"]
for a, b in zip(dec["syn_wav_list"], ww):
    d = (a.float().cpu() - b).abs().max().item() / max(b.abs().max().item(), 1e-9)
    print("  waveform rel err", round(d, 5), "finite", bool(torch.isfinite(a).all()))
# 2. 256 x 10 s in one call on one GPU (configs[3]'s total batch)
from bench import bench_inputs
w256 = [w.cuda() for w in bench_inputs(32, 160000)] * 8
torch.cuda.reset_peak_memory_stats()
t0 = time.time(); enc = m.encode(w256); dec = m.decode(enc["codes_list"]); torch.cuda.synchronize(); t1 = time.time() - t0
t0 = time.time(); enc = m.encode(w256); dec = m.decode(enc["codes_list"]); torch.cuda.synchronize(); t2 = time.time() - t0
same = all(torch.equal(enc["codes_list"][i], enc["codes_list"][i % 32]) for i in range(256))
print(f"256 x 10 s in one call: {t2*1e3:.1f} ms ({2560/t2:.0f} audio-s/s), first {t1*1e3:.0f} ms, peak memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB, rows repeat exactly: {same}")
