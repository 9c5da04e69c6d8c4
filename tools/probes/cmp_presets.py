import os, sys, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from simwhisper_codec_amd import synth
from simwhisper_codec_amd.codec import AudioCodec
from bench import bench_inputs
gp = yaml.safe_load(open(os.path.join(ROOT, "config", "SimWhisperCodec.yaml")))["generator_params"]
sd = synth.synth_state_dict(gp)
wavs = [w.cuda() for w in bench_inputs(32, 160000)]
def timed(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ms = {}
for prec in ("bf16", "fp8_fc1", "fp8"):
    for pol in ("fallback", "off"):
        m = AudioCodec(gp, precision=prec); m.load_state_dict(sd, strict=True); m = m.to("cuda:0").eval()
        m.saturation_policy = pol
        for _ in range(3): c = m.encode(wavs)["codes_list"]; m.decode(c)
        enc = [timed(lambda: m.encode(wavs)) for _ in range(3)]
        dec = [timed(lambda: m.decode(c)) for _ in range(3)]
        stp = [timed(lambda: m.decode(m.encode(wavs)["codes_list"])) for _ in range(3)]
        print(prec, pol, "encode %.3f decode %.3f step %.3f" % (min(enc), min(dec), min(stp)), flush=True)
        del m; torch.cuda.empty_cache()
