"""Which decode stage the `mixed` preset's waveform error comes from: every combination of (up-sampler, decoder, Vocos) run by the
`mixed` model or by the exact-f32 one, from the reference's own quantised latents (st_zq) to the waveform, against the reference's
waveform (st_y).  usage: python tools/probes/decode_error_attribution.py"""
import itertools, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from common import golden
from test_parity_gpu import _relerr, model
from test_stages_gpu import _to_f32

DEV = "cuda"
mm = {"mixed": model("real", "mixed"), "fp32": model("real", "fp32")}
for name in ("single", "ragged"):
    g = golden("real", name)
    T = int(g["st_code_lens"].max())
    lat = [int(v) for v in g["st_code_lens"]]
    zq = torch.from_numpy(g["st_zq"][:, :, :T]).transpose(1, 2).contiguous().to(DEV)
    B = zq.shape[0]
    for a, b, c in itertools.product(("mixed", "fp32"), repeat=3):
        with torch.cuda.device(0), torch.inference_mode():
            Pa, Pb, Pc = mm[a]._packed(), mm[b]._packed(), mm[c]._packed()
            Tt = Pa.stack * T
            x = mm[a]._upsample(zq, B, T, Pa)
            mel = _to_f32(mm[b]._decoder(x.clone(), lat, B, Tt, Pb), Pb.vin)[..., :Pb.vin].reshape(B, 2 * Tt, Pb.vin).contiguous()
            y = mm[c]._vocos(mm[c]._cast(mel, Pc.ddt), B, 2 * Tt, Pc).cpu().numpy()
        n = g["st_y"].shape[-1]
        print(f"{name}: up={a:5s} dec={b:5s} vocos={c:5s}  waveform rel err {_relerr(y[:, :n], g['st_y']):.2e}", flush=True)
