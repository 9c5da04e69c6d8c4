"""Where the `mixed` preset's waveform error comes from, and what two cheap changes buy: plain-f16 operands inside the fused
ConvNeXt block (AudioCodec.convnext_f16) and the ISTFT head's two GEMMs on split-f16 operands (AudioCodec.vocos_head_split_f16).
Per setting: Vocos stage error against the reference's own st_y (fixtures), the whole decode against the golden waveforms, and the
B = 32 x 10 s step time on this box.  usage: python tools/probes/decode_error_budget.py"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from common import golden
from test_parity_gpu import _relerr, model
from bench import bench_inputs

DEV = "cuda"
m = model("real", "mixed")
wavs = [w.to(DEV) for w in bench_inputs(32, 160000)]


def step_ms(n=6):
    for _ in range(2):
        m.decode(m.encode(wavs)["codes_list"])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        m.decode(m.encode(wavs)["codes_list"])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for cx16, head16 in ((False, False), (True, False), (False, True), (True, True)):
    m.convnext_f16, m.vocos_head_split_f16 = cx16, head16
    m._pk = None
    keep = m.fused_mlp_min_rows
    out = []
    for name in ("single", "ragged"):
        g = golden("real", name)
        dm = torch.from_numpy(g["st_dec_mel"]).transpose(1, 2).contiguous().to(DEV)
        B, Tv, _ = dm.shape
        for forced in (True, False):
            m.fused_mlp_min_rows = 0 if forced else keep
            with torch.cuda.device(0), torch.inference_mode():
                P = m._packed()
                y = m._vocos(m._cast(dm, P.ddt), B, Tv, P).cpu().numpy()
            out.append(f"vocos/{name}/{'fused' if forced else 'two-gemm'} {_relerr(y, g['st_y']):.2e}")
    m.fused_mlp_min_rows = keep
    print(f"convnext_f16={cx16} head_split_f16={head16}:  " + "  ".join(out) + f"   step {step_ms():.2f} ms", flush=True)
