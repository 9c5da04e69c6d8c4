"""Waveform error of the `mixed` preset's decode against the reference's own waveforms (fixtures) for every combination of the three
split-f16 small-stage options.  usage: python tools/probes/decode_stage_options.py"""
import itertools, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from common import golden
from test_parity_gpu import _relerr, model

m = model("real", "mixed")
m.fused_layer_mlp_min_rows = 0    # the fixtures are a few seconds long: run the decoder layers on swc_layer_tail as the metric's shapes do
for t16, (head, io, up) in itertools.product((False, True), ((False, False, False), (True, False, False), (True, True, False), (True, True, True))):
    m.layer_tail_f16 = t16
    m.vocos_head_split_f16, m.decoder_io_split_f16, m.upsample_split_f16 = head, io, up
    m._pk = None
    errs = []
    for name in ("single", "ragged", "short"):
        g = golden("real", name)
        n = len(g["spec_n"])
        codes = [torch.from_numpy(g[f"codes_{i}"]).cuda() for i in range(n)]
        wav = m.decode(codes)["syn_wav_list"]
        errs.append(max(_relerr(wav[i].cpu().numpy(), g[f"wav_{i}"]) for i in range(n) if g[f"wav_{i}"].size))
    print(f"tail_f16={t16!s:5s} head={head!s:5s} io={io!s:5s} up={up!s:5s}  " + "  ".join(f"{e:.2e}" for e in errs), flush=True)
