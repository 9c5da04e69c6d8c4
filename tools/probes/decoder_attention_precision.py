"""How much of the bf16 decoder layers' error is the attention's (q / k / v and the softmax weights rounded to bf16)?  The 12 decoder
layers of the `mixed` preset fed the reference's up-sampler output (st_up), unfused launches, (a) as shipped, (b) with q / k / v
written and the attention computed in split-f16 (f32-class) while every linear keeps its bf16 operands, (c) additionally the
out-projection on the f32-class attention output, against the reference's decoder output (st_dec_mel).
usage: python tools/probes/decoder_attention_precision.py"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from common import golden
from test_parity_gpu import _relerr, model
from test_stages_gpu import _to_f32
from simwhisper_codec_amd import ops

m = model("real", "mixed")
bf = torch.bfloat16


def layers(h, lens, B, T, P, mode):
    D, M, H = h.shape[-1], h.shape[0], P.Hd
    for L in P.dec_layers:
        x = ops.layernorm(h, L.ln1[0], L.ln1[1], 1e-5, B=B, t_in=T, C_=D, out_dtype=bf)
        if mode == "bf16":
            qkv = m._mm(x, L.wqkv, M, 3 * D, D, lda=D, bias=L.bqkv, out_dtype=bf)
            a = ops.attention(qkv, lens, B, T, H)
        else:
            qkv = ops.cast_f16s(m._mm(x, L.wqkv, M, 3 * D, D, lda=D, bias=L.bqkv), 3 * D)   # f32 out of the bf16 GEMM -> split-f16
            a16 = ops.attention(qkv.view(B, T, -1), lens, B, T, H)
            a = _to_f32(a16.view(M, -1), D).to(bf).contiguous()
        m._mm(a, L.wo, M, D, D, lda=D, bias=L.bo, residual=h, out=h)
        x = ops.layernorm(h, L.ln2[0], L.ln2[1], 1e-5, B=B, t_in=T, C_=D, out_dtype=bf)
        f = m._mm(x, L.w1, M, L.b1.shape[0], D, lda=D, bias=L.b1, act=ops.ACT_GELU, out_dtype=bf)
        m._mm(f, L.w2, M, D, L.b1.shape[0], lda=L.b1.shape[0], bias=L.b2, residual=h, out=h)
    return h


for name in ("single", "ragged"):
    g = golden("real", name)
    up = torch.from_numpy(g["st_up"]).transpose(1, 2).contiguous().cuda()
    B, Tt, D = up.shape
    lat = [int(v) for v in g["st_code_lens"]]
    with torch.cuda.device(0), torch.inference_mode():
        P = m._packed()
        lens_h = [min(l * P.stack, Tt) for l in lat]
        lens = m._dev_ints(lens_h, up.device)
        for mode in ("bf16", "attention f32-class"):
            h = layers(up.reshape(B * Tt, D).clone(), lens, B, Tt, P, mode)
            keep = m._transformer
            try:
                m._transformer = lambda x, *a, **k: x          # the layers have run: the output stage of _decoder on their result
                mel = _to_f32(m._decoder(h, lat, B, Tt, P), P.vin)[..., :P.vin].cpu().numpy()
            finally:
                m._transformer = keep
            print(f"{name}: decoder layers {mode:22s} dec_mel rel err {_relerr(mel, g['st_dec_mel'].transpose(0, 2, 1)):.2e}", flush=True)
