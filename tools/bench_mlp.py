#!/usr/bin/env python3
"""Fused transformer MLP block (swc_mlp_block) vs the launches it replaces (swc_layernorm, swc_gemm fc1 + GELU, swc_gemm
fc2 + residual, swc_layernorm of the next sub-block) at the bench shape (M = 16000 tokens, D = 768, F = 3072), random
operands, buffers rotated so that the residual stream comes from HBM as in the pipeline.  Interleaved rounds, median."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simwhisper_codec_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
D, F_, NB = 768, 3072, 8
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
xs = [torch.randn(M, D, device=dev, generator=g) for _ in range(NB)]
w1 = (torch.randn(F_, D, device=dev, generator=g) * D ** -0.5).to(torch.bfloat16)
w2 = (torch.randn(D, F_, device=dev, generator=g) * F_ ** -0.5).to(torch.bfloat16)
b1, b2 = torch.randn(F_, device=dev) * 0.1, torch.randn(D, device=dev) * 0.1
lw, lb = 1 + 0.2 * torch.randn(D, device=dev), 0.1 * torch.randn(D, device=dev)
ws = ops.mlp_pack(w1, w2)
wo = (torch.randn(D, D, device=dev, generator=g) * D ** -0.5).to(torch.bfloat16)
bo = torch.randn(D, device=dev) * 0.1
wts = ops.layer_tail_pack(wo, w1, w2)
att = [(torch.randn(M, D, device=dev, generator=g) * 0.5).to(torch.bfloat16) for _ in range(NB)]
hh = torch.empty(M, F_, device=dev, dtype=torch.bfloat16)
yb = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
yn = torch.empty(M, D, device=dev, dtype=torch.bfloat16)


def fused(i):
    ops.mlp_block(xs[i % NB], lw, lb, 1e-5, ws, b1, b2, M=M, D=D, F=F_, next_ln=(lw, lb), y_next=yn)


def fused_no_next(i):
    ops.mlp_block(xs[i % NB], lw, lb, 1e-5, ws, b1, b2, M=M, D=D, F=F_)


def four(i):
    x = xs[i % NB]
    ops.layernorm(x, lw, lb, 1e-5, B=1, t_in=M, C_=D, out=yb.view(1, M, D))
    ops.gemm(yb, w1, M, F_, D, bias=b1, act=ops.ACT_GELU, out=hh)
    ops.gemm(hh, w2, M, D, F_, bias=b2, residual=x, out=x)
    ops.layernorm(x, lw, lb, 1e-5, B=1, t_in=M, C_=D, out=yn.view(1, M, D))


def two_gemm(i):
    x = xs[i % NB]
    ops.gemm(yb, w1, M, F_, D, bias=b1, act=ops.ACT_GELU, out=hh)
    ops.gemm(hh, w2, M, D, F_, bias=b2, residual=x, out=x)


def tail(i):  # out-proj + the whole MLP sub-block + next LayerNorm: one kernel
    ops.layer_tail(att[i % NB], xs[i % NB], wts, bo, lw, lb, 1e-5, b1, b2, M=M, D=D, F=F_, next_ln=(lw, lb), y_next=yn)


import math
w1f = torch.randn(F_, D, device=dev, generator=g) * D ** -0.5
sw8 = 2.0 ** math.floor(math.log2(448.0 / float(w1f.abs().max())))
w1q = ops.cast_fp8(w1f, sw8)
wt8 = ops.layer_tail_pack(wo, w1q, w2)


def tail_fp8(i):  # the same kernel with fc1 on the block-scaled fp8 MFMA (preset fp8_fc1)
    ops.layer_tail(att[i % NB], xs[i % NB], wt8, bo, lw, lb, 1e-5, b1, b2, M=M, D=D, F=F_, next_ln=(lw, lb), y_next=yn,
                   fc1_dtype=ops.FP8_T, fc1_alpha=1.0 / (ops.FP8_ACT_SCALE * sw8))


def oproj_fused(i):  # the same work as two launches: out-proj GEMM, swc_mlp_block
    x = xs[i % NB]
    ops.gemm(att[i % NB], wo, M, D, D, bias=bo, residual=x, out=x)
    ops.mlp_block(x, lw, lb, 1e-5, ws, b1, b2, M=M, D=D, F=F_, next_ln=(lw, lb), y_next=yn)


def oproj_four(i):  # ... and as the five launches of round 3
    x = xs[i % NB]
    ops.gemm(att[i % NB], wo, M, D, D, bias=bo, residual=x, out=x)
    four(i)


ALL = (("layer_tail fp8fc1", tail_fp8), ("layer_tail", tail), ("oproj+fused", oproj_fused), ("oproj+ln+2gemm+ln", oproj_four), ("fused", fused), ("fused_no_next", fused_no_next), ("ln+2gemm+ln", four), ("two_gemm", two_gemm))
res = {n: [] for n, _ in ALL}
for _, fn in ALL:
    for i in range(3):
        fn(i)
torch.cuda.synchronize()
for rnd in range(7):
    for name, fn in ALL:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(8):
            fn(i)
        e1.record(); torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / 8)
fl = 4.0 * M * D * F_
for name, ts in res.items():
    t = statistics.median(ts)
    print(f"{name:18s} M={M} {t*1e3:8.1f} us  {fl/t/1e9:8.1f} TFLOP/s   (min {min(ts)*1e3:.1f} us)")
