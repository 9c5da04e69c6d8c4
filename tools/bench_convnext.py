#!/usr/bin/env python3
"""Fused ConvNeXt MLP kernel vs the two swc_gemm calls it replaces, at the bench shape (M = 32000 Vocos frames, C = 512,
I = 4096), random operands, buffers rotated so that y / x come from HBM as in the pipeline.  Interleaved rounds, median."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simwhisper_codec_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 32000
C, I, NB = 512, 4096, 6
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
ys = [torch.randn(M, C, device=dev, generator=g).to(torch.bfloat16) for _ in range(NB)]
xs = [torch.randn(M, C, device=dev, generator=g) for _ in range(NB)]
w1 = (torch.randn(I, C, device=dev, generator=g) * C ** -0.5).to(torch.bfloat16)
w2 = (torch.randn(C, I, device=dev, generator=g) * I ** -0.5).to(torch.bfloat16)
b1, b2, gam = torch.randn(I, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1
ws = ops.convnext_pack(w1, w2, gam)
hh = torch.empty(M, I, device=dev, dtype=torch.bfloat16)


def fused(i):
    ops.convnext_mlp(ys[i % NB], ws, b1, b2, gam, xs[i % NB], M=M, C_=C, I=I)


w7, db = torch.randn(7, C, device=dev) * 0.3, torch.randn(C, device=dev) * 0.1
lw, lb = 1 + 0.2 * torch.randn(C, device=dev), 0.1 * torch.randn(C, device=dev)
Bu, Tu = M // 1000 if M % 1000 == 0 else 1, 1000 if M % 1000 == 0 else M
xo = [torch.empty(M, C, device=dev) for _ in range(2)]


def block(i):  # the whole ConvNeXt block in one kernel (depthwise conv + LayerNorm included)
    ops.convnext_block(xs[i % NB], xo[i % 2], w7, db, lw, lb, 1e-6, ws, b1, b2, gam, B=Bu, T=Tu, C_=C, I=I)


def three(i):  # what the block was before: dwconv7_ln + two GEMMs
    yy = ops.dwconv7_ln(xs[i % NB], w7, db, lw, lb, 1e-6, B=Bu, T=Tu, C_=C, out_dtype=torch.bfloat16)
    ops.gemm(yy, w1, M, I, C, bias=b1, act=ops.ACT_GELU, out=hh)
    ops.gemm(hh, w2, M, C, I, bias=b2, gamma=gam, residual=xs[i % NB], out=xs[i % NB])


def dw_mlp(i):  # dwconv7_ln + fused MLP kernel
    yy = ops.dwconv7_ln(xs[i % NB], w7, db, lw, lb, 1e-6, B=Bu, T=Tu, C_=C, out_dtype=torch.bfloat16)
    ops.convnext_mlp(yy, ws, b1, b2, gam, xs[i % NB], M=M, C_=C, I=I)


def two(i):
    ops.gemm(ys[i % NB], w1, M, I, C, bias=b1, act=ops.ACT_GELU, out=hh)
    ops.gemm(hh, w2, M, C, I, bias=b2, gamma=gam, residual=xs[i % NB], out=xs[i % NB])


ws64 = ops.convnext64_pack(w1, w2)
STAG = [int(v) for v in os.environ.get("CX64_STAGGER", "0,20000,40000,80000").split(",")]


def mk64(st):
    def f(i):  # the round-4 experiment: 64-frame tiles, two workgroups per CU, second one started `st` cycles late
        ops.convnext64_mlp(ys[i % NB], ws64, b1, b2, gam, xs[i % NB], M=M, C_=C, I=I, stagger_cycles=st)
    return f


ALL = (("fused", fused), ("two_gemm", two), ("block", block), ("dw+mlp", dw_mlp), ("dw+2gemm", three)) + tuple(
    (f"mlp64 st{st}", mk64(st)) for st in STAG)
res = {n: [] for n, _ in ALL}
for _, fn in ALL:
    for i in range(3):
        fn(i)
torch.cuda.synchronize()
for rnd in range(7):
    for name, fn in ALL:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(6):
            fn(i)
        e1.record(); torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / 6)
fl = 4.0 * M * C * I
for name, ts in res.items():
    t = statistics.median(ts)
    print(f"{name:14s} M={M} {t*1e3:8.1f} us  {fl/t/1e9:8.1f} TFLOP/s   (min {min(ts)*1e3:.1f} us)")
