#!/usr/bin/env python3
"""Micro-benchmark of the small implicit-conv GEMMs of the down- / up-sampler (k7 dilated, 512 -> 512)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simwhisper_codec_amd import ops
dev = "cuda"
for kind, T in (("f16s", 189), ("bf16", 125)):
    B, C, k, dil = 32, 512, 7, 3
    M = B * T
    x = torch.randn(M, C, device=dev); w = torch.randn(C, k * C, device=dev) * 0.02
    if kind == "f16s":
        A = ops.cast_f16s(x, C); W = ops.cast_f16s(w, k * C, scale=2.0 ** 12); alpha = 1.0 / (64 * 2.0 ** 12)
    else:
        A = x.to(torch.bfloat16); W = w.to(torch.bfloat16); alpha = 1.0
    out = torch.empty(M, C, device=dev)
    def run():
        ops.gemm(A, W, M, C, C, lda=C, ldw=k * C, taps=k, dil=dil, pad=3 * dil, t_in=T, t_out=T, alpha=alpha, out=out)
    for _ in range(3): run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 10)
    t = statistics.median(ts)
    print(f"{kind} k7 conv M={M} N={C} K={C}x{k}: {t*1e3:.1f} us  {2.0*M*C*C*k/t/1e9:.1f} TFLOP/s")
