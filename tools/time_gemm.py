#!/usr/bin/env python3
"""time_gemm.py M N K [kind=bf16|f16s|f32] [pad_elems]: GEMM timing (f32 out), optional row padding of A and W."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simwhisper_codec_amd import ops
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kind = sys.argv[4] if len(sys.argv) > 4 else "bf16"
pad = int(sys.argv[5]) if len(sys.argv) > 5 else 0
dev = "cuda"
if kind == "f16s":
    A = ops.cast_f16s(torch.randn(M, K, device=dev) * 0.5, K); W = ops.cast_f16s(torch.randn(N, K, device=dev) * 0.05, K, scale=2.0 ** 14)
    kw = {}
else:
    dt = torch.bfloat16 if kind == "bf16" else torch.float32
    A = (torch.randn(M, K + pad, device=dev) * 0.5).to(dt); W = (torch.randn(N, K + pad, device=dev) * 0.05).to(dt)
    kw = dict(lda=K + pad, ldw=K + pad)
out = torch.empty(M, N, device=dev, dtype=torch.float32)
kw["out"] = out
for _ in range(3): ops.gemm(A, W, M, N, K, **kw)
torch.cuda.synchronize()
ts=[]
for _ in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.gemm(A, W, M, N, K, **kw)
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)/5)
t=statistics.median(ts)
print(f"{kind} pad={pad} M={M} N={N} K={K} {t*1e3:.1f} us {2.0*M*N*K/t/1e9:.1f} TFLOP/s")
