#!/usr/bin/env python3
"""Yardstick only (not a product path): the vendor library's bf16 GEMM (torch.mm -> hipBLASLt / rocBLAS) on the hot-path shapes,
random operands, no epilogue, beside swc_gemm with its fused epilogue on the same box."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simwhisper_codec_amd import ops

SHAPES = [("qkv", 16000, 2304, 768), ("out_proj", 16000, 768, 768), ("fc1", 16000, 3072, 768), ("fc2", 16000, 768, 3072),
          ("pw1", 32000, 4096, 512), ("pw2", 32000, 512, 4096), ("square 8192", 8192, 8192, 8192)]

def timed(fn, reps=5, rounds=7):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return statistics.median(ts)

for name, M, N, K in SHAPES:
    A = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    Wt = W.t().contiguous()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t_nt = timed(lambda: torch.mm(A, W.t(), out=out))
    t_nn = timed(lambda: torch.mm(A, Wt, out=out))
    t_sw = timed(lambda: ops.gemm(A, W, M, N, K, out=out, ldc=N))
    fl = 2.0 * M * N * K / 1e9
    print(f"{name:12s} M={M:6d} N={N:5d} K={K:5d}  vendor A.W^T {fl/t_nt:7.1f}  vendor A.Wt {fl/t_nn:7.1f}  swc_gemm {fl/t_sw:7.1f} TFLOP/s", flush=True)
