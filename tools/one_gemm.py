#!/usr/bin/env python3
"""Launch one GEMM shape a few times (for rocprofv3 --pmc runs): one_gemm.py M N K [bf16|f16s|f32] [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simwhisper_codec_amd import ops
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kind = sys.argv[4] if len(sys.argv) > 4 else "bf16"
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = "cuda"
if kind == "f16s":
    A = ops.cast_f16s(torch.randn(M, K, device=dev) * 0.5, K); W = ops.cast_f16s(torch.randn(N, K, device=dev) * 0.05, K, scale=2.0 ** 14)
else:
    dt = torch.bfloat16 if kind == "bf16" else torch.float32
    A = (torch.randn(M, K, device=dev) * 0.5).to(dt); W = (torch.randn(N, K, device=dev) * 0.05).to(dt)
out = torch.empty(M, N, device=dev, dtype=torch.float32)
for _ in range(iters):
    ops.gemm(A, W, M, N, K, out=out)
torch.cuda.synchronize()
print("done")
