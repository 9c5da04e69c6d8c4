#!/usr/bin/env python3
"""swc_proj_ln (split-f16 projection + residual + LayerNorm -> split-f16, one kernel) vs the two launches it replaces (swc_gemm
with an f32 output and a residual, swc_layernorm) at the bench shape of the `mixed` encoder (M = 16000 tokens, N = 768; K = 768:
out-proj, K = 3072: fc2), random operands, buffers rotated so that the residual stream comes from HBM as in the pipeline.
Interleaved rounds, median."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simwhisper_codec_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
N, NB = 768, 8
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
xs = [torch.randn(M, N, device=dev, generator=g) for _ in range(NB)]
lw, lb = 1 + 0.2 * torch.randn(N, device=dev), 0.1 * torch.randn(N, device=dev)
bias = torch.randn(N, device=dev) * 0.1
yn = torch.empty(M, 2 * N, device=dev, dtype=torch.float16)
sa, sw = 64.0, 2.0 ** 12
alpha = 1.0 / (sa * sw)
cases = {}
for K in (768, 3072):
    A = [ops.cast_f16s(torch.randn(M, K, device=dev, generator=g) * 0.5, K, scale=sa) for _ in range(NB if K == 768 else 3)]
    W = ops.cast_f16s(torch.randn(N, K, device=dev, generator=g) * K ** -0.5, K, scale=sw)
    cases[K] = (A, W, ops.proj_ln_pack(W))


def make(K):
    A, W, st = cases[K]

    def fused(i):
        ops.proj_ln(A[i % len(A)], st, bias, alpha, xs[i % NB], M=M, N=N, K=K, ln=(lw, lb), y_next=yn)

    def fused_no_ln(i):
        ops.proj_ln(A[i % len(A)], st, bias, alpha, xs[i % NB], M=M, N=N, K=K)

    def two(i):
        x = xs[i % NB]
        ops.gemm(A[i % len(A)], W, M, N, K, bias=bias, alpha=alpha, residual=x, out=x)
        ops.layernorm(x, lw, lb, 1e-5, B=1, t_in=M, C_=N, out=yn.view(1, M, 2 * N))

    def gemm_only(i):
        x = xs[i % NB]
        ops.gemm(A[i % len(A)], W, M, N, K, bias=bias, alpha=alpha, residual=x, out=x)

    return ((f"K={K} proj_ln", fused), (f"K={K} proj_ln (no LN)", fused_no_ln), (f"K={K} gemm+layernorm", two), (f"K={K} gemm", gemm_only))


ALL = make(768) + make(3072)
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
for name, ts in res.items():
    K = int(name.split()[0][2:])
    t = statistics.median(ts)
    print(f"{name:24s} M={M} {t*1e3:8.1f} us  {2.0*M*N*K/t/1e9:8.1f} TFLOP/s algorithmic  (min {min(ts)*1e3:.1f} us)")
