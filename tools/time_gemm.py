#!/usr/bin/env python3
"""time_gemm.py M N K [pad_elems]: bf16 GEMM timing with optional row padding of A and W (lda = ldw = K + pad)."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simwhisper_codec_amd import ops
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
pad = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = "cuda"
A = (torch.randn(M, K + pad, device=dev) * 0.5).to(torch.bfloat16); W = (torch.randn(N, K + pad, device=dev) * 0.05).to(torch.bfloat16)
out = torch.empty(M, N, device=dev, dtype=torch.float32)
kw = dict(lda=K + pad, ldw=K + pad, out=out)
for _ in range(3): ops.gemm(A, W, M, N, K, **kw)
torch.cuda.synchronize()
ts=[]
for _ in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.gemm(A, W, M, N, K, **kw)
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)/5)
t=statistics.median(ts)
print(f"dbg={os.environ.get('SWC_GEMM_DBG','0')} pad={pad} M={M} N={N} K={K} {t*1e3:.1f} us {2.0*M*N*K/t/1e9:.1f} TFLOP/s")
