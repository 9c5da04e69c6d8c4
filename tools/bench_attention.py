#!/usr/bin/env python3
"""swc_attention16 at the bench shape (32 utterances x 500 tokens x 12 heads x 64): time per call, algorithmic TFLOP/s."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simwhisper_codec_amd import ops
B, T, H = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 500, 12
lens = torch.full((B,), T, dtype=torch.int32, device="cuda")
x = torch.randn(B * T, 3 * H * 64, device="cuda")
for name, q in (("bf16", x.to(torch.bfloat16).view(B, T, -1)), ("f16s", ops.cast_f16s(x, 3 * H * 64).view(B, T, -1))):
    out = None
    for _ in range(3):
        out = ops.attention(q, lens, B, T, H, out=out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.attention(q, lens, B, T, H, out=out)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    t = statistics.median(ts)
    print(f"{name} T={T}: {t*1e3:7.1f} us  {4.0*B*H*T*T*64/t/1e9:7.1f} TFLOP/s", flush=True)
