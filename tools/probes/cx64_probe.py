"""occupancy / stagger probe of the swc_convnext64_mlp experiment: kernel time against M and against the start stagger"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simwhisper_codec_amd import ops
C, I = 512, 4096
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
w1 = (torch.randn(I, C, device=dev, generator=g) * C ** -0.5).to(torch.bfloat16)
w2 = (torch.randn(C, I, device=dev, generator=g) * I ** -0.5).to(torch.bfloat16)
b1, b2, gam = torch.randn(I, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1
ws64, ws = ops.convnext64_pack(w1, w2), ops.convnext_pack(w1, w2, gam)


def t(fn, n=6):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M in (8000, 16000, 16064, 24000, 32000, 64000):
    y = torch.randn(M, C, device=dev, generator=g).to(torch.bfloat16)
    x = torch.randn(M, C, device=dev, generator=g)
    base = t(lambda: ops.convnext_mlp(y, ws, b1, b2, gam, x, M=M, C_=C, I=I))
    row = [f"M={M:6d} 128-frame kernel {base:7.1f} us | 64-frame:"]
    for st in (0, 50000, 400000):
        row.append(f"st{st} {t(lambda: ops.convnext64_mlp(y, ws64, b1, b2, gam, x, M=M, C_=C, I=I, stagger_cycles=st)):7.1f}")
    print(" ".join(row), flush=True)
