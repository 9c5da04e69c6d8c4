#!/usr/bin/env python3
"""Micro-benchmark of swc_gemm on the hot-path shapes (B=32 x 10 s): TFLOP/s per shape and dtype.
Random operands (zero-filled data reads high on this chip), interleaved rounds, median."""
import sys, os, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simwhisper_codec_amd import ops

SHAPES = [  # (name, M, N, K, gelu, out_bf16)
    ("qkv      ", 16000, 2304, 768, 0, 1), ("out_proj ", 16000, 768, 768, 0, 0), ("fc1+gelu ", 16000, 3072, 768, 1, 1),
    ("fc2      ", 16000, 768, 3072, 0, 0), ("pw1+gelu ", 32000, 4096, 512, 1, 1), ("pwconv2  ", 32000, 512, 4096, 0, 0),
    ("head     ", 32000, 642, 512, 0, 0), ("idft     ", 32000, 640, 648, 0, 0),
]

def main():
    dts = [torch.bfloat16, torch.float32] if len(sys.argv) < 2 else [dict(bf16=torch.bfloat16, f32=torch.float32, f16s=torch.float16)[sys.argv[1]]]
    dev = "cuda"
    for dt in dts:
        tot_f = tot_t = 0.0
        for name, M, N, K, gelu, obf in SHAPES:
            if dt == torch.float16 and K % 32:
                continue
            if dt == torch.float16:
                A = ops.cast_f16s(torch.randn(M, K, device=dev) * 0.5, K)
                W = ops.cast_f16s(torch.randn(N, K, device=dev) * 0.05, K, scale=2.0 ** 14)
            else:
                A = (torch.randn(M, K, device=dev) * 0.5).to(dt)
                W = (torch.randn(N, K, device=dev) * 0.05).to(dt)
            bias = torch.randn(N, device=dev)
            ldc = (N + 7) // 8 * 8
            odt = dt if (obf and dt != torch.float32) else torch.float32
            if odt == torch.float16 and N % 32:
                odt = torch.float32
            out = torch.empty(M, ldc * (2 if odt == torch.float16 else 1), device=dev, dtype=odt)
            res = None if obf else torch.randn(M, ldc, device=dev)
            kw = dict(bias=bias, out=out, ldc=ldc, act=ops.ACT_GELU if gelu else ops.ACT_NONE)
            if res is not None:
                kw.update(residual=res, ldr=ldc)
            for _ in range(3):
                ops.gemm(A, W, M, N, K, **kw)
            torch.cuda.synchronize()
            ts = []
            for _ in range(7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    ops.gemm(A, W, M, N, K, **kw)
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 5)
            t = statistics.median(ts)
            fl = 2.0 * M * N * K
            tot_f += fl; tot_t += t
            print(f"{str(dt)[6:]:9s} {name} M={M:6d} N={N:5d} K={K:5d}  {t*1e3:8.1f} us  {fl/t/1e9:8.1f} TFLOP/s", flush=True)
        print(f"{str(dt)[6:]:9s} total {tot_f/tot_t/1e9:8.1f} TFLOP/s")

if __name__ == "__main__":
    main()
