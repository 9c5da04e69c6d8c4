#!/usr/bin/env python3
"""HBM roofline of the memory-bound kernels at the headline shapes (B=32 x 10 s).
For each kernel: algorithmic bytes (its own input + output once), average launch time over
`--iters` back-to-back launches on buffers rotated through > 512 MiB (so nothing is served from the 256 MiB
Infinity Cache), achieved GB/s and the fraction of the 8 TB/s HBM3E peak.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from simwhisper_codec_amd import ops  # noqa: E402

PEAK = 8000.0  # GB/s


_PLUG = None


def timed(fn, sets, iters):
    """Device time per launch.  The host cannot enqueue these short kernels as fast as the GPU runs them, so a
    long matmul is queued first: while it runs the host enqueues the event pair and all `iters` launches, which
    then execute back to back."""
    global _PLUG
    if _PLUG is None:
        _PLUG = torch.randn(12288, 12288, device="cuda", dtype=torch.bfloat16)
    for i in range(3):
        fn(*sets[i % len(sets)])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(4):
        _PLUG @ _PLUG
    e0.record()
    for i in range(iters):
        fn(*sets[i % len(sets)])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--json", default=None)
    ap.add_argument("--only", default=None, help="run only kernels whose name contains this")
    a = ap.parse_args()
    dev = torch.device("cuda")
    B = a.batch
    g = torch.Generator(device="cpu").manual_seed(0)

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(dev)

    rows = []

    def add(name, nbytes, fn, sets):
        if a.only and a.only not in name:
            return
        t = timed(fn, sets, a.iters)
        rows.append({"kernel": name, "bytes": int(nbytes), "us": round(t * 1e6, 2), "GB/s": round(nbytes / t / 1e9, 1),
                     "frac_of_8TB/s": round(nbytes / t / 1e9 / PEAK, 3)})
        print(f"{name:<44s} {nbytes / 1e6:9.2f} MB {t * 1e6:9.2f} us {nbytes / t / 1e9:8.1f} GB/s  {nbytes / t / 1e9 / PEAK:6.1%}",
              flush=True)

    # Vocos ConvNeXt depthwise k7 conv + LayerNorm: f32 residual stream in, bf16 GEMM operand out
    T, C = 1000, 512
    nset = max(2, int(600e6 // (B * T * C * 6)) + 1)
    w7, bia, lw, lb = rnd(7, C, scale=0.3), rnd(C), rnd(C), rnd(C)
    sets = [(rnd(B, T, C), torch.empty(B, T, C, device=dev, dtype=torch.bfloat16)) for _ in range(nset)]
    add("dwconv7_ln f32->bf16 (B,1000,512)", B * T * C * 6,
        lambda x, y: ops.dwconv7_ln(x, w7, bia, lw, lb, 1e-6, B=B, T=T, C_=C, out=y), sets)
    sets = [(rnd(B, T, C), torch.empty(B, T, C, device=dev)) for _ in range(nset)]
    add("dwconv7_ln f32->f32 (B,1000,512)", B * T * C * 8,
        lambda x, y: ops.dwconv7_ln(x, w7, bia, lw, lb, 1e-6, B=B, T=T, C_=C, out=y), sets)

    # yardstick: the practical ceiling at this size — a plain streaming cast/copy of the same bytes by PyTorch
    sets = [(rnd(B, T, C), torch.empty(B, T, C, device=dev, dtype=torch.bfloat16)) for _ in range(nset)]
    add("(yardstick) torch f32->bf16 cast, same bytes", B * T * C * 6, lambda x, y: y.copy_(x), sets)
    sets = [(rnd(B, T, C), torch.empty(B, T, C, device=dev)) for _ in range(nset)]
    add("(yardstick) torch f32->f32 copy, same bytes", B * T * C * 8, lambda x, y: y.copy_(x), sets)

    # transformer LayerNorm 768: f32 in, bf16 / split-f16 out
    T, C = 500, 768
    w, b_ = rnd(C), rnd(C)
    nset = max(2, int(600e6 // (B * T * C * 8)) + 1)
    sets = [(rnd(B, T, C), torch.empty(B, T, C, device=dev, dtype=torch.bfloat16)) for _ in range(nset)]
    add("layernorm f32->bf16 (B,500,768)", B * T * C * 6,
        lambda x, y: ops.layernorm(x, w, b_, 1e-5, B=B, t_in=T, C_=C, out=y), sets)
    sets = [(rnd(B, T, C), torch.empty(B, T, 2 * C, device=dev, dtype=torch.float16)) for _ in range(nset)]
    add("layernorm f32->split-f16 (B,500,768)", B * T * C * 8,
        lambda x, y: ops.layernorm(x, w, b_, 1e-5, B=B, t_in=T, C_=C, out=y), sets)

    # anti-aliased SnakeBeta (down-sampler frames 125+64, up-sampler 125), f32 in, split-f16 / bf16 out
    filt = [0.00202896, 0.00938947, -0.02554346, -0.05765738, 0.12857258, 0.4432098,
            0.4432098, 0.12857258, -0.05765738, -0.02554346, 0.00938947, 0.00202896]
    C = 512
    al, be = rnd(C, scale=0.2), rnd(C, scale=0.2)
    for T, dt, wd, nm in ((189, torch.float16, 2 * C, "split-f16"), (125, torch.bfloat16, C, "bf16")):
        nset = 24
        sets = [(rnd(B, T, C), torch.empty(B, T, wd, device=dev, dtype=dt)) for _ in range(nset)]
        ob = 2 * wd
        add(f"snake_aa f32->{nm} (B,{T},512)", B * T * C * 4 + B * T * ob,
            lambda x, y, T=T: ops.snake_aa(x, al, be, filt, B=B, T=T, C_=C, out=y), sets)

    # FSQ encode / decode (B,125,32)
    T, G = 125, 8
    consts = [3.4965, 2.997, 2.4975, 2.4975, 0.5, 0.0, 0.5, 0.5, 0.14398292, 0.0, 0.20291847, 0.20291847]
    lens = torch.full((B,), T, dtype=torch.int32, device=dev)
    sets = [(rnd(B, T, 32),) for _ in range(8)]
    add("fsq_encode (B,125,32) f32 -> zq f32 + codes i32", B * T * 32 * 8 + B * T * G * 4,
        lambda z: ops.fsq_encode(z, 32, lens, consts, B=B, T=T, t_pad=T, G=G), sets)
    sets = [(torch.randint(0, 2016, (G, B, T), device=dev),) for _ in range(8)]
    add("fsq_decode (8,B,125) i64 -> zq f32", G * B * T * 8 + B * T * 32 * 4,
        lambda c: ops.fsq_decode(c, lens, B=B, T=T, G=G), sets)

    # ISTFT overlap-add: frames (B,1000,640) f32 -> wav (B,160000) f32
    T = 1000
    wsq = torch.hann_window(640, device=dev).square().contiguous()
    sets = [(rnd(B, T, 640),) for _ in range(8)]
    add("istft_ola (B,1000,640) f32 -> (B,160000) f32", B * T * 640 * 4 + B * T * 160 * 4,
        lambda f: ops.istft_ola(f, wsq, B=B, T=T), sets)

    # log-mel framing: wav (B,160000) -> frames (B,1002,400)
    n = torch.full((B,), 160000, dtype=torch.int32, device=dev)
    sets = [(rnd(B, 160000, scale=0.1),) for _ in range(8)]
    add("mel_frames (B,160000) f32 -> (B,1002,400) f32", B * 160000 * 4 + B * 1002 * 400 * 4,
        lambda wv: ops.mel_frames(wv, n, 160000, B=B, T=1002), sets)

    if a.json:
        with open(a.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
