import os, sys, statistics
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from simwhisper_codec_amd import ops
dev = "cuda"
def run(kind, M, N, K, taps, T):
    if kind == "f16s":
        A = ops.cast_f16s(torch.randn(M, K, device=dev) * 0.5, K); W = ops.cast_f16s(torch.randn(N, taps * K, device=dev) * 0.05, taps * K, scale=2.0 ** 14)
    else:
        A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16); W = (torch.randn(N, taps * K, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    kw = dict(bias=bias, lda=K, ldw=taps * K)
    if taps > 1:
        kw.update(taps=taps, dil=3, pad=3 * (taps // 2), t_in=T, t_out=T)
    out = None
    for _ in range(5):
        out = ops.gemm(A, W, M, N, K, out_dtype=torch.float32, **kw)
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gemm(A, W, M, N, K, out_dtype=torch.float32, **kw)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 100)
    return statistics.median(ts), float(out.abs().sum())
for kind in ("f16s", "bf16"):
    for name, (M, N, K, taps, T) in {"sampler k7 conv (32 x 125)": (4000, 512, 512, 7, 125), "same as plain GEMM": (4000, 512, 3584, 1, 125),
                                     "B=8 out-proj": (4000, 768, 768, 1, 500), "B=8 fc2": (4000, 768, 3072, 1, 500)}.items():
        us, chk = run(kind, M, N, K, taps, T)
        print(f"{kind:5s} {name:28s} {us:7.1f} us  {2.0 * M * N * K * taps / us / 1e6:7.1f} TFLOP/s  checksum {chk:.6e}", flush=True)
