#!/usr/bin/env python3
"""Where the waves of swc_gemm's K loop spend their cycles (diagnostic build: tools/build_variant.sh stamp -DSWC_GEMM_STAMP,
run with SWC_LIB=.../libswc_stamp.so).  Per slice: cycles between fences (matrix work + fragment reads + DMA issue), in the
`s_waitcnt vmcnt(0)` of the fence, in its barrier; per tile: epilogue cycles.  s_memtime ticks = shader cycles."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from simwhisper_codec_amd import ops, _lib

SHAPES = {"qkv": (16000, 2304, 768, 0, 1), "out_proj": (16000, 768, 768, 0, 0), "fc1": (16000, 3072, 768, 1, 1),
          "fc2": (16000, 768, 3072, 0, 0), "pw1": (32000, 4096, 512, 1, 1), "pw2": (32000, 512, 4096, 0, 0),
          # the k7 convolutions of the samplers at the metric's batch, as a plain GEMM of the same M, N and K (4-wave 64 x 128 tiles)
          "samp_k7": (4000, 512, 3584, 0, 0)}

def main():
    lib = _lib.load()
    fn = lib.swc_debug_stamps
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]; fn.restype = ctypes.c_int
    buf = np.zeros(512 * 8 * 8, dtype=np.uint64)
    only = [a[5:] for a in sys.argv[1:] if a.startswith("only=")]
    kinds = [a for a in sys.argv[1:] if not a.startswith(("M=", "only="))] or ["bf16", "f16s"]
    mscale = [float(a[2:]) for a in sys.argv[1:] if a.startswith("M=")] or [1.0]   # M=0.25: a quarter of the rows (fewer workgroups)
    for kind in kinds:
      for ms in mscale:
        for name, (M, N, K, gelu, obf) in SHAPES.items():
            if only and name not in only:
                continue
            M = int(M * ms)
            dev = "cuda"
            if kind == "f16s":
                A = ops.cast_f16s(torch.randn(M, K, device=dev) * 0.5, K)
                W = ops.cast_f16s(torch.randn(N, K, device=dev) * 0.05, K, scale=2.0 ** 14)
                odt = torch.float16 if obf else torch.float32
            else:
                A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
                W = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
                odt = torch.bfloat16 if obf else torch.float32
            bias = torch.randn(N, device=dev)
            out = torch.empty(M, N * (2 if odt == torch.float16 else 1), device=dev, dtype=odt)
            res = None if obf else torch.randn(M, N, device=dev)
            kw = dict(bias=bias, out=out, ldc=N, act=ops.ACT_GELU if gelu else ops.ACT_NONE)
            if res is not None:
                kw.update(residual=res, ldr=N)
            for _ in range(5):
                ops.gemm(A, W, M, N, K, **kw)
            torch.cuda.synchronize()
            fn(None, 1)
            R = 10
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(R):
                ops.gemm(A, W, M, N, K, **kw)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / R
            fn(buf.ctypes.data, 0)
            s = buf.reshape(512, 8, 8).astype(np.float64)
            act = s[:, :, 3] > 0
            n = s[:, :, 3][act]
            cmp_, vm, bar = s[:, :, 0][act] / n, s[:, :, 1][act] / n, s[:, :, 2][act] / n
            tiles = s[:, :, 5][act]
            epi = s[:, :, 4][act] / tiles
            tail = s[:, :, 6][act] / tiles
            print(f"{kind:5s} {name:9s} {us:7.1f} us {2.0*M*N*K/us/1e6:7.1f} TF | per slice: work {cmp_.mean():7.0f}  vmcnt-wait {vm.mean():6.0f} (max wave {vm.max():6.0f})"
                  f"  barrier-wait {bar.mean():6.0f} | per tile: {n.mean()/ (tiles.mean()):5.1f} slices, last half + handoff {tail.mean():6.0f}, epilogue {epi.mean():7.0f} cycles, tiles/wg {tiles.mean()/R:4.1f}", flush=True)

if __name__ == "__main__":
    main()
