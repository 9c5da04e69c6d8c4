"""The three f32-output bf16 GEMMs of a B = 8 x 10 s step (39 % of its device time) under every tile geometry swc_gemm has, on a tuning
build (tools/build_variant.sh tune; SWC_LIB=.../libswc_tune.so): out-proj (4000 x 768 x 768), fc2 (4000 x 768 x 3072), pwconv2
(8000 x 512 x 4096), residual in place as in the pipeline.  One process per setting (the overrides are read per call, but keep it simple)."""
import os, statistics, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from simwhisper_codec_amd import ops
    dev = "cuda"
    for name, (M, N, K) in {"out-proj": (4000, 768, 768), "fc2": (4000, 768, 3072), "pwconv2": (8000, 512, 4096)}.items():
        A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
        bias = torch.randn(N, device=dev); xs = [torch.randn(M, N, device=dev) for _ in range(8)]
        for i in range(5):
            ops.gemm(A, W, M, N, K, bias=bias, residual=xs[i % 8], out=xs[i % 8])
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(16):
                ops.gemm(A, W, M, N, K, bias=bias, residual=xs[i % 8], out=xs[i % 8])
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 16 * 1e3)
        print(f"   {name:9s} {statistics.median(ts):7.1f} us  {2.0 * M * N * K / statistics.median(ts) / 1e6:7.1f} TFLOP/s", flush=True)
    sys.exit(0)
for label, env in (("shipped (64 x 128)", {}), ("128 x 128", {"SWC_GEMM_SMALL": "0"}), ("128 x 256 (8 waves)", {"SWC_GEMM_TILE": "256", "SWC_GEMM_MT": "4"}),
                   ("192 x 256", {"SWC_GEMM_TILE": "256", "SWC_GEMM_MT": "6"}), ("256 x 256", {"SWC_GEMM_TILE": "256", "SWC_GEMM_MT": "8"})):
    print(label, flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env={**os.environ, **env}, check=True)
