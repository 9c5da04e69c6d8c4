"""How the sustained MFMA rate depends on how many CUs are busy: swc_proj_ln built with every memory operation and the epilogue
compiled out (tools/build_src_variant.sh pl_a29 swc_projln.hip -DPL_ABL=29: 144 v_mfma_f32_32x32x16_f16 per 64-k stage and wave on
register operands + 16 LDS reads, one workgroup of 4 waves per CU), K = 3072, for grids of 16 ... 250 workgroups.  Wrong results by
construction; timing only.   usage: SWC_LIB=simwhisper_codec_amd/libswc_pl_a29.so python tools/probes/mfma_ceiling_vs_grid.py"""
import os, statistics, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from simwhisper_codec_amd import ops

N, K = 768, 3072
dev = "cuda"
W = ops.cast_f16s(torch.randn(N, K, device=dev) * K ** -0.5, K, scale=2.0 ** 12)
st = ops.proj_ln_pack(W)
for wgs in (16, 32, 64, 128, 192, 250, 256, 500):
    M = 64 * wgs
    A = ops.cast_f16s(torch.randn(M, K, device=dev) * 0.5, K, scale=64.0)
    x = torch.zeros(M, N, device=dev)
    ts = []
    for rnd in range(5):
        for _ in range(3):
            ops.proj_ln(A, st, None, 1.0, x, M=M, N=N, K=K)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.proj_ln(A, st, None, 1.0, x, M=M, N=N, K=K)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20)
    t = statistics.median(ts) * 1e-3
    fl = 2.0 * M * N * K * 3
    print(f"{wgs:4d} workgroups  {t*1e6:7.1f} us per launch   {fl/t/1e12:7.1f} TFLOP/s executed   {fl/t/1e12/min(wgs,256):6.2f} per busy CU", flush=True)
