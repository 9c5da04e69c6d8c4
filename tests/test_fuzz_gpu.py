"""Seeded random shapes through the C-ABI against float64 references: the geometry chooser (64 / 128 / 192 / 256-row tiles,
plain / implicit-conv staging), both epilogue paths (transposed / direct; N tails, leading dimensions wider than N, views that
are only 4-byte aligned), every epilogue input, ragged attention lengths (padded and packed), snake strips."""
import math
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ops():
    from simwhisper_codec_amd import ops
    return ops


def _rel(a, b):
    a = a.double().cpu(); b = b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("seed", range(24))
def test_gemm_random_shapes(seed):
    ops = _ops()
    g = torch.Generator().manual_seed(1000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    dtype = [torch.float32, torch.bfloat16, torch.bfloat16][seed % 3]
    M = [ri(1, 300), ri(300, 3000), ri(3000, 9000), 16 * ri(100, 600)][seed % 4]
    N = [ri(1, 100), 64 * ri(1, 12), 8 * ri(8, 200), 256 * ri(1, 6)][(seed // 2) % 4]
    K = 8 * ri(1, 96) if seed % 5 else 64 * ri(1, 48)
    use_res, use_gamma, use_bias = seed % 2 == 0, seed % 3 == 0, seed % 4 != 1
    act = ops.ACT_GELU if seed % 3 == 1 else ops.ACT_NONE
    out_bf16 = dtype == torch.bfloat16 and seed % 4 == 3
    ldc = N + (8 * ri(0, 3) if seed % 2 else 0)               # output rows wider than N
    A = torch.randn(M, K, generator=g).to(dtype)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype)
    bias = torch.randn(N + 1, generator=g)
    gamma = torch.randn(N + 1, generator=g)
    res = torch.randn(M, ldc, generator=g)
    off = seed % 2                                              # odd seeds: bias / gamma views that are only 4-byte aligned
    bd, gd = bias.to(DEV)[off:off + N], gamma.to(DEV)[off:off + N]
    ref = A.double() @ W.double().T
    if use_bias:
        ref = ref + bias[off:off + N].double()
    if act == ops.ACT_GELU:
        ref = F.gelu(ref)
    if use_gamma:
        ref = ref * gamma[off:off + N].double()
    if use_res and not out_bf16:
        ref = ref + res[:, :N].double()
    out = torch.full((M, ldc), 7.0, device=DEV, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    ops.gemm(A.to(DEV), W.to(DEV), M, N, K, out=out, ldc=ldc, bias=bd if use_bias else None, gamma=gd if use_gamma else None,
             residual=res.to(DEV) if (use_res and not out_bf16) else None, ldr=ldc, act=act)
    tol = 8e-3 if out_bf16 else (4e-6 if dtype == torch.float32 else 4e-5)
    assert _rel(out[:, :N].float(), ref) < tol, (M, N, K, dtype, _rel(out[:, :N].float(), ref))
    if ldc > N:
        assert torch.all(out[:, N:].float() == 7.0)             # nothing is written beyond column N


@pytest.mark.parametrize("seed", range(8))
def test_attention_random_lengths(seed):
    ops = _ops()
    g = torch.Generator().manual_seed(2000 + seed)
    B, H = 1 + seed % 4, 1 + seed % 3
    T = [70, 129, 257, 500][seed % 4]
    lens = [int(torch.randint(1, T + 1, (1,), generator=g)) for _ in range(B)]
    lens[0] = T
    D = H * 64
    qkv = torch.randn(B, T, 3 * D, generator=g) * 0.7
    dt = torch.bfloat16 if seed % 2 else torch.float16
    if dt == torch.bfloat16:
        q16 = qkv.to(torch.bfloat16)
        src = q16.double()
        dev_in = q16.to(DEV)
    else:
        dev_in = ops.cast_f16s(qkv.reshape(B * T, 3 * D).to(DEV), 3 * D).view(B, T, -1)
        src = qkv.double()
    lens_d = torch.tensor(lens, dtype=torch.int32, device=DEV)
    out = ops.attention(dev_in, lens_d, B, T, H)
    if dt == torch.float16:  # split-f16 [rows, 2 D] at scale 64 -> float64
        v = out.cpu().double().view(B * T, D // 32, 2, 32)
        out = ((v[:, :, 0] + v[:, :, 1]).reshape(B, T, D)) / 64.0
    for b in range(B):
        n = lens[b]
        q, k, v = [src[b, :n, i * D:(i + 1) * D].view(n, H, 64).transpose(0, 1) for i in range(3)]
        ref = (torch.softmax(q @ k.transpose(1, 2), dim=-1) @ v).transpose(0, 1).reshape(n, D)  # the 1/sqrt(d) scale lives in the weights
        got = out[b, :n].double().cpu()
        assert float((got - ref).abs().max()) < (2e-2 if dt == torch.bfloat16 else 2e-5), (seed, b)
