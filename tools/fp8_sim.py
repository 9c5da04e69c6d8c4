#!/usr/bin/env python3
"""CPU simulation of the fp8 encoder-linear preset (BASELINE.json configs[4]): how many FSQ levels survive when the 48
encoder-transformer linears (qkv, out, fc1, fc2) quantise their operands to OCP e4m3 with
  tensor : one power-of-two scale per tensor (activations x 16, weights to [224, 448)) — what the kernels do today
  block  : one E8M0 scale per 32-element block along K (the MX format the block-scaled MFMA consumes), activations and weights
  bf16   : bf16 operands (the `bf16` preset's class), for comparison
against the float32 oracle on the inputs of tests/test_parity_gpu.py::test_reduced_precision_encoder_levels.
Products are accumulated in float32 (as the MFMA does).  Uses oracle/ref_cpu.py: a measurement tool, not product code."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F
from common import PARAMS, state_dict
from oracle.ref_cpu import Oracle
from simwhisper_codec_amd import synth

F8 = torch.float8_e4m3fn


def q_tensor(x, scale):
    return (x * scale).clamp(-448, 448).to(F8).float() / scale


def q_block(x):
    """per-32 block along the last dim: shared exponent floor(log2 max) - 8 (OCP MX: e4m3 emax = 8), saturating cast"""
    sh = x.shape
    b = x.reshape(-1, sh[-1] // 32, 32)
    mx = b.abs().amax(-1, keepdim=True).clamp_min(2.0 ** -120)
    e = torch.floor(torch.log2(mx)) - 8
    s = torch.exp2(e)
    return ((b / s).clamp(-448, 448).to(F8).float() * s).reshape(sh)


class Sim(Oracle):
    mode = "tensor"

    which = None  # names of the linears that are fp8 (None: all four); the others are bf16

    def _lin(self, x, w, b=None, name=None):
        if self.which is not None and name not in self.which:
            return F.linear(x.bfloat16().float(), w.bfloat16().float(), b)
        if self.mode == "bf16":
            return F.linear(x.bfloat16().float(), w.bfloat16().float(), b)
        if self.mode == "tensor":
            mx = float(w.abs().max())
            import math
            sw = 2.0 ** math.floor(math.log2(448.0 / mx))
            return F.linear(q_tensor(x, 16.0), q_tensor(w, sw), b)
        if self.mode == "block":
            return F.linear(q_block(x), q_block(w), b)
        if self.mode == "block_act":   # block scales on the activations only, weights per tensor
            mx = float(w.abs().max())
            import math
            sw = 2.0 ** math.floor(math.log2(448.0 / mx))
            return F.linear(q_block(x), q_tensor(w, sw), b)
        return F.linear(x, w, b)

    def _layer(self, h, lens, p, heads):
        sd = self.sd
        if not p.startswith("acoustic_encoder"):
            return super()._layer(h, lens, p, heads)
        B, T, D = h.shape
        hd = D // heads
        x = F.layer_norm(h, (D,), sd[p + "self_attn_layer_norm.weight"], sd[p + "self_attn_layer_norm.bias"], 1e-5)
        q = self._lin(x, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"], "qkv") * hd ** -0.5
        k = self._lin(x, sd[p + "self_attn.k_proj.weight"], None, "qkv")
        v = self._lin(x, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"], "qkv")
        q, k, v = [t.view(B, T, heads, hd).transpose(1, 2) for t in (q, k, v)]
        s = q @ k.transpose(-1, -2)
        ok = torch.arange(T)[None, :] < lens[:, None]
        both = (ok[:, None, :, None] & ok[:, None, None, :]).to(s.dtype)
        s = s + (both + (1.0 - both) * torch.finfo(s.dtype).min)
        a = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, T, D)
        h = h + self._lin(a, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"], "out")
        x = F.layer_norm(h, (D,), sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"], 1e-5)
        x = F.gelu(self._lin(x, sd[p + "fc1.weight"], sd[p + "fc1.bias"], "fc1"))
        return h + self._lin(x, sd[p + "fc2.weight"], sd[p + "fc2.bias"], "fc2")


def main():
    tag = "real"
    torch.set_num_threads(8)
    wavs = [synth.synth_audio(80000 - 331 * i, index=700 + i, kind="speech" if i % 2 else "noise") for i in range(8)]
    gp, sd = PARAMS[tag](), state_dict(tag)
    want = Oracle(gp, sd).encode(wavs, trim=True)["codes_list"]
    base, lev = torch.tensor([1, 8, 56, 336]), torch.tensor([8, 7, 6, 6])
    for mode in sys.argv[1:] or ["bf16", "tensor", "block_act", "block"]:
        o = Sim(gp, sd)
        if ":" in mode:   # e.g. tensor:fc1+fc2 — only these linears in fp8, the others bf16
            mode, names = mode.split(":")
            o.which = set(names.split("+"))
            mode_label = f"{mode}:{names}"
        else:
            mode_label = mode
        o.mode = mode
        got = o.encode(wavs, trim=True)["codes_list"]
        same = w1 = tot = cs = ct = 0
        for a, b in zip(got, want):
            la, lb = (a.long()[..., None] // base) % lev, (b.long()[..., None] // base) % lev
            d = (la - lb).abs()
            same += int((d == 0).sum()); w1 += int((d <= 1).sum()); tot += d.numel()
            cs += int((a == b).sum()); ct += a.numel()
        print(f"{mode_label:22s} levels equal {same / tot:.4f}  within one {w1 / tot:.5f}  codes equal {cs / ct:.4f}", flush=True)


if __name__ == "__main__":
    main()
