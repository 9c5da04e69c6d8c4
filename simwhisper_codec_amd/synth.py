"""Closed-form synthetic checkpoint in the reference's `.pt` layout.

There is no network on any box this runs on, so the trained weights
(`weights/SimWhisperCodec.pt`, README.md:147-153 of the reference) are never
available.  Every tensor here is a pure function of (tensor name, element index,
seed): a 32-bit integer hash mapped to a 24-bit uniform in [-1, 1) and scaled per
tensor — no dependence on torch's RNG streams, bit-identical on every host, cheap to
regenerate (a few seconds for the 291 M parameters of config/SimWhisperCodec.yaml).

The key set and shapes follow `AudioCodec.state_dict()` of the reference
(audiocodec/model.py:40-57 and the module constructors it calls): old-style
weight-norm pairs (`weight_g`, `weight_v`), the recomputable buffers
(`positional_embedding`, kaiser-sinc `filter`s, `istft.window`, FSQ
`dim_base_index` / `num_levels`) included, so that the same dict loads with
`strict=True` into the reference and into this package.
"""
import math
import zlib

import numpy as np
import torch

DEFAULT_SEED = 20251226


def _uniform_chunk(out, lo, hi, s):
    x = np.arange(lo, hi, dtype=np.uint64).astype(np.uint32)
    x = x * np.uint32(0x9E3779B1) + s
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x21F0AAAD)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x735A2D97)
    x ^= x >> np.uint32(15)
    out[lo:hi] = (x >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -23) - np.float32(1.0)


_POOL = None


def _uniform(name, n, seed):
    """n values in [-1, 1), element i = f(crc32(name), i, seed).  Large tensors are filled in 4 Mi-element chunks by
    a small thread pool (numpy releases the GIL inside the ufuncs); the values do not depend on the chunking."""
    global _POOL
    s = np.uint32((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFF)
    out = np.empty(n, dtype=np.float32)
    step = 1 << 22
    if n <= step:
        _uniform_chunk(out, 0, n, s)
        return out
    if _POOL is None:
        import os
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=max(1, min(8, os.cpu_count() or 1)))
    list(_POOL.map(lambda lo: _uniform_chunk(out, lo, min(n, lo + step), s), range(0, n, step)))
    return out


def _t(name, shape, amp, seed, base=0.0):
    n = int(np.prod(shape))
    v = _uniform(name, n, seed) * np.float32(amp)
    if base != 0.0:
        v = v + np.float32(base)
    return torch.from_numpy(v.reshape(shape))


def sinusoids(length, channels, max_timescale=10000):
    inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2))
    st = torch.arange(length)[:, None] * inv[None, :]
    return torch.cat([torch.sin(st), torch.cos(st)], dim=1)


def kaiser_sinc_filter(cutoff=0.25, half_width=0.3, kernel_size=12):
    """The 12-tap low-pass of the anti-aliased activation (cutoff 0.5/ratio, half width 0.6/ratio)."""
    half = kernel_size // 2
    a = 2.285 * (half - 1) * math.pi * 4 * half_width + 7.95
    if a > 50.0:
        beta = 0.1102 * (a - 8.7)
    elif a >= 21.0:
        beta = 0.5842 * (a - 21) ** 0.4 + 0.07886 * (a - 21.0)
    else:
        beta = 0.0
    win = torch.kaiser_window(kernel_size, beta=beta, periodic=False)
    t = torch.arange(-half, half) + 0.5
    f = 2 * cutoff * win * torch.sinc(2 * cutoff * t)
    return f / f.sum()


def synth_state_dict(gp, seed=DEFAULT_SEED):
    """gp: the `generator_params` dict of the YAML config. Returns {name: tensor} (fp32 / int32)."""
    sd = {}

    def lin(prefix, out_f, in_f, bias=True, k=None, gain=1.0):
        fan = in_f * (k or 1)
        shape = (out_f, in_f) if k is None else (out_f, in_f, k)
        sd[prefix + ".weight"] = _t(prefix + ".weight", shape, gain / math.sqrt(fan), seed)
        if bias:
            sd[prefix + ".bias"] = _t(prefix + ".bias", (out_f,), 1.0 / math.sqrt(fan), seed)

    def norm(prefix, c):
        sd[prefix + ".weight"] = _t(prefix + ".weight", (c,), 0.1, seed, base=1.0)
        sd[prefix + ".bias"] = _t(prefix + ".bias", (c,), 0.1, seed)

    def wn(prefix, out_f, in_f, k, g=0.6):
        sd[prefix + ".bias"] = _t(prefix + ".bias", (out_f,), 0.05, seed)
        sd[prefix + ".weight_g"] = _t(prefix + ".weight_g", (out_f, 1, 1), 0.2 * g, seed, base=g)
        sd[prefix + ".weight_v"] = _t(prefix + ".weight_v", (out_f, in_f, k), 1.0 / math.sqrt(in_f * k), seed)

    def layer(prefix, d, ffn):
        for p in ("k_proj", "v_proj", "q_proj", "out_proj"):
            lin(f"{prefix}.self_attn.{p}", d, d, bias=(p != "k_proj"))
        norm(f"{prefix}.self_attn_layer_norm", d)
        lin(f"{prefix}.fc1", ffn, d)
        lin(f"{prefix}.fc2", d, ffn)
        norm(f"{prefix}.final_layer_norm", d)

    filt = kaiser_sinc_filter().view(1, 1, 12)

    def res_blocks(prefix, hidden, dilations=(1, 3, 9)):
        for i, _ in enumerate(dilations):
            p = f"{prefix}.res_blocks.{i}.block"
            for a in (0, 2):
                sd[f"{p}.{a}.act.alpha"] = _t(f"{p}.{a}.act.alpha", (hidden,), 0.3, seed)
                sd[f"{p}.{a}.act.beta"] = _t(f"{p}.{a}.act.beta", (hidden,), 0.3, seed)
                sd[f"{p}.{a}.upsample.filter"] = filt.clone()
                sd[f"{p}.{a}.downsample.lowpass.filter"] = filt.clone()
            wn(f"{p}.1", hidden, hidden, 7)
            wn(f"{p}.3", hidden, hidden, 1)

    # --- acoustic encoder (modules.py:236-285)
    e = gp["acoustic_encoder"]
    d, k = e["d_model"], e["kernel_size"]
    max_pos = (e["max_audio_seconds"] * e["sampling_rate"] // e["hop_length"]) // e["stride_size"]
    sd["acoustic_encoder.positional_embedding"] = sinusoids(max_pos, d)
    lin("acoustic_encoder.conv1", d, e["num_mel_bins"], k=k)
    lin("acoustic_encoder.conv2", d, d, k=k)
    for i in range(e["encoder_layers"]):
        layer(f"acoustic_encoder.layers.{i}", d, e["encoder_ffn_dim"])
    norm("acoustic_encoder.layer_norm", d)

    # --- frame-stack down / up (modules.py:476-634)
    ds = gp["downsample"]
    wn("downsample.in_proj", ds["hidden_dim"], ds["in_dim"] * ds["stack_factor"], 1)
    res_blocks("downsample", ds["hidden_dim"])
    wn("downsample.to_latent", ds["latent_dim"], ds["hidden_dim"], 1, g=2.5)  # spread the FSQ levels
    q = gp["quantizer"]
    levels = q["num_levels_per_group"]
    for g in range(q["num_groups"]):
        base = torch.cumprod(torch.tensor([1] + levels[:-1]), dim=0).to(torch.int32).view(1, -1, 1)
        sd[f"quantizer.fsqs.{g}.dim_base_index"] = base
        sd[f"quantizer.fsqs.{g}.num_levels"] = torch.tensor(levels, dtype=torch.int32).view(1, -1, 1)
    us = gp["upsample"]
    wn("upsample.from_latent", us["hidden_dim"], us["latent_dim"], 1)
    res_blocks("upsample", us["hidden_dim"])
    wn("upsample.to_stacked", us["out_dim"] * us["stack_factor"], us["hidden_dim"], 1)

    # --- acoustic decoder (modules.py:380-435)
    dc = gp["acoustic_decoder"]
    dd, dk = dc["d_model"], dc["kernel_size"]
    dmax = (dc["max_audio_seconds"] * dc["sampling_rate"] // dc["hop_length"]) // dc["stride_size"]
    sd["acoustic_decoder.positional_embedding"] = sinusoids(dmax, dd)
    # ConvTranspose1d weights are (in, out, k)
    sd["acoustic_decoder.deconv1.weight"] = _t("acoustic_decoder.deconv1.weight", (dd, dd, dk), 1 / math.sqrt(dd * dk), seed)
    sd["acoustic_decoder.deconv1.bias"] = _t("acoustic_decoder.deconv1.bias", (dd,), 1 / math.sqrt(dd * dk), seed)
    sd["acoustic_decoder.deconv2.weight"] = _t("acoustic_decoder.deconv2.weight", (dd, dc["num_mel_bins"], dk),
                                               1 / math.sqrt(dd * dk), seed)
    sd["acoustic_decoder.deconv2.bias"] = _t("acoustic_decoder.deconv2.bias", (dc["num_mel_bins"],), 0.05, seed)
    for i in range(dc["decoder_layers"]):
        layer(f"acoustic_decoder.layers.{i}", dd, dc["decoder_ffn_dim"])
    norm("acoustic_decoder.layer_norm", dd)

    # --- Vocos (modules.py:1441-1573, 1190-1248, 1033-1082)
    v = gp["vocos"]
    dim, inter, nl = v["dim"], v["intermediate_dim"], v["num_layers"]
    lin("vocos.backbone.embed", dim, v["input_channels"], k=7)
    norm("vocos.backbone.norm", dim)
    for i in range(nl):
        p = f"vocos.backbone.convnext.{i}"
        sd[p + ".gamma"] = _t(p + ".gamma", (dim,), 0.5 / nl, seed, base=1.0 / nl)
        sd[p + ".dwconv.weight"] = _t(p + ".dwconv.weight", (dim, 1, 7), 1 / math.sqrt(7), seed)
        sd[p + ".dwconv.bias"] = _t(p + ".dwconv.bias", (dim,), 0.05, seed)
        norm(p + ".norm", dim)
        lin(p + ".pwconv1", inter, dim)
        lin(p + ".pwconv2", dim, inter)
    norm("vocos.backbone.final_layer_norm", dim)
    lin("vocos.head.out", v["n_fft"] + 2, dim, gain=0.5)
    sd["vocos.head.istft.window"] = torch.hann_window(v["n_fft"])
    return sd


def synth_audio(n_samples, index=0, seed=1234, kind="noise"):
    """Synthetic 16 kHz mono test signals, closed form (no RNG state).

    noise : 0.1 * sqrt(3) * U(-1, 1)  (std 0.1, the survey's white-noise level)
    speech: five harmonics of a gliding f0 under a slow envelope, plus a little noise.
    """
    if kind == "zero":
        return torch.zeros(n_samples)
    u = _uniform(f"audio/{kind}/{index}", n_samples, seed)
    if kind == "noise":
        return torch.from_numpy(u * np.float32(0.1 * math.sqrt(3.0)))
    t = np.arange(n_samples, dtype=np.float64) / 16000.0
    f0 = 110.0 + 40.0 * index + 30.0 * np.sin(2 * math.pi * 0.7 * t)
    ph = 2 * math.pi * np.cumsum(f0) / 16000.0
    x = sum(np.sin(h * ph) / h for h in range(1, 6))
    env = 0.5 * (1 + np.sin(2 * math.pi * 1.3 * t + index)) ** 2
    return torch.from_numpy((0.08 * env * x).astype(np.float32) + u * np.float32(0.003))
