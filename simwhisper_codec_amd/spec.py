"""Shapes, length laws and constant tables of the hot path (host side, load time).

Everything here is derived from `generator_params` (the reference's YAML schema,
config/SimWhisperCodec.yaml) or is a closed-form table; citations are to the reference
files the numbers come from.
"""
import math

import numpy as np
import torch

N_FFT = 400           # feature_extractor.py:28 (n_fft), hann window, hop 160
HOP = 160
CHUNK_SAMPLES = 480000  # 30 s at 16 kHz: every tokenize call is padded to this (feature_extractor.py:207-214)
MEL_FRAMES = 3000
N_MELS = 80
HEAD_DIM = 64         # the attention kernel's head size (768/12; the tiny test config keeps it)


def cdiv(a, b):
    return -(-a // b)


def mel_len(n):
    """attention_mask[:, ::160].sum() (feature_extractor.py:237, model.py:191)."""
    return cdiv(min(n, CHUNK_SAMPLES), HOP)


def token_len(n):
    """OmniAudioEncoder: mel_len // stride (modules.py:322)."""
    return mel_len(n) // 2


def latent_len(n, stack=4):
    """FrameStackDownConv: ceil(len / stack) (modules.py:533)."""
    return cdiv(token_len(n), stack)


def state_shapes(gp):
    """name -> (shape, dtype) of AudioCodec.state_dict() in the reference (model.py:40-57)."""
    out = {}

    def add(name, *shape, dtype=torch.float32):
        out[name] = (tuple(shape), dtype)

    def layer(p, d, ffn):
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            add(f"{p}.self_attn.{n}.weight", d, d)
            if n != "k_proj":
                add(f"{p}.self_attn.{n}.bias", d)
        for n in ("self_attn_layer_norm", "final_layer_norm"):
            add(f"{p}.{n}.weight", d); add(f"{p}.{n}.bias", d)
        add(f"{p}.fc1.weight", ffn, d); add(f"{p}.fc1.bias", ffn)
        add(f"{p}.fc2.weight", d, ffn); add(f"{p}.fc2.bias", d)

    def wn(p, o, i, k):
        add(p + ".bias", o); add(p + ".weight_g", o, 1, 1); add(p + ".weight_v", o, i, k)

    def res(p, h):
        for i in range(3):
            b = f"{p}.res_blocks.{i}.block"
            for a in (0, 2):
                add(f"{b}.{a}.act.alpha", h); add(f"{b}.{a}.act.beta", h)
                add(f"{b}.{a}.upsample.filter", 1, 1, 12); add(f"{b}.{a}.downsample.lowpass.filter", 1, 1, 12)
            wn(f"{b}.1", h, h, 7); wn(f"{b}.3", h, h, 1)

    e = gp["acoustic_encoder"]
    d, k = e["d_model"], e["kernel_size"]
    pos = (e["max_audio_seconds"] * e["sampling_rate"] // e["hop_length"]) // e["stride_size"]
    add("acoustic_encoder.positional_embedding", pos, d)
    add("acoustic_encoder.conv1.weight", d, e["num_mel_bins"], k); add("acoustic_encoder.conv1.bias", d)
    add("acoustic_encoder.conv2.weight", d, d, k); add("acoustic_encoder.conv2.bias", d)
    for i in range(e["encoder_layers"]):
        layer(f"acoustic_encoder.layers.{i}", d, e["encoder_ffn_dim"])
    add("acoustic_encoder.layer_norm.weight", d); add("acoustic_encoder.layer_norm.bias", d)

    ds = gp["downsample"]
    wn("downsample.in_proj", ds["hidden_dim"], ds["in_dim"] * ds["stack_factor"], 1)
    res("downsample", ds["hidden_dim"])
    wn("downsample.to_latent", ds["latent_dim"], ds["hidden_dim"], 1)
    q = gp["quantizer"]
    nd = len(q["num_levels_per_group"])
    for g in range(q["num_groups"]):
        add(f"quantizer.fsqs.{g}.dim_base_index", 1, nd, 1, dtype=torch.int32)
        add(f"quantizer.fsqs.{g}.num_levels", 1, nd, 1, dtype=torch.int32)
    us = gp["upsample"]
    wn("upsample.from_latent", us["hidden_dim"], us["latent_dim"], 1)
    res("upsample", us["hidden_dim"])
    wn("upsample.to_stacked", us["out_dim"] * us["stack_factor"], us["hidden_dim"], 1)

    dc = gp["acoustic_decoder"]
    dd, dk = dc["d_model"], dc["kernel_size"]
    dpos = (dc["max_audio_seconds"] * dc["sampling_rate"] // dc["hop_length"]) // dc["stride_size"]
    add("acoustic_decoder.positional_embedding", dpos, dd)
    add("acoustic_decoder.deconv1.weight", dd, dd, dk); add("acoustic_decoder.deconv1.bias", dd)
    add("acoustic_decoder.deconv2.weight", dd, dc["num_mel_bins"], dk); add("acoustic_decoder.deconv2.bias", dc["num_mel_bins"])
    for i in range(dc["decoder_layers"]):
        layer(f"acoustic_decoder.layers.{i}", dd, dc["decoder_ffn_dim"])
    add("acoustic_decoder.layer_norm.weight", dd); add("acoustic_decoder.layer_norm.bias", dd)

    v = gp["vocos"]
    dim, inter = v["dim"], v["intermediate_dim"]
    add("vocos.backbone.embed.weight", dim, v["input_channels"], 7); add("vocos.backbone.embed.bias", dim)
    add("vocos.backbone.norm.weight", dim); add("vocos.backbone.norm.bias", dim)
    for i in range(v["num_layers"]):
        p = f"vocos.backbone.convnext.{i}"
        add(p + ".gamma", dim)
        add(p + ".dwconv.weight", dim, 1, 7); add(p + ".dwconv.bias", dim)
        add(p + ".norm.weight", dim); add(p + ".norm.bias", dim)
        add(p + ".pwconv1.weight", inter, dim); add(p + ".pwconv1.bias", inter)
        add(p + ".pwconv2.weight", dim, inter); add(p + ".pwconv2.bias", dim)
    add("vocos.backbone.final_layer_norm.weight", dim); add("vocos.backbone.final_layer_norm.bias", dim)
    add("vocos.head.out.weight", v["n_fft"] + 2, dim); add("vocos.head.out.bias", v["n_fft"] + 2)
    add("vocos.head.istft.window", v["n_fft"])
    return out


def slaney_mel_filters(n_freq=201, n_mels=N_MELS, fmin=0.0, fmax=8000.0, sr=16000):
    """(n_freq, n_mels) float64 slaney-scale, slaney-normalised triangular bank: what
    feature_extractor.py:50-58 obtains from transformers.audio_utils.mel_filter_bank."""
    lin_step = 200.0 / 3.0
    log_step = math.log(6.4) / 27.0

    def to_mel(f):
        return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1.0) / 1000.0) / log_step, f / lin_step)

    def to_hz(m):
        return np.where(m >= 15.0, 1000.0 * np.exp(log_step * (m - 15.0)), lin_step * m)

    edges = to_hz(np.linspace(to_mel(np.float64(fmin)), to_mel(np.float64(fmax)), n_mels + 2))
    bins = np.linspace(0, sr // 2, n_freq)
    width = np.diff(edges)
    rel = edges[None, :] - bins[:, None]
    tri = np.maximum(0.0, np.minimum(-rel[:, :-2] / width[:-1], rel[:, 2:] / width[1:]))
    return tri * (2.0 / (edges[2:] - edges[:-2]))[None, :]


def dft_basis_400():
    """[402][400] f32: rows 0..200 hann*cos, rows 201..401 hann*sin — the windowed real DFT
    behind torch.stft(n_fft=400, window=hann_window(400)) (feature_extractor.py:94-99)."""
    n = np.arange(N_FFT, dtype=np.float64)
    k = np.arange(N_FFT // 2 + 1, dtype=np.float64)
    win = 0.5 - 0.5 * np.cos(2 * np.pi * n / N_FFT)  # periodic hann
    ang = 2 * np.pi * np.outer(k, n) / N_FFT
    return torch.from_numpy(np.concatenate([np.cos(ang) * win, np.sin(ang) * win], 0).astype(np.float32))


def idft_basis(n_fft, window, ld):
    """[n_fft][ld] f32 such that frames = spec_row @ basis^T equals
    irfft(S, n_fft, norm="backward") * window (modules.py:861-862); spec_row = (Re 0..n/2 | Im 0..n/2 | 0)."""
    nb = n_fft // 2 + 1
    n = np.arange(n_fft, dtype=np.float64)
    k = np.arange(nb, dtype=np.float64)
    c = np.full(nb, 2.0); c[0] = 1.0; c[-1] = 1.0
    ang = 2 * np.pi * np.outer(n, k) / n_fft
    w = window.double().cpu().numpy()[:, None]
    out = np.zeros((n_fft, ld), dtype=np.float64)
    out[:, :nb] = np.cos(ang) * c / n_fft * w
    out[:, nb:2 * nb] = -np.sin(ang) * c / n_fft * w
    return torch.from_numpy(out.astype(np.float32))


def fsq_constants(levels, eps):
    """scale | offset | shift exactly as FiniteScalarQuantizer.compress derives them in fp32
    (quantizer.py:131-137)."""
    lv = torch.tensor(levels, dtype=torch.int32)
    scale = (lv - 1) / 2
    scale = scale * (1 - eps)
    offset = torch.where(lv % 2 == 0, 0.5, 0)
    shift = (offset / scale).tan()
    return torch.cat([scale, offset, shift]).tolist()
