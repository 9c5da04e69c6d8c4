"""Shared helpers for the test-suite (fixtures, configs, synthetic inputs)."""
import os

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def real_params():
    return yaml.safe_load(open(os.path.join(ROOT, "config", "SimWhisperCodec.yaml")))["generator_params"]


def tiny_params():
    """Must equal oracle/make_golden.py:tiny_params (the fixtures were produced with it)."""
    gp = real_params()
    gp["acoustic_encoder"].update(d_model=128, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=256)
    gp["acoustic_decoder"].update(d_model=128, decoder_layers=2, decoder_attention_heads=2, decoder_ffn_dim=256)
    gp["downsample"].update(in_dim=128, hidden_dim=64)
    gp["upsample"].update(out_dim=128, hidden_dim=64)
    gp["vocos"].update(dim=64, intermediate_dim=128, num_layers=3)
    return gp


PARAMS = {"tiny": tiny_params, "real": real_params}


def golden(tag, name):
    path = os.path.join(GOLD, f"{tag}_{name}.npz")
    return np.load(path, allow_pickle=False)


def golden_audio(g):
    from simwhisper_codec_amd import synth
    return [synth.synth_audio(int(n), index=int(i), kind=str(k))
            for k, i, n in zip(g["spec_kind"], g["spec_index"], g["spec_n"])]


_SD = {}


def state_dict(tag):
    from simwhisper_codec_amd import synth
    if tag not in _SD:
        _SD[tag] = synth.synth_state_dict(PARAMS[tag]())
    return _SD[tag]


_ORACLE = {}


def oracle(tag):
    from oracle.ref_cpu import Oracle
    if tag not in _ORACLE:
        _ORACLE[tag] = Oracle(PARAMS[tag](), state_dict(tag))
    return _ORACLE[tag]
