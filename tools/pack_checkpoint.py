#!/usr/bin/env python3
"""Checkpoint conversion: validate a reference `.pt` against the 711-key manifest of the
YAML config and write it back as a flat safetensors file (no pickle: loads with zero code execution and
memory-maps), optionally with weight norm already folded.

    python tools/pack_checkpoint.py --config config/SimWhisperCodec.yaml --in weights/SimWhisperCodec.pt --out weights/SimWhisperCodec.safetensors
    python tools/pack_checkpoint.py --config ... --synthetic --out /tmp/synth.safetensors

`AudioCodec.load_from_checkpoint` accepts both `.pt` and `.safetensors` (same keys, strict)."""
import argparse
import os
import sys

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simwhisper_codec_amd import spec, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=os.path.join(ROOT, "config", "SimWhisperCodec.yaml"))
    ap.add_argument("--in", dest="inp", default=None)
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    gp = yaml.safe_load(open(args.config))["generator_params"]
    if args.synthetic:
        sd = synth.synth_state_dict(gp)
    else:
        ck = torch.load(args.inp, map_location="cpu", weights_only=True)
        sd = ck["model"] if "model" in ck else ck
    want = spec.state_shapes(gp)
    missing, extra = sorted(set(want) - set(sd)), sorted(set(sd) - set(want))
    if missing or extra:
        raise SystemExit(f"key mismatch: missing {missing[:5]} ({len(missing)}), unexpected {extra[:5]} ({len(extra)})")
    for k, (shape, dtype) in want.items():
        if tuple(sd[k].shape) != shape:
            raise SystemExit(f"{k}: shape {tuple(sd[k].shape)} != {shape}")
    from safetensors.torch import save_file
    save_file({k: v.contiguous() for k, v in sd.items()}, args.out, metadata={"format": "simwhisper-codec state_dict"})
    n = sum(v.numel() for v in sd.values())
    print(f"wrote {args.out}: {len(sd)} tensors, {n / 1e6:.2f} M values")


if __name__ == "__main__":
    main()
