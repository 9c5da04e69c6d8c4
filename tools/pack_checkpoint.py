#!/usr/bin/env python3
"""Checkpoint conversion: validate a reference `.pt` against the 711-key manifest of the YAML config and
write it as safetensors (no pickle: loads with zero code execution and memory-maps), in one of two forms:

  plain (default)   the same 711 tensors; `load_from_checkpoint` then folds / scales / casts them at first use as for a `.pt`
  --fold            GEMM-READY OPERANDS of one precision preset (needs the GPU: the packing kernels run once, here):
                    weight norm folded, conv kernels as [Cout][tap][Cin], frame-stack columns re-ordered, per-tensor
                    power-of-two scales, bf16 / split-f16 / e4m3 casts, the fused ConvNeXt operand streams, DFT / mel /
                    inverse-DFT tables.  `load_from_checkpoint` maps this file straight to the device: no 12 s module
                    construction, no fold / cast pass, no host synchronisation per tensor.  The file is bound to its
                    precision preset, configuration and library ABI (checked at load).

    python tools/pack_checkpoint.py --config config/SimWhisperCodec.yaml --in weights/SimWhisperCodec.pt --out weights/SimWhisperCodec.safetensors
    python tools/pack_checkpoint.py --config ... --in ... --fold --precision mixed --out weights/SimWhisperCodec.mixed.safetensors
    python tools/pack_checkpoint.py --config ... --synthetic --out /tmp/synth.safetensors

`AudioCodec.load_from_checkpoint` accepts `.pt` and both safetensors forms."""
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
    ap.add_argument("--fold", action="store_true", help="write GEMM-ready operands (needs the GPU)")
    ap.add_argument("--precision", default="mixed", choices=["fp32", "mixed", "mixed_f32", "bf16", "fp8", "fp8_fc1", "f16s"])
    ap.add_argument("--device", default="cuda:0")
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
    if args.fold:
        from simwhisper_codec_amd.codec import AudioCodec
        m = AudioCodec(gp, precision=args.precision)
        m.load_state_dict(sd, strict=True)
        m = m.to(args.device).eval()
        with torch.cuda.device(torch.device(args.device)):
            nt, nbytes = m.export_packed(args.out)
        print(f"wrote {args.out}: {nt} operand tensors, {nbytes / 1e6:.1f} MB, precision {args.precision}")
        return
    from safetensors.torch import save_file
    save_file({k: v.contiguous() for k, v in sd.items()}, args.out, metadata={"format": "simwhisper-codec state_dict"})
    n = sum(v.numel() for v in sd.values())
    print(f"wrote {args.out}: {len(sd)} tensors, {n / 1e6:.2f} M values")


if __name__ == "__main__":
    main()
