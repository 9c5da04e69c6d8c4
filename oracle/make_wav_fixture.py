#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (not on the product path).  Pins the RIFF/WAVE reader of simwhisper_codec_amd/wavio.py against the
only audio data the reference ships: docs/assets/codec/*.wav (24 kHz originals `gt_*` and the codecs' outputs, among them
the reference's own 16 kHz round trips `simwhisper_*`).  Each file is read here with the standard library's `wave` module
— an independent reader — and reduced to a small record: sample rate, channels, sample width, frame count, CRC32 of the
PCM bytes and the first 256 samples.  The records are data (inputs and expected values), not reference source.

    python oracle/make_wav_fixture.py [--ref /root/reference] [--out tests/golden/ref_wav_assets.json]

Runs in the build container only (the reference does not travel to the GPU box); the JSON is committed.
"""
import argparse
import glob
import json
import os
import struct
import wave
import zlib

ap = argparse.ArgumentParser()
ap.add_argument("--ref", default="/root/reference")
ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                              "ref_wav_assets.json"))
args = ap.parse_args()
rec = {}
for path in sorted(glob.glob(os.path.join(args.ref, "docs", "assets", "codec", "*.wav"))):
    with wave.open(path, "rb") as w:
        ch, width, sr, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        pcm = w.readframes(n)
    assert width == 2, (path, width)
    head = struct.unpack(f"<{min(256, len(pcm) // 2)}h", pcm[: 2 * min(256, len(pcm) // 2)])
    rec[os.path.basename(path)] = {"rate": sr, "channels": ch, "sample_width": width, "frames": n,
                                   "crc32_pcm": zlib.crc32(pcm) & 0xFFFFFFFF, "first_samples": list(head),
                                   "file_bytes": os.path.getsize(path)}
with open(args.out, "w") as f:
    json.dump(rec, f, indent=0, sort_keys=True)
print(f"{len(rec)} files -> {args.out}")
