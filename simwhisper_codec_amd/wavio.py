"""Audio file IO around the hot path (the role of utils/helpers.py:77-111 in the reference).

torchaudio is not available offline, so this is a small self-contained RIFF/WAVE reader
and writer (stdlib + numpy): PCM 8/16/24/32-bit and IEEE float 32/64 in, PCM16 out.
Conventions the reference leaves to torchaudio and which are therefore OUR choice:
  * multi-channel input is averaged to mono (helpers.py:82-83 does the same);
  * sample-rate conversion: polyphase Kaiser-windowed sinc (scipy.signal.resample_poly);
  * float -> PCM16: round(clip(x, -1, 1) * 32767).
`.flac` / `.mp3` are listed (helpers.py:106) but cannot be decoded here: a clear error is raised.
"""
import glob
import logging
import os
import struct
from math import gcd

import numpy as np
import torch

AUDIO_EXTENSIONS = ("*.flac", "*.mp3", "*.wav")


def find_audio_files(input_dir):
    """Recursive, sorted list of audio files (helpers.py:105-111)."""
    found = []
    for ext in AUDIO_EXTENSIONS:
        found.extend(glob.glob(os.path.join(input_dir, "**", ext), recursive=True))
    logging.info(f"Found {len(found)} audio files in {input_dir}")
    return sorted(found)


def _read_wav(path):
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            tag, ch, sr, _, _, bits = struct.unpack("<HHIIHH", body[:16])
            if tag == 0xFFFE and len(body) >= 26:  # WAVE_FORMAT_EXTENSIBLE: real tag in the sub-format GUID
                tag = struct.unpack("<H", body[24:26])[0]
            fmt = (tag, ch, sr, bits)
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError(f"{path}: missing fmt or data chunk")
    tag, ch, sr, bits = fmt
    if tag == 1:
        if bits == 8:
            x = (np.frombuffer(pcm, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
        elif bits == 16:
            x = np.frombuffer(pcm[: len(pcm) // 2 * 2], dtype="<i2").astype(np.float32) / 32768.0
        elif bits == 24:
            b = np.frombuffer(pcm[: len(pcm) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            x = ((v ^ 0x800000) - 0x800000).astype(np.float32) / 8388608.0
        elif bits == 32:
            x = np.frombuffer(pcm[: len(pcm) // 4 * 4], dtype="<i4").astype(np.float32) / 2147483648.0
        else:
            raise ValueError(f"{path}: unsupported PCM width {bits}")
    elif tag == 3:
        x = np.frombuffer(pcm, dtype="<f4" if bits == 32 else "<f8").astype(np.float32)
    else:
        raise ValueError(f"{path}: unsupported WAVE format tag {tag}")
    n = len(x) // ch
    return x[: n * ch].reshape(n, ch), sr


def load_audio(audio_path, target_sample_rate):
    """-> FloatTensor (1, 1, T) at target_sample_rate, mono (helpers.py:77-94)."""
    ext = os.path.splitext(audio_path)[1].lower()
    if ext != ".wav":
        raise RuntimeError(f"{audio_path}: only .wav can be decoded offline (no torchaudio / codec libraries here)")
    x, sr = _read_wav(audio_path)
    x = x.mean(axis=1) if x.shape[1] > 1 else x[:, 0]
    if sr != target_sample_rate:
        from scipy.signal import resample_poly
        g = gcd(int(sr), int(target_sample_rate))
        x = resample_poly(x.astype(np.float64), target_sample_rate // g, sr // g).astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).reshape(1, 1, -1)


def save_audio(audio_outpath, audio_out, sample_rate):
    """audio_out: tensor (1, T) or (T,), float in [-1, 1] -> 16-bit PCM mono WAV (helpers.py:96-104)."""
    x = audio_out.detach().to("cpu", torch.float32).reshape(-1).numpy()
    pcm = np.round(np.clip(x, -1.0, 1.0) * 32767.0).astype("<i2").tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVE" + b"fmt " + struct.pack(
        "<IHHIIHH", 16, 1, 1, int(sample_rate), int(sample_rate) * 2, 2, 16) + b"data" + struct.pack("<I", len(pcm))
    with open(audio_outpath, "wb") as f:
        f.write(hdr + pcm)
    logging.info(f"Successfully saved audio at {audio_outpath}")
