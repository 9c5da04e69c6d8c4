"""Audio file IO around the hot path (the role of utils/helpers.py:77-111 in the reference).

torchaudio is not available offline, so this is a small self-contained RIFF/WAVE reader
and writer (stdlib + numpy): PCM 8/16/24/32-bit and IEEE float 32/64 in, PCM16 out — and a native FLAC decoder
(csrc/swc_flac.c -> libswc_io.so, plain C through ctypes; LibriSpeech, the codec's evaluation corpus, is FLAC).
Conventions the reference leaves to torchaudio and which are therefore OUR choice:
  * multi-channel input is averaged to mono (helpers.py:82-83 does the same);
  * sample-rate conversion: `resample` below, the sinc / Hann-window algorithm torchaudio.functional.resample documents
    (its defaults), restated — bit-parity with torchaudio itself is unpinned here;
  * float -> PCM16: round(clip(x, -1, 1) * 32767);
  * FLAC integers are scaled by 2^-(bits-1), as torchaudio does; every decode checks both frame CRCs and the stream's
    MD5 signature and raises on a mismatch (no other FLAC decoder exists here to pin this one against).
`.mp3` is listed (helpers.py:106) but cannot be decoded here: a clear error is raised.
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


_FLAC_ERRORS = {-1: "not a FLAC stream or malformed", -2: "a frame failed its CRC", -3: "uses a reserved / unsupported coding",
                -4: "output buffer too small", -5: "decoded audio does not match the stream's MD5 signature"}
_io_lib = None


def _io():
    """libswc_io.so (built on demand with the host C compiler)."""
    global _io_lib
    if _io_lib is None:
        import ctypes as C
        from . import build
        lib = C.CDLL(build.build_io_library())
        lib.swc_flac_info.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int64)]
        lib.swc_flac_info.restype = C.c_int
        lib.swc_flac_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_int64, C.POINTER(C.c_int32)]
        lib.swc_flac_decode.restype = C.c_int64
        lib.swc_flac_max_samples.argtypes = [C.c_size_t]
        lib.swc_flac_max_samples.restype = C.c_int64
        _io_lib = lib
    return _io_lib


def _read_flac(path):
    """-> (float32 array (n, channels) in [-1, 1), sample rate).  Raises ValueError on anything that does not verify."""
    pcm, sr, bits = _decode_flac(path)
    return pcm.astype(np.float32) / np.float32(2.0 ** (bits - 1)), sr


FLAC_MAX_SAMPLES = 1 << 31  # decoded samples (all channels) one FLAC file may expand to, whatever its header says


def _decode_flac(path):
    """-> (int32 array (n, channels), sample rate, bits per sample)"""
    import ctypes as C
    with open(path, "rb") as f:
        data = f.read()
    lib = _io()
    sr, ch, bits, total = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
    rc = lib.swc_flac_info(data, len(data), C.byref(sr), C.byref(ch), C.byref(bits), C.byref(total))
    if rc != 0:
        raise ValueError(f"{path}: {_FLAC_ERRORS.get(rc, rc)}")
    # STREAMINFO's 36-bit sample count is not trusted for the allocation (a crafted header would ask for 256 GiB): start
    # from what the file's size makes plausible and grow on FLAC_E_SPACE up to the declared (or, undeclared, a hard) bound
    guess = int(lib.swc_flac_max_samples(len(data)))
    # hard ceiling whatever STREAMINFO declares (its 36-bit count is attacker-controlled: a small file of constant subframes
    # could otherwise ask for 2^36 samples x channels x 4 bytes): FLAC_MAX_SAMPLES per channel (default 2^31 / channels
    # int32 samples = 8 GiB of decoded audio at most, > 37 h at 16 kHz mono), SWC_FLAC_MAX_SECONDS overrides by duration
    hard = FLAC_MAX_SAMPLES // max(1, ch.value)
    if os.environ.get("SWC_FLAC_MAX_SECONDS"):
        hard = min(hard, int(float(os.environ["SWC_FLAC_MAX_SECONDS"]) * max(1, sr.value)))
    limit = min(int(total.value) if total.value else hard, hard)
    if total.value and total.value > hard:
        raise ValueError(f"{path}: STREAMINFO declares {total.value} samples per channel, more than the decoder's ceiling of "
                         f"{hard} (FLAC_MAX_SAMPLES / SWC_FLAC_MAX_SECONDS)")
    cap = max(1, min(limit, guess))
    md5 = C.c_int32(0)
    while True:
        out = np.empty((cap, ch.value), dtype=np.int32)
        n = lib.swc_flac_decode(data, len(data), out.ctypes.data_as(C.c_void_p), cap, C.byref(md5))
        if n == -4 and cap < limit:  # constant / highly compressible audio can exceed the first guess: grow
            cap = min(limit, cap * 4)
            continue
        break
    if n < 0:
        raise ValueError(f"{path}: {_FLAC_ERRORS.get(int(n), int(n))}")
    if not md5.value:
        logging.warning(f"{path}: the stream carries no MD5 signature; frame CRCs verified only")
    return out[:n], int(sr.value), int(bits.value)


def read_pcm16(audio_path, target_sample_rate):
    """The common case without any host arithmetic: a mono 16-bit PCM .wav already at target_sample_rate -> its samples as an
    int16 tensor (T,) (a copy of the file's bytes), else None.  `samples * 2^-15` (ops.pcm16_to_f32 on the GPU) is then exactly
    what load_audio returns for the same file."""
    if os.path.splitext(audio_path)[1].lower() != ".wav":
        return None
    with open(audio_path, "rb") as f:
        data = f.read()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        return None
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        if cid == b"fmt " and size >= 16:
            fmt = struct.unpack("<HHIIHH", data[pos + 8:pos + 24])
        elif cid == b"data":
            pcm = (pos + 8, min(size, len(data) - pos - 8))
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None or fmt[0] != 1 or fmt[1] != 1 or fmt[2] != int(target_sample_rate) or fmt[5] != 16:
        return None
    return torch.from_numpy(np.frombuffer(data, dtype="<i2", count=pcm[1] // 2, offset=pcm[0]).copy())


def save_pcm16(audio_outpath, pcm, sample_rate):
    """int16 samples (tensor or array, any shape, mono) -> 16-bit PCM WAV: the file save_audio writes once the conversion
    round(clip(x, -1, 1) * 32767) has been done elsewhere (ops.f32_to_pcm16 on the GPU)."""
    raw = (pcm.detach().cpu().numpy() if isinstance(pcm, torch.Tensor) else np.asarray(pcm)).astype("<i2", copy=False).reshape(-1).tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(raw)) + b"WAVE" + b"fmt " + struct.pack(
        "<IHHIIHH", 16, 1, 1, int(sample_rate), int(sample_rate) * 2, 2, 16) + b"data" + struct.pack("<I", len(raw))
    with open(audio_outpath, "wb") as f:
        f.write(hdr + raw)
    logging.info(f"Successfully saved audio at {audio_outpath}")


def load_audio(audio_path, target_sample_rate):
    """-> FloatTensor (1, 1, T) at target_sample_rate, mono (helpers.py:77-94)."""
    ext = os.path.splitext(audio_path)[1].lower()
    if ext == ".flac":
        x, sr = _read_flac(audio_path)
    elif ext == ".wav":
        x, sr = _read_wav(audio_path)
    else:
        raise RuntimeError(f"{audio_path}: only .wav and .flac can be decoded offline (no torchaudio / codec libraries here)")
    x = x.mean(axis=1) if x.shape[1] > 1 else x[:, 0]
    wav = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    if sr != target_sample_rate:
        wav = resample(wav, int(sr), int(target_sample_rate))
    return wav.reshape(1, 1, -1)


def resample(wav, orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """Band-limited sinc interpolation with a Hann window: the algorithm torchaudio.functional.resample documents and uses
    with its defaults (`sinc_interp_hann`, lowpass_filter_width 6, rolloff 0.99), which is what helpers.py:86 calls.
    Restated from that published algorithm — torchaudio is absent here, so bit-parity with it is unpinned: kernel of
    new/gcd phases x (2 width + orig/gcd) taps built in float64, float32 convolution with stride orig/gcd over the signal
    padded by (width, width + orig/gcd), output length ceil(new * n / orig).  wav: 1-D float32 tensor."""
    import math
    g = gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    if orig == new:
        return wav
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kernel = torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * (base / orig)
    kernel = kernel.to(torch.float32)                                   # [new, 1, 2 width + orig]
    n = wav.shape[-1]
    padded = torch.nn.functional.pad(wav.reshape(1, 1, -1).to(torch.float32), (width, width + orig))
    out = torch.nn.functional.conv1d(padded, kernel, stride=orig)       # [1, new, frames]
    out = out.transpose(1, 2).reshape(-1)
    return out[: int(math.ceil(new * n / orig))].contiguous()


def save_audio(audio_outpath, audio_out, sample_rate):
    """audio_out: tensor (1, T) or (T,), float in [-1, 1] -> 16-bit PCM mono WAV (helpers.py:96-104)."""
    x = audio_out.detach().to("cpu", torch.float32).reshape(-1).numpy()
    # (np.minimum / np.maximum: the same values as np.clip at a quarter of its time on 160 000 samples)
    pcm = np.round(np.minimum(np.maximum(x, np.float32(-1.0)), np.float32(1.0)) * np.float32(32767.0)).astype("<i2").tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVE" + b"fmt " + struct.pack(
        "<IHHIIHH", 16, 1, 1, int(sample_rate), int(sample_rate) * 2, 2, 16) + b"data" + struct.pack("<I", len(pcm))
    with open(audio_outpath, "wb") as f:
        f.write(hdr + pcm)
    logging.info(f"Successfully saved audio at {audio_outpath}")
