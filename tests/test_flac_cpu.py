"""The native FLAC decoder (csrc/swc_flac.c, the input side of inference.py: LibriSpeech is FLAC) against streams written by
tests/flac_encode.py.  Parity note: no FLAC library or tool exists in this environment (no torchaudio, soundfile, flac, ffmpeg),
so decoder and test encoder are both written from the format description — parity with libFLAC is UNPINNED here; what pins
real files at run time is the decoder's own verification of CRC-8, CRC-16 and the stream's MD5 signature (tested below)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import flac_encode as fe  # noqa: E402

from simwhisper_codec_amd import wavio  # noqa: E402


def _signal(n, ch, bps, seed):
    g = np.random.default_rng(seed)
    t = np.arange(n)[:, None]
    amp = (1 << (bps - 1)) * 0.4
    x = amp * (np.sin(2 * np.pi * (220.0 + 37 * np.arange(ch)) * t / 16000.0) * 0.6 + 0.1 * g.standard_normal((n, ch)))
    if ch == 2:
        x[:, 1] = 0.7 * x[:, 0] + 0.3 * x[:, 1]  # correlated channels: side signals stay small
    return np.clip(np.round(x), -(1 << (bps - 1)), (1 << (bps - 1)) - 1).astype(np.int64)


def _roundtrip(tmp_path, x, sr, bps, **kw):
    p = tmp_path / "t.flac"
    p.write_bytes(fe.encode(x, sr, bps, **kw))
    pcm, got_sr, got_bits = wavio._decode_flac(str(p))
    assert got_sr == sr and got_bits == bps and np.array_equal(pcm.astype(np.int64), x)
    got, _ = wavio._read_flac(str(p))
    assert got.dtype == np.float32 and got.shape == x.shape and float(np.abs(got).max()) <= 1.0
    return p


@pytest.mark.parametrize("ch,bps", [(1, 16), (2, 16), (1, 24), (2, 24), (1, 8), (2, 12), (2, 32), (1, 20)])
def test_flac_subframe_types_and_stereo_modes(tmp_path, ch, bps):
    n = 5 * 1024 + 333
    x = _signal(n, ch, bps, seed=ch * 100 + bps)
    x[1024:2048] = x[1024]                      # a constant block
    x[2048:3072] &= ~np.int64(7)                # three wasted bits in block 2
    kinds = ["verbatim", "constant", ("fixed", 2), ("lpc", 8), ("fixed", 4), ("fixed", 0), ("lpc", 1), ("fixed", 1), ("lpc", 32), ("fixed", 3)]

    def plan(fi, c):
        if c is None:
            return [0, 8, 9, 10][fi % 4]
        return dict(kind=kinds[(fi + (c or 0)) % len(kinds)], porder=[0, 2, 3, 1][fi % 4], rice2=(bps > 16 or fi % 2 == 1),
                    escape_part=(1 if fi % 3 == 2 else None), wasted=(3 if fi == 2 else 0))
    _roundtrip(tmp_path, x, 16000, bps, blocksize=1024, plan=plan)


@pytest.mark.parametrize("blocksize", [192, 256, 1000, 4096, 4608])
def test_flac_block_sizes_and_short_last_block(tmp_path, blocksize):
    x = _signal(3 * blocksize + 17, 1, 16, seed=blocksize)
    _roundtrip(tmp_path, x, 22050, 16, blocksize=blocksize, plan=lambda fi, c: dict(kind=("lpc", 6), porder=0) if c is not None else 0)


def test_flac_load_audio_mono_resample_and_id3(tmp_path):
    x = _signal(16000, 2, 16, seed=5)
    p = tmp_path / "a.flac"
    p.write_bytes(fe.encode(x, 8000, 16, blocksize=4096, id3=True))
    wav = wavio.load_audio(str(p), 16000)
    assert wav.shape == (1, 1, 32000) and wav.dtype == torch.float32
    # the same samples as a WAV file go through the same mono / resampling code: identical result
    import struct
    pcm = x.astype("<i2").tobytes()
    (tmp_path / "a.wav").write_bytes(b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVE" + b"fmt " + struct.pack(
        "<IHHIIHH", 16, 1, 2, 8000, 8000 * 4, 4, 16) + b"data" + struct.pack("<I", len(pcm)) + pcm)
    assert torch.equal(wav, wavio.load_audio(str(tmp_path / "a.wav"), 16000))


def test_flac_corruption_is_detected(tmp_path):
    x = _signal(4096, 1, 16, seed=9)
    good = fe.encode(x, 16000, 16, blocksize=1024)
    p = tmp_path / "bad.flac"
    # a flipped bit in a residual: the frame's CRC-16 no longer matches
    bad = bytearray(good); bad[len(bad) // 2] ^= 0x10
    p.write_bytes(bytes(bad))
    with pytest.raises(ValueError):
        wavio._read_flac(str(p))
    # a stream whose frames are intact but whose MD5 signature belongs to other audio
    other = bytearray(good); other[8 + 18] ^= 0xFF   # first byte of the MD5 field in STREAMINFO
    p.write_bytes(bytes(other))
    with pytest.raises(ValueError, match="MD5"):
        wavio._read_flac(str(p))
    # no signature (all zero): accepted on the CRCs alone
    p.write_bytes(fe.encode(x, 16000, 16, blocksize=1024, md5=False))
    got, _ = wavio._read_flac(str(p))
    assert got.shape == (4096, 1)
    # not FLAC at all
    p.write_bytes(b"fLaC")
    with pytest.raises(ValueError):
        wavio._read_flac(str(p))


def test_flac_decoder_survives_damaged_streams(tmp_path):
    """random byte damage, truncation and garbage tails: the decoder must answer with an error or (when the damage is
    outside what the CRCs and the MD5 cover, e.g. in STREAMINFO's frame-size hints) the right samples, never crash."""
    x = _signal(3000, 2, 16, seed=21)
    good = fe.encode(x, 16000, 16, blocksize=1024,
                     plan=lambda fi, c: [0, 8, 9, 10][fi % 4] if c is None else dict(kind=("lpc", 6), porder=1, rice2=bool(fi % 2)))
    g = np.random.default_rng(0)
    p = tmp_path / "f.flac"
    ok = bad = 0
    for trial in range(400):
        b = bytearray(good)
        mode = trial % 4
        if mode == 0:
            for _ in range(1 + trial % 3):
                b[int(g.integers(0, len(b)))] ^= int(g.integers(1, 256))
        elif mode == 1:
            b = b[: int(g.integers(0, len(b)))]
        elif mode == 2:
            i = int(g.integers(4, len(b)))
            b[i:i + 8] = bytes(g.integers(0, 256, 8, dtype=np.uint8))
        else:
            b += bytes(g.integers(0, 256, int(g.integers(1, 64)), dtype=np.uint8))
        p.write_bytes(bytes(b))
        try:
            got, sr = wavio._read_flac(str(p))
            ok += 1
            assert np.array_equal(np.round(got.astype(np.float64) * 32768).astype(np.int64), x)
        except ValueError:
            bad += 1
    assert bad > 250 and ok + bad == 400



def test_flac_crafted_header_cannot_demand_memory(tmp_path, monkeypatch):
    """STREAMINFO's 36-bit sample count is not trusted for the output allocation.  A header that claims 2^36 - 1
    samples per channel (a 256 GiB buffer if believed) is decoded into buffers sized by the file, and rejected because the
    stream ends early; constant audio, which really does exceed the first size guess, still decodes (the buffer grows)."""
    x = _signal(4096, 1, 16, seed=5)
    raw = bytearray(fe.encode(x, 16000, 16, blocksize=1024, plan=lambda fi, c: dict(kind=("fixed", 2), porder=1) if c is not None else 0))
    assert raw[:4] == b"fLaC"
    q = 8  # STREAMINFO body starts after "fLaC" + 4-byte block header; total samples = low nibble of q[13] + q[14..17]
    raw[q + 13] |= 0x0F
    raw[q + 14:q + 18] = b"\xff\xff\xff\xff"
    p = tmp_path / "crafted.flac"
    p.write_bytes(bytes(raw))
    biggest = []
    real_empty = np.empty
    monkeypatch.setattr(np, "empty", lambda shape, *a, **k: (biggest.append(int(np.prod(shape))), real_empty(shape, *a, **k))[1])
    with pytest.raises(ValueError, match="ceiling"):
        wavio._decode_flac(str(p))
    assert not biggest                                            # beyond the decoder's hard ceiling (2^31 samples): refused before any allocation
    # a declared count UNDER the ceiling (2^30 samples, 4 GiB if believed) that the stream does not hold: buffers are sized by the
    # file and grow a few steps, the decode fails because the stream ends early
    raw[q + 13] &= 0xF0
    raw[q + 14:q + 18] = (1 << 30).to_bytes(4, "big")
    p.write_bytes(bytes(raw))
    with pytest.raises(ValueError):
        wavio._decode_flac(str(p))
    assert biggest and max(biggest) <= 64 * (16 * len(raw) + 65536)   # a few growth steps from the file-size guess, never 2^30
    biggest.clear()
    z = np.zeros((3_000_000, 1), dtype=np.int64)                 # 3 M samples of silence in a few hundred bytes
    pz = tmp_path / "silence.flac"
    pz.write_bytes(fe.encode(z, 16000, 16, blocksize=4096, plan=lambda fi, c: dict(kind="constant") if c is not None else 0))
    pcm, sr, bits = wavio._decode_flac(str(pz))
    assert pcm.shape == (3_000_000, 1) and not pcm.any() and len(biggest) >= 2   # the buffer grew at least once
