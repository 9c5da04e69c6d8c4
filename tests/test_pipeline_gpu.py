"""pipeline.InFlight: consecutive batches on two host threads / two streams over ONE set of device operands
(AudioCodec.replica()).  Same kernels on the same data, only beside another batch: results must be bit-identical to
the serial loop, for ragged batches and with every batch different."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from common import PARAMS, state_dict  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"
_MODELS = {}


def model(tag, precision):
    from simwhisper_codec_amd.codec import AudioCodec
    key = (tag, precision)
    if key not in _MODELS:
        m = AudioCodec(PARAMS[tag](), precision=precision)
        m.load_state_dict(state_dict(tag), strict=True)
        _MODELS[key] = m.to(DEV).eval()
    return _MODELS[key]


def _round_trip(m, wavs):
    enc = m.encode(wavs)
    return enc["codes_list"], m.decode(enc["codes_list"])["syn_wav_list"]


@pytest.mark.parametrize("tag,precision", [("tiny", "mixed"), ("real", "mixed"), ("tiny", "fp32")])
def test_batches_in_flight_equal_the_serial_loop(tag, precision):
    from simwhisper_codec_amd import synth
    from simwhisper_codec_amd.pipeline import InFlight
    m = model(tag, precision)
    batches = []
    for b in range(5):
        secs = [3.0 + b, 0.4, 1.7 + 0.3 * b, 2.2][: 2 + b % 3]
        batches.append([synth.synth_audio(int(16000 * t) + 13 * i, index=1500 + 10 * b + i, kind="speech" if i % 2 else "noise").to(DEV)
                        for i, t in enumerate(secs)])
    want = [_round_trip(m, w) for w in batches]
    with InFlight(m, 2) as pipe:
        r = pipe.models[1]
        assert r is not m and r._packed() is m._packed()          # one set of operands, nothing re-packed
        got = pipe.map(_round_trip, batches)
        got2 = pipe.map(_round_trip, batches[::-1])[::-1]           # another interleaving
    for (wc, ww), (gc, gw), (hc, hw) in zip(want, got, got2):
        for a, b, c in zip(wc + ww, gc + gw, hc + hw):
            assert a.shape == b.shape and torch.equal(a, b) and torch.equal(a, c)


def test_replica_cannot_repack():
    from simwhisper_codec_amd._lib import SwcError
    m = model("tiny", "mixed")
    r = m.replica()
    with pytest.raises(SwcError):
        r.precision = "bf16"


def test_host_stager_moves_the_same_values_as_the_per_file_path(tmp_path):
    """pipeline.HostStager (the file loop of inference.py): a ragged batch incl. an empty utterance staged as 16-bit samples and
    converted on the GPU equals load_audio's floats bit for bit; decode()'s rows converted on the GPU and copied once equal the
    host conversion of wavio.save_audio; the f32 staging path equals a plain copy; lists without a common buffer fall back."""
    import numpy as np
    from simwhisper_codec_amd import synth, wavio
    from simwhisper_codec_amd.pipeline import HostStager
    st = HostStager()
    lens = [16000, 0, 4803, 1, 31999]
    pcm, flt = [], []
    for i, n in enumerate(lens):
        p = str(tmp_path / f"u{i}.wav")
        wavio.save_audio(p, (synth.synth_audio(n, index=40 + i, kind="speech") * 3.0).reshape(1, -1), 16000)   # clips in places
        pcm.append(wavio.read_pcm16(p, 16000))
        flt.append(wavio.load_audio(p, 16000).reshape(-1))
    dev = torch.device("cuda", 0)
    got = st.to_device_pcm16(pcm, dev)
    assert [int(g.numel()) for g in got] == lens and all(g.dtype == torch.float32 for g in got)
    for g, f in zip(got, flt):
        assert torch.equal(g.cpu(), f)
        assert g.data_ptr() % 16 == 0 or g.numel() == 0
    got = st.to_device(flt, dev)
    for g, f in zip(got, flt):
        assert torch.equal(g.cpu(), f)
    # rows of one padded buffer (what decode() returns), values beyond +-1 and exact ties included
    base = torch.randn(4, 5000, device=dev) * 0.7
    base[0, :6] = torch.tensor([1.5, -1.5, 0.5 / 32767, 1.5 / 32767, -2.5 / 32767, 1.0], device=dev)
    rows = [base[i, :n] for i, n in enumerate([5000, 0, 1234, 4999])]
    dev16 = st.pcm16_on_device(rows)
    assert all(d._base is dev16[0]._base for d in dev16)
    host = st.to_host(dev16)
    for r, h in zip(rows, host):
        want = torch.from_numpy(np.round(np.clip(r.cpu().numpy(), -1.0, 1.0) * 32767.0).astype("<i2"))
        assert h.dtype == torch.int16 and torch.equal(h, want)
    # rows of two buffers (what rank 0 of DataParallelCodec gets: one buffer per rank), interleaved: one conversion and copy each
    base2 = torch.randn(3, 700, device=dev)
    mixed = [base[0, :100], base2[1, :700], base[3, :9], base2[0, :1]]
    d16 = st.pcm16_on_device(mixed)
    assert d16[0]._base is d16[2]._base and d16[1]._base is d16[3]._base and d16[0]._base is not d16[1]._base
    for r, h in zip(mixed, st.to_host(d16)):
        assert torch.equal(h, torch.from_numpy(np.round(np.clip(r.cpu().numpy(), -1.0, 1.0) * 32767.0).astype("<i2")))
    # no common buffer: per-tensor fall-back, same values
    loose = [torch.randn(100, device=dev), torch.randn(7, device=dev)]
    for r, h in zip(loose, st.to_host(st.pcm16_on_device(loose))):
        assert torch.equal(h, torch.from_numpy(np.round(np.clip(r.cpu().numpy(), -1.0, 1.0) * 32767.0).astype("<i2")))
    assert st.to_host([]) == [] and st.pcm16_on_device([]) == []
