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
