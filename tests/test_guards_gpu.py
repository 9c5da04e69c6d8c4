"""Guards around the hot path: range guard of the split-f16 / fp8 operands, device guard, default-device fast path,
out-of-range inputs that the reference tolerates."""
import logging

import numpy as np
import pytest
import torch

from common import PARAMS, oracle, state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(tag, precision, sd=None):
    from simwhisper_codec_amd.codec import AudioCodec
    m = AudioCodec(PARAMS[tag](), precision=precision)
    m.load_state_dict(sd if sd is not None else state_dict(tag), strict=True)
    return m.to(DEV).eval()


def test_saturation_counter_unit():
    """|x * 64| > 65504 in a split-f16 producer and |x * 16| > 448 in an fp8 producer set their counters; in-range
    data leaves them at zero."""
    from simwhisper_codec_amd import ops
    cnt = torch.zeros(2, dtype=torch.int32, device=DEV)
    ops.set_saturation_counter(cnt)
    try:
        x = torch.full((4, 64), 3.0, device=DEV)
        ops.cast_f16s(x, 64)
        ops.cast_fp8(x)
        assert cnt.tolist() == [0, 0]
        x[1, 5] = 1023.4  # 1023.4 * 64 = 65497.6 < 65504: still representable
        ops.cast_f16s(x, 64)
        assert cnt.tolist() == [0, 0]
        x[1, 5] = -1024.0
        y = ops.cast_f16s(x, 64)
        assert cnt.tolist()[0] == 1
        v = y.view(4, 2, 2, 32).float()
        assert float((v[1, 0, 0, 5] + v[1, 0, 1, 5]) / 64) == -65504.0 / 64  # clipped, not inf
        x[1, 5] = 30.0  # 30 * 16 = 480 > 448
        ops.cast_fp8(x)
        assert cnt.tolist() == [1, 1]
        # LayerNorm with a large gain: output 50 * ~N(0,1) * 64 stays in range; gain 2000 does not
        h = torch.randn(8, 128, device=DEV)
        w, b = torch.full((128,), 50.0, device=DEV), torch.zeros(128, device=DEV)
        ops.layernorm(h, w, b, 1e-5, B=1, t_in=8, C_=128, out_dtype=torch.float16)
        assert cnt.tolist() == [1, 1]
        ops.layernorm(h, w * 40, b, 1e-5, B=1, t_in=8, C_=128, out_dtype=torch.float16)
        assert cnt.tolist()[0] > 1
    finally:
        ops.set_saturation_counter(None)
    n0 = cnt.tolist()
    ops.cast_f16s(torch.full((4, 64), 5000.0, device=DEV), 64)  # accounting off: nothing is counted
    assert cnt.tolist() == n0


def _outlier_state_dict(tag):
    """Whisper-style outlier channels on top of the synthetic checkpoint, as a function-preserving re-scaling: the
    LayerNorm gain / bias of a few channels x1024 and the consuming linears' columns / 1024 (powers of two: exact in
    fp32, so the reference's codes do not move), which pushes those LayerNorm outputs beyond the split-f16 range
    |x| < 1023."""
    sd = {k: v.clone() for k, v in state_dict(tag).items()}
    ch = [3, 4, 77]
    p = "acoustic_encoder.layers.0."
    sd[p + "final_layer_norm.weight"][ch] *= 1024.0
    sd[p + "final_layer_norm.bias"][ch] *= 1024.0
    sd[p + "fc1.weight"][:, ch] /= 1024.0
    p = "acoustic_encoder.layers.1."
    sd[p + "self_attn_layer_norm.weight"][ch] *= 1024.0
    sd[p + "self_attn_layer_norm.bias"][ch] *= 1024.0
    for w in ("q_proj", "k_proj", "v_proj"):
        sd[p + f"self_attn.{w}.weight"][:, ch] /= 1024.0
    return sd


def test_outlier_checkpoint_falls_back_and_stays_bit_exact(caplog):
    """A checkpoint whose activations leave the split-f16 range must still give the reference's codes: the encode is
    re-run on exact-f32 operands (policy "fallback"), and "raise" reports it instead."""
    from oracle.ref_cpu import Oracle
    from simwhisper_codec_amd import synth
    from simwhisper_codec_amd._lib import SwcError
    tag = "tiny"
    sd = _outlier_state_dict(tag)
    wavs = [synth.synth_audio(30000, index=300, kind="speech"), synth.synth_audio(21111, index=301, kind="noise")]
    want = Oracle(PARAMS[tag](), sd).encode(wavs, trim=True)["codes_list"]
    m = _model(tag, "mixed", sd)
    m.saturation_policy = "raise"
    with pytest.raises(SwcError, match="clipped"):
        m.encode([w.to(DEV) for w in wavs])
    m.saturation_policy = "ignore"
    bad = m.encode([w.to(DEV) for w in wavs])["codes_list"]
    assert m.precision == "mixed" and m.saturation_count()["f16s"] > 0
    m.saturation_policy = "fallback"
    with caplog.at_level(logging.WARNING):
        got = m.encode([w.to(DEV) for w in wavs])["codes_list"]
    assert m.precision == "mixed_f32" and "clipped" in caplog.text
    for a, b in zip(got, want):
        assert torch.equal(a.cpu().long(), b.long())
    # the clipped run really was different (otherwise this test exercises nothing)
    assert any(not torch.equal(a.cpu().long(), b.long()) for a, b in zip(bad, want))
    # the synthetic checkpoint itself never clips
    m2 = _model(tag, "mixed")
    m2.encode([w.to(DEV) for w in wavs])
    assert m2.saturation_count() == {"f16s": 0, "fp8": 0} and m2.precision == "mixed"


def test_replica_follows_the_range_fallback():
    """Batches in flight (pipeline.InFlight): a replica has no weights to pack from.  When ITS batch clips, the origin packs
    the exact-f32 preset once and the replica adopts it; the other replica follows at its next call.  Codes = the oracle's."""
    from oracle.ref_cpu import Oracle
    from simwhisper_codec_amd import synth
    from simwhisper_codec_amd.pipeline import InFlight
    tag = "tiny"
    sd = _outlier_state_dict(tag)
    wavs = [synth.synth_audio(30000, index=300, kind="speech"), synth.synth_audio(21111, index=301, kind="noise")]
    want = Oracle(PARAMS[tag](), sd).encode(wavs, trim=True)["codes_list"]
    m = _model(tag, "mixed", sd)
    assert m.precision == "mixed"
    dw = [w.to(DEV) for w in wavs]
    with InFlight(m, 2) as pipe:
        outs = pipe.map(lambda mdl, w: mdl.encode(w)["codes_list"], [dw] * 6)
        outs += [mm.encode(dw)["codes_list"] for mm in pipe.models]   # a model that sat idle follows at its next call
        assert all(mm.precision == "mixed_f32" for mm in pipe.models)
        assert pipe.models[1]._packed() is m._packed()
    for got in outs:
        for a, b in zip(got, want):
            assert torch.equal(a.cpu().long(), b.long())


def test_device_guard(monkeypatch):
    """Kernels run on the CURRENT device's stream: a tensor of another GPU is refused, and the public entry points make
    the model's GPU current (no second GPU needed: the current-device query is patched)."""
    from simwhisper_codec_amd import ops
    from simwhisper_codec_amd._lib import SwcError
    x = torch.zeros(4, 64, device=DEV)
    real = torch._C._cuda_getDevice
    monkeypatch.setattr(torch._C, "_cuda_getDevice", lambda: 1)
    with pytest.raises(SwcError, match="is current"):
        ops.cast_f16s(x, 64)
    monkeypatch.setattr(torch._C, "_cuda_getDevice", real)
    ops.cast_f16s(x, 64)
    m = _model("tiny", "mixed")
    with pytest.raises(SwcError, match="move the model"):
        m.encode([torch.zeros(4000, device=DEV)], device=torch.device("cuda", 1))


def test_default_device_takes_the_gather_path(monkeypatch):
    """encode()/decode() with the reference's default device=torch.device("cuda") (no index) must reach the one-kernel
    batch assembly and give the same bits as the indexed-device call."""
    from simwhisper_codec_amd import ops, synth
    m = _model("tiny", "mixed")
    wavs = [synth.synth_audio(20000 + 999 * i, index=500 + i).to(DEV) for i in range(3)]
    calls = []
    real = ops.gather_rows
    monkeypatch.setattr(ops, "gather_rows", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    c_def = m.encode(wavs)["codes_list"]
    assert len(calls) == 1
    w_def = m.decode(c_def)["syn_wav_list"]
    assert len(calls) == 2
    c_idx = m.encode(wavs, device=torch.device("cuda", 0))["codes_list"]
    w_idx = m.decode(c_idx, device=torch.device("cuda", 0))["syn_wav_list"]
    assert len(calls) == 4
    for a, b in zip(c_def + w_def, c_idx + w_idx):
        assert torch.equal(a, b)


def test_fsq_decode_out_of_range_codes():
    """quantizer.py:214-216 uses torch's floor `//` and `%`: negative and >= 2016 codes wrap the same way."""
    from simwhisper_codec_amd import ops
    codes = torch.tensor([-1, -7, -2016, -2017, 2016, 2017, 5000, 0, 2015, 336 * 6 + 5], dtype=torch.int64)
    G, B, T = 8, 1, codes.numel()
    c = codes.view(1, 1, T).expand(G, B, T).contiguous().to(DEV)
    lens = torch.tensor([T], dtype=torch.int32, device=DEV)
    zq = ops.fsq_decode(c, lens, B=B, T=T, G=G).cpu()
    base = torch.tensor([1, 8, 56, 336])
    lev = torch.tensor([8, 7, 6, 6])
    nn = (codes[:, None] // base) % lev
    want = (nn - lev // 2).float() / (lev // 2).float()
    for g in range(G):
        assert torch.equal(zq[0, :, 4 * g:4 * g + 4], want)


def test_tokenize_length_beyond_the_row():
    """model.py:180 slices xi[:, :x_len]: a length larger than the tensor is the tensor (no read past the row)."""
    from simwhisper_codec_amd import synth
    m = _model("tiny", "mixed")
    w = synth.synth_audio(16000, index=600).to(DEV)
    a = m.inference_tokenize(w.view(1, 1, -1), torch.tensor([16000]))
    b = m.inference_tokenize(w.view(1, 1, -1), torch.tensor([999999]))
    assert torch.equal(a["codes"], b["codes"]) and torch.equal(a["codes_lengths"], b["codes_lengths"])


@pytest.mark.parametrize("precision", ["mixed", "fp8"])
def test_packed_operand_checkpoint_is_identical(tmp_path, precision):
    """tools/pack_checkpoint.py --fold: a model loaded from the packed-operand file never runs the fold /
    scale / cast pass and must give bit-identical codes and waveforms to the model loaded from the reference layout."""
    import subprocess, sys, yaml
    from audiocodec.model import AudioCodec
    from simwhisper_codec_amd import synth
    from simwhisper_codec_amd._lib import SwcError
    from common import ROOT
    import os
    gp = PARAMS["tiny"]()
    cfg = tmp_path / "cfg.yaml"
    cfg.write_text(yaml.safe_dump({"generator_params": gp}))
    torch.save(state_dict("tiny"), tmp_path / "ref.pt")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pack_checkpoint.py"), "--config", str(cfg), "--in",
                        str(tmp_path / "ref.pt"), "--fold", "--precision", precision, "--out", str(tmp_path / "pk.safetensors")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-1500:]
    a = AudioCodec.load_from_checkpoint(str(cfg), str(tmp_path / "ref.pt"))
    a.precision = precision
    a = a.to(DEV).eval()
    b = AudioCodec.load_from_checkpoint(str(cfg), str(tmp_path / "pk.safetensors")).to(DEV).eval()
    assert b.precision == precision and len(list(b.buffers())) == 1  # no state_dict tensors were built or moved
    called = []
    b._pack = lambda dev: called.append(1)
    wavs = [synth.synth_audio(30000 + 777 * i, index=900 + i, kind="speech" if i % 2 else "noise").to(DEV) for i in range(3)]
    ca, cb = a.encode(wavs)["codes_list"], b.encode(wavs)["codes_list"]
    wa, wb = a.decode(ca)["syn_wav_list"], b.decode(cb)["syn_wav_list"]
    assert not called
    for x, y in zip(ca + wa, cb + wb):
        assert torch.equal(x, y)
    with pytest.raises(SwcError, match="packed for precision"):
        b.precision = "fp32"
    # a packed file is bound to its configuration
    gp2 = PARAMS["tiny"]()
    gp2["vocos"]["num_layers"] = 2
    cfg2 = tmp_path / "cfg2.yaml"
    cfg2.write_text(yaml.safe_dump({"generator_params": gp2}))
    with pytest.raises(SwcError, match="another configuration"):
        AudioCodec.load_from_checkpoint(str(cfg2), str(tmp_path / "pk.safetensors"))


def test_deferred_range_check_redo():
    """deferred_range_check(): no read-back (no stream sync) between encode and decode; one read-back when the block ends;
    on clipping the model has switched to exact-f32 operands and the redone block gives the reference's codes."""
    from oracle.ref_cpu import Oracle
    from simwhisper_codec_amd import synth
    tag = "tiny"
    sd = _outlier_state_dict(tag)
    wavs = [synth.synth_audio(30000, index=300, kind="speech"), synth.synth_audio(21111, index=301, kind="noise")]
    want = Oracle(PARAMS[tag](), sd).encode(wavs, trim=True)["codes_list"]
    m = _model(tag, "mixed", sd)
    dw = [w.to(DEV) for w in wavs]
    with m.deferred_range_check() as chk:
        c = m.encode(dw)["codes_list"]
        assert m.precision == "mixed"          # nothing was read back yet
        m.decode(c)
    assert chk.clipped and m.precision == "mixed_f32"
    with m.deferred_range_check() as chk2:
        c = m.encode(dw)["codes_list"]
        m.decode(c)
    assert not chk2.clipped
    for a, b in zip(c, want):
        assert torch.equal(a.cpu().long(), b.long())
    m2 = _model(tag, "mixed")               # the plain checkpoint: nothing clips, nothing changes
    with m2.deferred_range_check() as chk3:
        m2.decode(m2.encode(dw)["codes_list"])
    assert not chk3.clipped and m2.precision == "mixed"


def test_packed_outlier_checkpoint_falls_back(tmp_path, caplog):
    """The deployment case of the range guard: an outlier checkpoint shipped as a packed-operand file (`export_packed` /
    tools/pack_checkpoint.py --fold, preset `mixed`).  The file carries the exact-f32 encoder operands of the fallback
    preset beside the split-f16 ones; when the encode clips, the model adopts them (no f32 weights exist to pack from)
    and returns the reference's codes — also from a replica running beside its origin (pipeline.InFlight)."""
    import yaml
    from audiocodec.model import AudioCodec
    from oracle.ref_cpu import Oracle
    from simwhisper_codec_amd import packed, synth
    from simwhisper_codec_amd.pipeline import InFlight
    tag = "tiny"
    sd = _outlier_state_dict(tag)
    wavs = [synth.synth_audio(30000, index=300, kind="speech"), synth.synth_audio(21111, index=301, kind="noise")]
    want = Oracle(PARAMS[tag](), sd).encode(wavs, trim=True)["codes_list"]
    cfg = tmp_path / "cfg.yaml"
    cfg.write_text(yaml.safe_dump({"generator_params": PARAMS[tag]()}))
    path = str(tmp_path / "outlier.mixed.safetensors")
    src = _model(tag, "mixed", sd)
    src.export_packed(path)
    assert src.precision == "mixed"                       # exporting the fallback operands did not switch the source
    assert packed.peek(path)["fallback"] == "mixed_f32"
    dw = [w.to(DEV) for w in wavs]
    m = AudioCodec.load_from_checkpoint(str(cfg), path).to(DEV).eval()
    assert m.precision == "mixed" and len(list(m.buffers())) == 1
    with caplog.at_level(logging.WARNING):
        got = m.encode(dw)["codes_list"]
    assert m.precision == "mixed_f32" and "clipped" in caplog.text
    for a, b in zip(got, want):
        assert torch.equal(a.cpu().long(), b.long())
    wav = m.decode(got)["syn_wav_list"]                    # the decode side of the file is untouched by the switch
    ref = src.decode([c.to(DEV) for c in want])["syn_wav_list"]
    for a, b in zip(wav, ref):
        assert torch.equal(a, b)
    # replicas of a packed-file model follow the same way
    m2 = AudioCodec.load_from_checkpoint(str(cfg), path).to(DEV).eval()
    with InFlight(m2, 2) as pipe:
        outs = pipe.map(lambda mdl, w: mdl.encode(w)["codes_list"], [dw] * 4)
        outs += [mm.encode(dw)["codes_list"] for mm in pipe.models]
        assert all(mm.precision == "mixed_f32" for mm in pipe.models)
    for got in outs:
        for a, b in zip(got, want):
            assert torch.equal(a.cpu().long(), b.long())
    # a file without the fallback part (older writer): the clip is reported as such, not as a packing mismatch
    P = src._packed()
    old = str(tmp_path / "old.safetensors")
    from simwhisper_codec_amd.codec import _PACK_CLASSES, _config_digest
    from simwhisper_codec_amd import ops
    from simwhisper_codec_amd._lib import SwcError
    packed.save(old, P, _PACK_CLASSES, {"precision": "mixed", "config": _config_digest(PARAMS[tag]()), "abi": ops.abi_version()})
    m3 = AudioCodec.load_from_checkpoint(str(cfg), old).to(DEV).eval()
    with pytest.raises(SwcError, match="clipped"):
        m3.encode(dw)


def test_nested_models_keep_their_range_counters():
    """_on_model_device restores the calling thread's counter pointer: a call into a second model from inside (or between)
    calls of a first one must not leave the first one's later kernels uncounted."""
    from simwhisper_codec_amd import synth
    tag = "tiny"
    sd = _outlier_state_dict(tag)
    a = _model(tag, "mixed", sd)
    a.saturation_policy = "ignore"
    b = _model(tag, "mixed")
    w = [synth.synth_audio(30000, index=300, kind="speech").to(DEV)]
    real = a._encode_padded

    def nested(*args, **kw):      # model b runs to completion in the middle of a's encode
        b.encode(w)
        return real(*args, **kw)
    a._encode_padded = nested
    try:
        a.encode(w)
    finally:
        del a._encode_padded
    assert a.saturation_count()["f16s"] > 0 and b.saturation_count() == {"f16s": 0, "fp8": 0}
