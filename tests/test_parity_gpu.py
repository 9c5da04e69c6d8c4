"""End-to-end parity of the HIP path (through the C-ABI) against
  (1) the golden fixtures produced by the reference itself (tests/golden), and
  (2) the CPU oracle run on this box on fresh seeded inputs.
Bar: codes bit-exact (fp32 encode path); waveforms within the stated tolerance."""
import os

import numpy as np
import pytest
import torch

from common import PARAMS, golden, golden_audio, oracle, state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda"

# waveform tolerances, relative to the golden waveform's peak amplitude
TOL_FP32 = 5e-5   # fp32 MFMA path vs the reference fp32 CPU path (measured 1-3e-6 on MI355X)
TOL_BF16 = 2.5e-2  # bf16 MFMA decode (f32 accumulate + residual stream) given IDENTICAL codes: north_star allows 5e-2; measured 0.6-1.1e-2
#                    since the ISTFT head runs on split-f16 operands (round 4), 1.2-2.1e-2 with every decode stage on bf16
TOL_F16S = 1e-4   # split-f16 x3 MFMA decode (preset `f16s`: f32-class operands on both sides; refit GELU / sine of the f32 path)

_MODELS = {}


def model(tag, precision):
    from simwhisper_codec_amd.codec import AudioCodec
    key = (tag, precision)
    if key not in _MODELS:
        m = AudioCodec(PARAMS[tag](), precision=precision)
        m.load_state_dict(state_dict(tag), strict=True)
        _MODELS[key] = m.to(DEV).eval()
    return _MODELS[key]


def _report(name, **kv):
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_report.txt", "a") as f:
        f.write(name + " " + " ".join(f"{k}={v:.3e}" if isinstance(v, float) else f"{k}={v}" for k, v in kv.items()) + "\n")


def _relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12)) if a.size else 0.0


@pytest.mark.parametrize("precision", ["fp32", "mixed"])
@pytest.mark.parametrize("tag", ["tiny", "real"])
@pytest.mark.parametrize("name", ["single", "ragged"])
def test_tokenize_stages(tag, name, precision):
    g = golden(tag, name)
    m = model(tag, precision)
    wavs = golden_audio(g)
    n = [len(w) for w in wavs]
    x = torch.zeros(len(wavs), 1, max(n))
    for i, w in enumerate(wavs):
        x[i, 0, : n[i]] = w
    r = m.inference_tokenize(x.to(DEV), torch.tensor(n))
    assert r["codes"].shape == g["st_codes"].shape and r["codes"].dtype == torch.int32
    assert np.array_equal(r["codes_lengths"].cpu().numpy(), g["st_code_lens"])
    codes = r["codes"].cpu().numpy()
    mism = int((codes != g["st_codes"]).sum())
    _report(f"tokenize/{tag}/{name}/{precision}", code_mismatch=mism, total=codes.size,
            zq_err=float(np.abs(r["zq"].cpu().numpy() - g["st_zq"]).max()))
    assert mism == 0
    assert np.array_equal(r["zq"].cpu().numpy(), g["st_zq"])


@pytest.mark.parametrize("precision", ["fp32", "mixed"])  # mixed = split-f16 encoder (3 f16 MFMAs, f32-class)
@pytest.mark.parametrize("tag", ["tiny", "real"])
@pytest.mark.parametrize("name", ["single", "ragged", "zeros", "short", "chunked"])
def test_encode_codes_bit_exact(tag, name, precision):
    g = golden(tag, name)
    m = model(tag, precision)
    wavs = [w.to(DEV) for w in golden_audio(g)]
    enc = m.encode(wavs, overlap_seconds=10)
    tot = mism = 0
    for i, c in enumerate(enc["codes_list"]):
        want = g[f"codes_{i}"]
        assert tuple(c.shape) == want.shape, (c.shape, want.shape)
        mism += int((c.cpu().numpy() != want).sum()); tot += want.size
    _report(f"encode/{tag}/{name}/{precision}", code_mismatch=mism, total=tot)
    assert mism == 0


@pytest.mark.parametrize("precision,tol", [("fp32", TOL_FP32), ("mixed", TOL_BF16), ("f16s", TOL_F16S)])
@pytest.mark.parametrize("tag", ["tiny", "real"])
@pytest.mark.parametrize("name", ["single", "ragged", "zeros", "short", "chunked"])
def test_decode_waveform(tag, name, precision, tol):
    g = golden(tag, name)
    m = model(tag, precision)
    nutt = len(g["spec_n"])
    codes = [torch.from_numpy(g[f"codes_{i}"]).to(DEV) for i in range(nutt)]
    dec = m.decode(codes, overlap_seconds=10)
    worst = 0.0
    for i, w in enumerate(dec["syn_wav_list"]):
        w = w.float().cpu().numpy()
        assert w.shape[0] == (int(g["spec_n"][i]) // 1280) * 1280
        assert np.isfinite(w).all()
        if f"wav_{i}" in g:
            e = _relerr(w, g[f"wav_{i}"])
        else:
            e = _relerr(w[::7], g[f"wav_stride7_{i}"])
            en = (w.reshape(-1, 1280).astype(np.float64) ** 2).sum(1)
            assert np.allclose(en, g[f"wav_energy_{i}"], rtol=20 * tol, atol=1e-6)
        worst = max(worst, e)
    _report(f"decode/{tag}/{name}/{precision}", rel_err=worst)
    assert worst < tol, worst


@pytest.mark.parametrize("opts,bound", [((True, False, False, False), 1.2e-2), ((True, True, False, False), 9e-3),
                                         ((True, True, True, False), 8e-3), ((True, True, True, True), 8e-3),
                                         ((False, False, False, False), TOL_BF16)])
def test_decode_small_stage_options(opts, bound):
    """The `mixed` preset's decode with the small stages around the bf16 decoder layers / ConvNeXt blocks on split-f16 operands
    (AudioCodec.vocos_head_split_f16 — the default —, decoder_io_split_f16, upsample_split_f16) and plain-f16 operands inside the
    fused ConvNeXt block (convnext_f16), against the reference's own waveforms: every combination the codec offers stays inside
    the preset's tolerance, and each step towards f32-class stages tightens the measured error (modules.py:601-631, 437-474,
    1053-1082)."""
    import simwhisper_codec_amd.codec as C
    names = ("vocos_head_split_f16", "decoder_io_split_f16", "upsample_split_f16", "convnext_f16")
    m = model("real", "mixed")
    keep = {n: getattr(m, n) for n in names}
    keep_rows = m.fused_mlp_min_rows
    try:
        for n, v in zip(names, opts):
            setattr(m, n, v)
        m._pk = None
        m.fused_mlp_min_rows = 0     # (the fused block kernel, where convnext_f16 lives)
        worst = 0.0
        for name in ("single", "ragged", "short"):
            g = golden("real", name)
            nutt = len(g["spec_n"])
            wav = m.decode([torch.from_numpy(g[f"codes_{i}"]).to(DEV) for i in range(nutt)])["syn_wav_list"]
            worst = max([worst] + [_relerr(wav[i].float().cpu().numpy(), g[f"wav_{i}"]) for i in range(nutt)])
        _report("decode_options/real/" + "".join("1" if v else "0" for v in opts), rel_err=worst)
        assert worst < bound, worst
    finally:
        for n, v in keep.items():
            setattr(m, n, v)
        m.fused_mlp_min_rows = keep_rows
        m._pk = None


@pytest.mark.parametrize("tag", ["tiny", "real"])
def test_forward(tag):
    from simwhisper_codec_amd import synth
    g = golden(tag, "forward")
    T, lens = int(g["T"]), g["lens"]
    mel = torch.from_numpy(synth._uniform("forward/mel", len(lens) * 80 * T, 77).reshape(len(lens), 80, T) * 0.8 + 0.2)
    r = model(tag, "fp32").forward({"mel_features": mel.to(DEV), "mel_lens": torch.from_numpy(lens).to(DEV)})
    assert np.array_equal(r["audio_lengths"].cpu().numpy(), g["audio_lengths"])
    a = r["reconstructed_audio"][:, 0].cpu().numpy()
    assert a.shape == g["audio"].shape
    # compare inside each utterance's valid length (beyond it the reference output is don't-care padding)
    worst = max(_relerr(a[i, :n], g["audio"][i, :n]) for i, n in enumerate(g["audio_lengths"]))
    _report(f"forward/{tag}", rel_err=worst)
    assert worst < 1e-4  # fp32 encode+decode, no code flipped (measured 3e-6)


def test_vs_oracle_fresh_input():
    """Fresh seeded input (not in any fixture): HIP path vs the CPU oracle on this box."""
    from simwhisper_codec_amd import synth
    tag = "tiny"
    wavs = [synth.synth_audio(30000 + 1777 * i, index=40 + i, kind="speech" if i % 2 else "noise") for i in range(4)]
    o = oracle(tag)
    want = o.encode(wavs, trim=True)["codes_list"]
    m = model(tag, "fp32")
    got = m.encode([w.to(DEV) for w in wavs])["codes_list"]
    for a, b in zip(got, want):
        assert torch.equal(a.cpu().long(), b.long())
    wr = o.decode(want)["syn_wav_list"]
    wg = m.decode(got)["syn_wav_list"]
    for a, b in zip(wg, wr):
        assert _relerr(a.cpu().numpy(), b.numpy()) < TOL_FP32


# Reduced-precision encoder presets (BASELINE.json configs[1] "bf16", configs[4] "fp8 encoder GEMMs"): indices can no
# longer be bit-exact, so the tolerance is re-stated on the FSQ levels themselves.  A code packs 4 levels
# (8, 7, 6, 6 steps); a flipped level is a latent that sat near a rounding boundary.  Floors sit a few points under the
# values measured on MI355X with the synthetic checkpoint (real config, 8 x 5 s fresh utterances):
#   bf16: 98.7 % of levels equal (95.1 % of codes), none off by more than 1;   fp8: 87.0 % equal (58 % of codes), 99.8 % within 1.
#   fp8_fc1 (fc1 + GELU of the encoder alone in fp8; round 3's CPU simulation predicted 95.4 % equal levels): floors at bf16's class
#   measured on MI355X: 95.1 % equal (81.6 % of codes), all within 1
LEVEL_FLOORS = {"bf16": (0.97, 0.9995), "fp8": (0.84, 0.995), "fp8_fc1": (0.94, 0.9995)}


@pytest.mark.parametrize("precision", ["bf16", "fp8", "fp8_fc1"])
def test_reduced_precision_encoder_levels(precision):
    from simwhisper_codec_amd import synth
    tag = "real"
    wavs = [synth.synth_audio(80000 - 331 * i, index=700 + i, kind="speech" if i % 2 else "noise") for i in range(8)]
    o = oracle(tag)
    want = o.encode(wavs, trim=True)["codes_list"]
    m = model(tag, precision)
    got = m.encode([w.to(DEV) for w in wavs])["codes_list"]
    base = torch.tensor([1, 8, 56, 336])
    lev = torch.tensor([8, 7, 6, 6])
    same = within1 = total = codes_same = codes_total = 0
    for a, b in zip(got, want):
        a, b = a.cpu().long(), b.long()
        assert a.shape == b.shape
        la = (a[..., None] // base) % lev
        lb = (b[..., None] // base) % lev
        d = (la - lb).abs()
        same += int((d == 0).sum()); within1 += int((d <= 1).sum()); total += d.numel()
        codes_same += int((a == b).sum()); codes_total += a.numel()
    _report(f"levels/{precision}", equal=same / total, within1=within1 / total, codes_equal=codes_same / codes_total)
    lo_eq, lo_w1 = LEVEL_FLOORS[precision]
    assert same / total >= lo_eq, (same / total, within1 / total)
    assert within1 / total >= lo_w1, (same / total, within1 / total)
    # the decoder is the bf16 one in both presets: identical codes in, waveform within the bf16 tolerance
    wr = o.decode(want)["syn_wav_list"]
    wg = m.decode([c.to(DEV) for c in want])["syn_wav_list"]
    for x, y in zip(wg, wr):
        assert _relerr(x.cpu().numpy(), y.numpy()) < TOL_BF16


def test_no_cpu_fallback():
    from simwhisper_codec_amd.codec import AudioCodec
    from simwhisper_codec_amd._lib import SwcError
    m = AudioCodec(PARAMS["tiny"]())
    with pytest.raises(SwcError):
        m.encode([torch.zeros(2000)], device=torch.device("cpu"))


def test_window_batching_is_exact():
    """>= 3 windows: batching independent 30 s windows into one call must not change a single bit
    relative to the reference's serial window loop (max_rows_per_call = B)."""
    from simwhisper_codec_amd import synth
    m = model("tiny", "mixed")
    wavs = [synth.synth_audio(16000 * 63 + 321, index=90, kind="speech").to(DEV), synth.synth_audio(16000 * 41, index=91).to(DEV),
            synth.synth_audio(5000, index=92).to(DEV)]
    try:
        m.max_rows_per_call = len(wavs)       # serial: one window per call, as the reference
        c_ser = m.encode(wavs)["codes_list"]
        w_ser = m.decode(c_ser)["syn_wav_list"]
        m.max_rows_per_call = 64              # batched
        c_bat = m.encode(wavs)["codes_list"]
        w_bat = m.decode(c_bat)["syn_wav_list"]
    finally:
        m.max_rows_per_call = 64
    assert [c.shape[-1] for c in c_bat] == [(16000 * 63 + 321) // 1280, 16000 * 41 // 1280, 5000 // 1280]
    for a, b in zip(c_ser, c_bat):
        assert torch.equal(a, b)
    for a, b in zip(w_ser, w_bat):
        assert torch.equal(a, b)
    # and the oracle agrees on the codes of the long utterances
    o = oracle("tiny")
    want = o.encode([w.cpu() for w in wavs], trim=True)["codes_list"]
    for a, b in zip(c_bat, want):
        assert torch.equal(a.cpu().long(), b.long())


@pytest.mark.parametrize("tag,precision", [("tiny", "fp32"), ("real", "mixed")])
def test_vocos_halo_trim_is_exact(tag, precision):
    """decode() runs Vocos only on the kept 2000 frames (+80 halo) of a 3000-frame window: the
    kept samples must be bit-identical to computing all 3000 frames as the reference does."""
    from simwhisper_codec_amd import synth
    m = model(tag, precision)
    g = torch.Generator().manual_seed(11)
    n_codes = [375 + 40, 375, 251]  # 2 windows (full + 165 codes), exactly one full window + tail, 250 + 1
    codes = [torch.randint(0, 2016, (m.num_groups, n), generator=g).to(DEV) for n in n_codes]
    try:
        m.trim_vocos = False
        full = m.decode(codes)["syn_wav_list"]
        m.trim_vocos = True
        trim = m.decode(codes)["syn_wav_list"]
    finally:
        m.trim_vocos = True
    for a, b, n in zip(full, trim, n_codes):
        assert a.shape == b.shape == (n * 1280,)
        assert torch.equal(a, b)


def test_large_batch_equals_small_batches():
    """Rows are independent: a batch of 96 equal-length utterances (long persistent tile walks, several window rows)
    must give, bit for bit, what batches of 8 give."""
    from simwhisper_codec_amd import synth
    m = model("real", "mixed")
    wavs = [synth.synth_audio(16000 * 2, index=800 + i, kind="speech" if i % 3 else "noise").to(DEV) for i in range(96)]
    big_c = m.encode(wavs)["codes_list"]
    big_w = m.decode(big_c)["syn_wav_list"]
    for i0 in range(0, 96, 8):
        c = m.encode(wavs[i0:i0 + 8])["codes_list"]
        w = m.decode(c)["syn_wav_list"]
        for k in range(8):
            assert torch.equal(c[k], big_c[i0 + k])
            assert torch.equal(w[k], big_w[i0 + k])


def test_api_edge_cases():
    """inputs on the host, empty members, single utterance, float64 audio: same answers as the oracle."""
    from simwhisper_codec_amd import synth
    m = model("tiny", "mixed")
    o = oracle("tiny")
    wavs = [synth.synth_audio(9000, index=70, kind="speech"), torch.zeros(0), synth.synth_audio(2560, index=71).double()]
    got = m.encode(wavs)["codes_list"]                      # CPU inputs, default device
    want = o.encode([w.float() for w in wavs], trim=True)["codes_list"]
    assert [tuple(c.shape) for c in got] == [(8, 7), (8, 0), (8, 2)]
    for a, b in zip(got, want):
        assert a.is_cuda and torch.equal(a.cpu().long(), b.long())
    dec = m.decode([c.cpu().long() for c in got])["syn_wav_list"]   # CPU int64 codes in
    ref = o.decode(want)["syn_wav_list"]
    assert [w.shape[0] for w in dec] == [7 * 1280, 0, 2 * 1280]
    for a, b in zip(dec, ref):
        if b.numel():
            assert _relerr(a.cpu().numpy(), b.numpy()) < TOL_BF16
    one = m.encode([wavs[0].to(DEV)])["codes_list"]
    assert torch.equal(one[0], got[0])                      # batch-size independence of encode
    assert m.encode([])["codes_list"] == [] and m.decode([])["syn_wav_list"] == []
    assert m.decode([torch.zeros(8, 0, dtype=torch.long)])["syn_wav_list"][0].numel() == 0


def test_cli_end_to_end(tmp_path):
    """inference.py flags, file naming and PCM16 output on a tiny config with the synthetic checkpoint."""
    import yaml
    import inference
    from simwhisper_codec_amd import synth
    from simwhisper_codec_amd.wavio import load_audio, save_audio
    cfg = tmp_path / "tiny.yaml"
    cfg.write_text(yaml.safe_dump({"generator_params": PARAMS["tiny"]()}))
    ind, outd = tmp_path / "in" / "sub", tmp_path / "out"
    ind.mkdir(parents=True)
    lens = {"a.wav": 16000, "b.wav": 5000, "c.wav": 40000}
    for i, (name, n) in enumerate(lens.items()):
        save_audio(str(ind / name), synth.synth_audio(n, index=80 + i, kind="speech").reshape(1, -1), 16000)
    # the PCM16 samples of c.wav once more as a FLAC file (native decoder, csrc/swc_flac.c): same batch, same output bytes
    import numpy as np
    import flac_encode
    from simwhisper_codec_amd.wavio import _read_wav
    pcm = np.round(_read_wav(str(ind / "c.wav"))[0] * 32768.0).astype(np.int64)
    (ind / "d.flac").write_bytes(flac_encode.encode(pcm, 16000, 16, blocksize=4096,
                                                    plan=lambda fi, c: dict(kind=("lpc", 8), porder=2) if c is not None else 0))
    inference.main(["--config_path", str(cfg), "--synthetic_checkpoint", "--device", "cuda", "--batch_size", "2",
                    "--input_dir", str(tmp_path / "in"), "--output_dir", str(outd), "--precision", "mixed"])
    assert (outd / "d.wav").read_bytes() == (outd / "c.wav").read_bytes()
    m = model("tiny", "mixed")
    for name, n in lens.items():
        y = load_audio(str(outd / name), 16000).reshape(-1)
        assert y.shape[0] == (n // 1280) * 1280
    # batch of the first two files (sorted order a, b) reproduces the CLI's first batch exactly up to PCM16 rounding
    wavs = [load_audio(str(ind / k), 16000).reshape(-1).to(DEV) for k in ("a.wav", "b.wav")]
    ref = m.decode(m.encode(wavs)["codes_list"])["syn_wav_list"]
    for k, r in zip(("a.wav", "b.wav"), ref):
        y = load_audio(str(outd / k), 16000).reshape(-1)
        assert (y - r.cpu().clamp(-1, 1)).abs().max().item() <= 1.0 / 32767 + 1e-6
        # the CLI moves 16-bit samples over PCIe and converts on the GPU (swc_pcm16_to_f32 / swc_f32_to_pcm16): the file must be
        # byte for byte the one the host conversion of the same waveform writes
        save_audio(str(tmp_path / "host.wav"), r.reshape(1, -1), 16000)
        assert (outd / k).read_bytes() == (tmp_path / "host.wav").read_bytes()


def test_vocos_two_stream_chains_are_exact():
    """vocos_streams = 2 runs the ConvNeXt blocks of the two half-batches as two chains on two streams (out of phase, so
    that one half's HBM phases fall into the other's compute): every frame's arithmetic is unchanged, the waveforms must
    be bit-identical to the single-launch-per-block form."""
    m = model("real", "mixed")
    g = torch.Generator().manual_seed(5)
    codes = [torch.randint(0, 2016, (m.num_groups, 120 - (i % 3)), generator=g).to(DEV) for i in range(24)]  # 24 x ~960 frames
    try:
        m.vocos_streams = 1
        one = m.decode(codes)["syn_wav_list"]
        m.vocos_streams = 2
        two = m.decode(codes)["syn_wav_list"]
        two_again = m.decode(codes)["syn_wav_list"]
    finally:
        m.vocos_streams = 1
    for a, b, c in zip(one, two, two_again):
        assert torch.equal(a, b) and torch.equal(b, c)


@pytest.mark.parametrize("tag,precision", [("tiny", "fp32"), ("real", "mixed")])
def test_length_bucketing_is_exact(tag, precision):
    """Ragged batches: encode() / decode() run rows of similar length together (length_groups) instead of every row at the
    longest row's length; decode rows see min(L, n_i + 64) code frames of the batch padding (the receptive field of the
    reference's un-masked up-sampler is +-59).  Codes and every kept sample must be bit-identical to the one-call form,
    and the codes equal the oracle's."""
    from simwhisper_codec_amd import synth
    from simwhisper_codec_amd.codec import length_groups
    m = model(tag, precision)
    secs = ([27.3, 1.1, 9.0, 22.5, 0.05, 3.3, 12.7, 2.0, 30.0, 41.0] if tag == "tiny" else
            [21.5, 20.2, 22.3, 19.1, 18.4, 21.0, 23.9, 20.7] + [1.1, 2.0, 0.5, 3.3, 1.7, 2.4, 0.9, 1.3] * 2)
    wavs = [synth.synth_audio(int(16000 * t) + 13 * i, index=950 + i, kind="speech" if i % 2 else "noise").to(DEV)
            for i, t in enumerate(secs)]
    assert len(length_groups(sorted([int(16000 * t) // 320 for t in secs], reverse=True))) > 1  # the case is really ragged
    keep_rows, keep_pack, keep_layer = m.fused_mlp_min_rows, m.varlen_packing, m.fused_layer_mlp_min_rows
    try:
        m.varlen_packing = False  # (with packing on, ragged calls are not grouped at all)
        # the same kernels on both sides (the fused ConvNeXt kernel is chosen by the number of Vocos frames of a call and
        # rounds its bf16 intermediate in another order than the two-GEMM form; likewise swc_mlp_block by the number of
        # decoder tokens): bit-identical
        m.fused_mlp_min_rows = m.fused_layer_mlp_min_rows = 1 << 40
        m.length_bucketing = False
        c0 = m.encode(wavs)["codes_list"]
        w0 = m.decode(c0)["syn_wav_list"]
        m.length_bucketing = True
        c1 = m.encode(wavs)["codes_list"]
        w1 = m.decode(c1)["syn_wav_list"]
        # default kernel choice: codes identical, waveforms within the bf16 decode tolerance
        m.fused_mlp_min_rows, m.fused_layer_mlp_min_rows = keep_rows, keep_layer
        w2 = m.decode(c1)["syn_wav_list"]
    finally:
        m.length_bucketing = True
        m.fused_mlp_min_rows, m.varlen_packing, m.fused_layer_mlp_min_rows = keep_rows, keep_pack, keep_layer
    for a, b in zip(c0 + w0, c1 + w1):
        assert a.shape == b.shape and torch.equal(a, b)
    for a, b in zip(w1, w2):
        if a.numel():
            assert _relerr(b.cpu().numpy(), a.cpu().numpy()) < (TOL_FP32 if precision == "fp32" else TOL_BF16)
    if tag == "tiny":
        want = oracle(tag).encode([w.cpu() for w in wavs], trim=True)["codes_list"]
        for a, b in zip(c1, want):
            assert torch.equal(a.cpu().long(), b.long())


@pytest.mark.parametrize("tag,precision", [("tiny", "mixed"), ("real", "mixed"), ("real", "bf16")])
def test_valid_token_packing_is_exact(tag, precision):
    """Ragged calls run the encoder / decoder transformers on the valid tokens only (packed rows, swc_pack_rows +
    row_start of swc_attention16 / swc_layernorm): bit-identical codes and waveforms to the padded layout, and the
    reference's codes (tiny config, oracle)."""
    from simwhisper_codec_amd import synth
    m = model(tag, precision)
    secs = [9.0, 1.1, 4.3, 0.3, 7.7, 2.0, 0.05, 5.5]
    wavs = [synth.synth_audio(int(16000 * t) + 29 * i, index=970 + i, kind="speech" if i % 2 else "noise").to(DEV)
            for i, t in enumerate(secs)]
    keep = (m.varlen_packing, m.length_bucketing)
    try:
        m.length_bucketing = False
        m.varlen_packing = False
        c0 = m.encode(wavs)["codes_list"]
        w0 = m.decode(c0)["syn_wav_list"]
        m.varlen_packing = True
        packed_calls = []
        from simwhisper_codec_amd import ops
        real = ops.pack_rows
        ops.pack_rows = lambda *a, **k: (packed_calls.append(1), real(*a, **k))[1]
        try:
            c1 = m.encode(wavs)["codes_list"]
            w1 = m.decode(c1)["syn_wav_list"]
        finally:
            ops.pack_rows = real
    finally:
        m.varlen_packing, m.length_bucketing = keep
    assert len(packed_calls) == 2  # encoder and decoder transformers both ran packed
    for a, b in zip(c0 + w0, c1 + w1):
        assert a.shape == b.shape and torch.equal(a, b)
    if tag == "tiny":
        want = oracle(tag).encode([w.cpu() for w in wavs], trim=True)["codes_list"]
        for a, b in zip(c1, want):
            assert torch.equal(a.cpu().long(), b.long())

@pytest.mark.parametrize("precision", ["mixed", "bf16"])
def test_ragged_vocos_tile_skipping_is_exact(precision):
    """decode() of a ragged batch: the ConvNeXt blocks skip the 128-frame tiles beyond a row's kept frames + halo
    (swc_convnext_block t_limit).  Every returned sample is bit-identical to the run that computes all padded frames."""
    from simwhisper_codec_amd import ops, synth
    m = model("real", precision)
    secs = [27.0, 2.2, 11.3, 0.7, 19.9, 6.1, 24.5, 3.3, 14.0, 1.0]   # 10 rows x 2700 frames: the fused block kernel is used
    wavs = [synth.synth_audio(int(16000 * t) + 31 * i, index=1200 + i, kind="speech" if i % 2 else "noise").to(DEV)
            for i, t in enumerate(secs)]
    codes = m.encode(wavs)["codes_list"]
    keep = m.ragged_vocos
    seen = []
    real = ops.convnext_block
    ops.convnext_block = lambda *a, **k: (seen.append(k.get("t_limit") is not None), real(*a, **k))[1]
    try:
        m.ragged_vocos = False
        w0 = m.decode(codes)["syn_wav_list"]
        n0 = len(seen)
        m.ragged_vocos = True
        w1 = m.decode(codes)["syn_wav_list"]
    finally:
        ops.convnext_block = real
        m.ragged_vocos = keep
    assert n0 > 0 and not any(seen[:n0]) and all(seen[n0:]) and len(seen) == 2 * n0   # fused blocks ran, with limits the second time
    for a, b in zip(w0, w1):
        assert a.shape == b.shape and torch.isfinite(b).all() and torch.equal(a, b)



@pytest.mark.parametrize("levels", [[5, 5, 5, 4], [16, 3, 2, 9]])
def test_other_fsq_level_sets_against_the_oracle(levels):
    """The reference's quantiser is config-driven (quantizer.py:47-120); the shipped YAML uses [8, 7, 6, 6].  Another level set
    on the tiny config: encode() codes bit for bit and decode() waveforms within the fp32 tolerance of the CPU oracle."""
    from oracle.ref_cpu import Oracle
    from simwhisper_codec_amd import synth
    from simwhisper_codec_amd.codec import AudioCodec
    gp = PARAMS["tiny"]()
    gp["quantizer"] = dict(gp["quantizer"], num_levels_per_group=list(levels))
    sd = synth.synth_state_dict(gp)
    m = AudioCodec(gp, precision="fp32")
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    o = Oracle(gp, sd)
    wavs = [synth.synth_audio(n, index=60 + i, kind="speech") for i, n in enumerate([16000, 7000, 23040])]
    want = o.encode(wavs, trim=True)["codes_list"]
    got = m.encode([w.to(DEV) for w in wavs])["codes_list"]
    n_code = 1
    for v in levels:
        n_code *= v
    for a, b in zip(got, want):
        assert a.shape == b.shape and int(a.max()) < n_code
        assert torch.equal(a.cpu().long(), b.long())
    assert len({int(v) for c in got for v in c.flatten().tolist()}) > 8      # the codebook is really used
    ww = o.decode(want)["syn_wav_list"]
    gw = m.decode(got)["syn_wav_list"]
    for a, b in zip(gw, ww):
        assert _relerr(a.float().cpu().numpy(), b.numpy()) < TOL_FP32
