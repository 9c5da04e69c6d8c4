"""Parity of the kernels the METRIC runs, at the metric's own shapes.

The golden fixtures are a few seconds long: at that size `_vocos` takes the two-GEMM form of a ConvNeXt block, the GEMMs
run their small-grid geometries and no persistent tile walk happens.  BASELINE.json's metric shape (32 x 10 s, and
32 x 30 s for configs[2]) instead runs `swc_convnext_block` (one kernel per block, >= 20 480 Vocos frames), the 8-wave
256 / 192-row GEMM geometries with several tiles per workgroup, window batching and the two-chain ConvNeXt launch.  Here:

  * the fused ConvNeXt kernel is FORCED on the reference-generated fixtures (`fused_mlp_min_rows = 0`; it handles partial
    tiles) and compared with the reference's own stage output / waveforms (tests/golden, written by the reference itself);
    the same for the fused transformer-layer kernels of the decoder (`fused_layer_mlp_min_rows = 0`: swc_layer_tail, swc_mlp_block; round 4);
  * the metric shapes run as `bench.py` runs them (same generator, same seed) and single rows are compared with the CPU
    oracle run on that row alone: uniform batches, so a row's result does not depend on the batch — codes bit for bit,
    waveform within the bf16-decode tolerance (modules.py:1229-1248, model.py:244-373).
"""
import os
import sys

import numpy as np
import pytest
import torch

from common import ROOT, golden, oracle
from test_parity_gpu import TOL_BF16, _relerr, _report, model
from test_stages_gpu import TOL as STAGE_TOL, _to_f32

pytestmark = pytest.mark.gpu
DEV = "cuda"


class _Spy:
    """counts calls of ops.<name> while active"""

    def __init__(self, name):
        from simwhisper_codec_amd import ops
        self.ops, self.name, self.calls, self.kw = ops, name, 0, []

    def __enter__(self):
        self.real = getattr(self.ops, self.name)

        def wrapped(*a, **k):
            self.calls += 1
            self.kw.append({n: k[n] for n in ("B", "T") if n in k})
            return self.real(*a, **k)
        setattr(self.ops, self.name, wrapped)
        return self

    def __exit__(self, *exc):
        setattr(self.ops, self.name, self.real)
        return False


@pytest.mark.parametrize("name", ["single", "ragged"])
def test_stage_vocos_fused_block_kernel(name):
    """Vocos + ISTFT fed the reference's decoder output, every ConvNeXt block on the fused kernel, against the reference's
    own waveform (st_y) at the stage's tolerance."""
    g, m = golden("real", name), model("real", "mixed")
    dm = torch.from_numpy(g["st_dec_mel"]).transpose(1, 2).contiguous().to(DEV)  # [B, Tv, 80]
    B, Tv, _ = dm.shape
    keep = m.fused_mlp_min_rows
    try:
        m.fused_mlp_min_rows = 0
        with torch.cuda.device(0), torch.inference_mode(), _Spy("convnext_block") as spy, _Spy("dwconv7_ln") as old:
            P = m._packed()
            y = m._vocos(m._cast(dm, P.ddt), B, Tv, P).cpu().numpy()
    finally:
        m.fused_mlp_min_rows = keep
    assert spy.calls == len(P.blocks) == 24 and old.calls == 0
    assert y.shape == g["st_y"].shape and np.isfinite(y).all()
    e = _relerr(y, g["st_y"])
    _report(f"stage/y_fused/real/{name}/mixed", rel_err=e, frames=B * Tv)
    assert e < STAGE_TOL["y"][1], e


@pytest.mark.parametrize("fusion", [2, 1])
@pytest.mark.parametrize("name", ["single", "ragged"])
def test_stage_decoder_fused_mlp_kernel(name, fusion):
    """The 12 decoder layers + deconvs fed the reference's up-sampler output, everything behind the attention of every layer on
    swc_layer_tail (fusion 2: out-proj + MLP sub-block + both LayerNorms in one kernel) or the MLP sub-block on swc_mlp_block
    behind an out-proj GEMM (fusion 1), against the reference's own decoder output (st_dec_mel) at the stage's tolerance.
    `ragged` runs the packed token layout."""
    g, m = golden("real", name), model("real", "mixed")
    up = torch.from_numpy(g["st_up"]).transpose(1, 2).contiguous().to(DEV)  # [B, 4T, D]
    B, Tt, D = up.shape
    lat = [int(v) for v in g["st_code_lens"]]
    keep, keep_f = m.fused_layer_mlp_min_rows, m.layer_fusion
    try:
        m.fused_layer_mlp_min_rows, m.layer_fusion = 0, fusion
        if fusion != keep_f:
            m._pk = None   # the operand streams are packed for one form (read at pack time)
        with torch.cuda.device(0), torch.inference_mode(), _Spy("mlp_block") as spy1, _Spy("layer_tail") as spy2, \
                _Spy("layernorm") as ln:
            P = m._packed()
            mel = _to_f32(m._decoder(up.reshape(B * Tt, D).clone(), lat, B, Tt, P), P.vin)[..., :P.vin].cpu().numpy()
    finally:
        m.fused_layer_mlp_min_rows, m.layer_fusion = keep, keep_f
        if fusion != keep_f:
            m._pk = None
    assert (spy1.calls, spy2.calls) == ((0, 12) if fusion == 2 else (12, 0)) and len(P.dec_layers) == 12
    assert ln.calls == 2   # the first layer's self_attn_layer_norm and the decoder's final LayerNorm; 24 are folded away
    e = _relerr(mel, g["st_dec_mel"].transpose(0, 2, 1))
    _report(f"stage/dec_mel_fused_mlp/real/{name}/mixed/fusion{fusion}", rel_err=e, tokens=B * Tt)
    assert e < STAGE_TOL["dec_mel"][1], e


@pytest.mark.parametrize("name", ["single", "ragged", "zeros", "short", "chunked"])
def test_decode_waveform_fused_block_kernel(name):
    """decode() of the reference's codes with the fused ConvNeXt kernel AND the fused transformer-MLP kernel forced (incl.
    the two-window 22 s fixture, ragged rows with tile limits and packed tokens, all-zero codes) against the reference's
    waveforms."""
    g, m = golden("real", name), model("real", "mixed")
    nutt = len(g["spec_n"])
    codes = [torch.from_numpy(g[f"codes_{i}"]).to(DEV) for i in range(nutt)]
    keep, keep_l = m.fused_mlp_min_rows, m.fused_layer_mlp_min_rows
    try:
        m.fused_mlp_min_rows = m.fused_layer_mlp_min_rows = 0
        with _Spy("convnext_block") as spy, _Spy("dwconv7_ln") as old, _Spy("layer_tail") as mlp:
            dec = m.decode(codes, overlap_seconds=10)
    finally:
        m.fused_mlp_min_rows, m.fused_layer_mlp_min_rows = keep, keep_l
    any_frames = any(int(n) // 1280 > 0 for n in g["spec_n"])
    assert old.calls == 0 and (spy.calls >= 24 if any_frames else spy.calls == 0)
    assert mlp.calls >= 12 if any_frames else mlp.calls == 0
    worst = 0.0
    for i, w in enumerate(dec["syn_wav_list"]):
        w = w.float().cpu().numpy()
        assert w.shape[0] == (int(g["spec_n"][i]) // 1280) * 1280 and np.isfinite(w).all()
        if f"wav_{i}" in g:
            e = _relerr(w, g[f"wav_{i}"])
        else:
            e = _relerr(w[::7], g[f"wav_stride7_{i}"])
            en = (w.reshape(-1, 1280).astype(np.float64) ** 2).sum(1)
            assert np.allclose(en, g[f"wav_energy_{i}"], rtol=20 * TOL_BF16, atol=1e-6)
        worst = max(worst, e)
    _report(f"decode_fused/real/{name}/mixed", rel_err=worst, block_launches=spy.calls)
    assert worst < TOL_BF16, worst


def _bench_inputs(n_utt, seconds):
    sys.path.insert(0, ROOT)
    import bench
    return bench.bench_inputs(n_utt, int(seconds * 16000))


def _check_rows_against_oracle(tag, wavs, codes, wav_out, rows):
    o = oracle("real")
    tot = 0
    for r in rows:
        want_c = o.encode([wavs[r]], trim=True)["codes_list"][0]
        got_c = codes[r].cpu().long()
        assert got_c.shape == want_c.shape, (got_c.shape, want_c.shape)
        mism = int((got_c != want_c.long()).sum())
        want_w = o.decode([want_c])["syn_wav_list"][0].numpy()
        got_w = wav_out[r].float().cpu().numpy()
        assert got_w.shape == want_w.shape
        e = _relerr(got_w, want_w)
        _report(f"metric_shape/{tag}/row{r}", code_mismatch=mism, total=want_c.numel(), wav_rel_err=e)
        assert mism == 0, f"{tag} row {r}: {mism} of {want_c.numel()} codes differ from the oracle"
        assert e < TOL_BF16, (tag, r, e)
        tot += want_c.numel()
    return tot


def test_metric_shape_32x10s_against_oracle():
    """BASELINE.json's metric shape exactly as bench.py runs it (32 x 10 s, preset `mixed`, seed-1234 inputs): the first and
    the last utterance against the CPU oracle run on that utterance alone."""
    m = model("real", "mixed")
    wavs = _bench_inputs(32, 10.0)
    dw = [w.to(DEV) for w in wavs]
    with _Spy("convnext_block") as spy, _Spy("dwconv7_ln") as old, _Spy("layer_tail") as mlp:
        codes = m.encode(dw, overlap_seconds=10)["codes_list"]
        out = m.decode(codes, overlap_seconds=10)["syn_wav_list"]
    assert spy.calls == 24 and old.calls == 0          # one fused launch per block over 32 000 frames: the metric's kernels
    assert spy.kw[0] == {"B": 32, "T": 1000}
    assert mlp.calls == 12                             # the decoder's 12 layers (16 000 tokens): everything behind the attention on swc_layer_tail
    assert all(tuple(c.shape) == (8, 125) for c in codes) and all(w.shape[0] == 160000 for w in out)
    _check_rows_against_oracle("32x10s", wavs, codes, out, rows=(0, 31))


def test_metric_shape_32x30s_against_oracle():
    """BASELINE.json configs[2] (32 x 30 s: two windows per utterance batched as 64 rows, 1500-token encoder windows,
    halo-trimmed Vocos, ConvNeXt blocks as two chains on two streams): one utterance against the CPU oracle."""
    m = model("real", "mixed")
    wavs = _bench_inputs(32, 30.0)
    dw = [w.to(DEV) for w in wavs]
    with _Spy("convnext_block") as spy:
        codes = m.encode(dw, overlap_seconds=10)["codes_list"]
        out = m.decode(codes, overlap_seconds=10)["syn_wav_list"]
    assert spy.calls >= 48                              # two windows' worth of fused block launches (two chains for window 0)
    assert all(tuple(c.shape) == (8, 375) for c in codes) and all(w.shape[0] == 480000 for w in out)
    _check_rows_against_oracle("32x30s", wavs, codes, out, rows=(17,))


# --- the remaining BASELINE.json configs at their own shapes (round 4; rounds 1-3 compared these presets with the oracle
# at 8 x 5 s only, where the small-grid geometries run) -----------------------------------------------------------------
def _levels_against_oracle(tag, precision, wavs, got_codes, rows):
    """FSQ-level agreement of `rows` with the CPU oracle run on each row alone (the reduced-precision presets' tolerance is
    stated on the levels: tests/test_parity_gpu.py LEVEL_FLOORS), and waveform-given-the-oracle's-codes for the same rows."""
    from test_parity_gpu import LEVEL_FLOORS
    o = oracle("real")
    base, lev = torch.tensor([1, 8, 56, 336]), torch.tensor([8, 7, 6, 6])
    same = within1 = total = 0
    want_codes = {}
    for r in rows:
        want = o.encode([wavs[r]], trim=True)["codes_list"][0].long()
        got = got_codes[r].cpu().long()
        assert got.shape == want.shape
        d = (((got[..., None] // base) % lev) - ((want[..., None] // base) % lev)).abs()
        same += int((d == 0).sum()); within1 += int((d <= 1).sum()); total += d.numel()
        want_codes[r] = want
    lo_eq, lo_w1 = LEVEL_FLOORS[precision]
    _report(f"metric_shape/{tag}/levels", equal=same / total, within1=within1 / total, rows=len(rows))
    assert same / total >= lo_eq and within1 / total >= lo_w1, (same / total, within1 / total)
    return o, want_codes


def test_config1_8x10s_bf16_against_oracle():
    """BASELINE.json configs[1]: batch = 8 x 10 s, preset `bf16` (bf16 encoder AND decoder): 4 000-token launches on the
    small-grid bf16 geometries and the two-GEMM form of the ConvNeXt blocks (8 000 frames < 20 480).  Rows 0 and 7: FSQ
    levels within the preset's floors; the decoder, given the ORACLE's codes for the whole batch, within the bf16 tolerance."""
    m = model("real", "bf16")
    wavs = _bench_inputs(8, 10.0)
    with _Spy("convnext_block") as fused, _Spy("dwconv7_ln") as two, _Spy("layer_tail") as mlp:
        codes = m.encode([w.to(DEV) for w in wavs], overlap_seconds=10)["codes_list"]
        o, want = _levels_against_oracle("8x10s_bf16", "bf16", wavs, codes, rows=(0, 7))
        # decode the oracle's codes of rows 0 and 7 inside a batch of 8 (the other rows: this preset's own codes)
        mix = [want[r].to(DEV).to(codes[r].dtype) if r in want else codes[r] for r in range(8)]
        out = m.decode(mix, overlap_seconds=10)["syn_wav_list"]
    assert fused.calls == 0 and two.calls == 24 and mlp.calls == 0   # the small-batch forms are what ran
    for r in (0, 7):
        ref = o.decode([want[r]])["syn_wav_list"][0].numpy()
        e = _relerr(out[r].float().cpu().numpy(), ref)
        _report(f"metric_shape/8x10s_bf16/row{r}", wav_rel_err=e)
        assert e < TOL_BF16, (r, e)


def test_config4_32x10s_fp8_against_oracle():
    """BASELINE.json configs[4]: batch = 32 x 10 s, preset `fp8` (encoder-transformer linears on the block-scaled fp8 MFMA
    with 256-row tiles, bf16 everywhere else, fused ConvNeXt and fused decoder MLP).  Rows 0 and 31: FSQ levels within the
    preset's floors; waveform given the oracle's codes within the bf16 tolerance."""
    m = model("real", "fp8")
    wavs = _bench_inputs(32, 10.0)
    with _Spy("convnext_block") as fused, _Spy("layer_tail") as mlp:
        codes = m.encode([w.to(DEV) for w in wavs], overlap_seconds=10)["codes_list"]
        o, want = _levels_against_oracle("32x10s_fp8", "fp8", wavs, codes, rows=(0, 31))
        mix = [want[r].to(DEV).to(codes[r].dtype) if r in want else codes[r] for r in range(32)]
        out = m.decode(mix, overlap_seconds=10)["syn_wav_list"]
    assert fused.calls == 24 and mlp.calls == 12
    for r in (0, 31):
        ref = o.decode([want[r]])["syn_wav_list"][0].numpy()
        e = _relerr(out[r].float().cpu().numpy(), ref)
        _report(f"metric_shape/32x10s_fp8/row{r}", wav_rel_err=e)
        assert e < TOL_BF16, (r, e)
