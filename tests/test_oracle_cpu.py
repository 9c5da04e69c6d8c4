"""Pin the CPU oracle (oracle/ref_cpu.py) to outputs of the reference itself
(tests/golden/*.npz, written by oracle/make_golden.py in the build container) and to
the analytic known answers of SURVEY.md §8c."""
import os

import numpy as np
import pytest
import torch

from common import GOLD, ROOT, golden, golden_audio, oracle, state_dict

TAGS = ["tiny"] + (["real"] if os.environ.get("SWC_REAL_ORACLE_TESTS", "1") == "1" else [])
# fp32 on the same CPU/BLAS: the oracle re-orders nothing on purpose, but folded weight-norm
# and fused calls move the last bits; codes must still be identical.
WAV_TOL = 2e-4


@pytest.mark.parametrize("tag", TAGS)
def test_facts(tag):
    f = golden(tag, "facts")
    sd = state_dict(tag)
    assert len(sd) == int(f["n_keys"]) and sum(v.numel() for v in sd.values()) == int(f["n_params"])
    from oracle.ref_cpu import slaney_mel_filters, fold_weight_norm
    assert np.abs(slaney_mel_filters() - f["mel_filters"]).max() < 1e-12
    o = oracle(tag)
    assert np.array_equal(o.filt.numpy(), f["aa_filter"])
    w = fold_weight_norm(sd["downsample.to_latent.weight_g"], sd["downsample.to_latent.weight_v"])
    assert np.abs(w.numpy() - f["wn_folded"]).max() < 1e-6


def test_real_checkpoint_size():
    if "real" not in TAGS:
        pytest.skip("real config disabled")
    f = golden("real", "facts")
    assert int(f["n_keys"]) == 711  # 711 tensors
    assert abs(int(f["n_params"]) / 1e6 - 293.63) < 0.01


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("name", ["single", "ragged"])
def test_stages(tag, name):
    g = golden(tag, name)
    o = oracle(tag)
    wavs = golden_audio(g)
    mel, ml = o.logmel(wavs)
    mf = g["st_mel"].shape[-1]
    assert np.array_equal(ml.numpy(), g["st_mel_lens"])
    assert np.abs(mel[:, :, :mf].numpy() - g["st_mel"]).max() < 2e-5
    assert np.abs(mel[:, :, -1].numpy() - g["st_mel_tail"]).max() < 2e-5
    gm = torch.zeros(len(wavs), 80, 3000)
    gm[:, :, :] = torch.from_numpy(g["st_mel_tail"])[:, :, None]
    gm[:, :, :mf] = torch.from_numpy(g["st_mel"])
    # from the golden mel onwards, stage by stage with golden inputs
    eo, el = o.encoder(gm, ml)
    te = g["st_enc"].shape[-1]
    assert np.abs(eo[:, :, :te].numpy() - g["st_enc"]).max() < 1e-4
    assert float(eo[:, :, te:].abs().max()) == 0.0
    eo_t, _ = o.encoder(gm, ml, trim=True)   # the exact-trim claim
    assert np.abs(eo_t.numpy() - eo.numpy()).max() < 2e-5
    z, zl = o.downsample(eo, el)
    assert np.array_equal(zl.numpy(), g["st_code_lens"])
    assert np.abs(z.numpy() - g["st_z"]).max() < 2e-4
    zq, codes = o.fsq_encode(torch.from_numpy(g["st_z"]), zl)  # bit-exact on identical z
    assert np.array_equal(codes.numpy(), g["st_codes"])
    assert np.array_equal(zq.numpy(), g["st_zq"])
    T = int(zl.max())
    zq2 = o.fsq_decode(torch.from_numpy(g["st_codes"][:, :, :T]).long(), zl)
    assert np.array_equal(zq2.numpy(), g["st_zq"][:, :, :T])
    up = o.upsample(zq2)
    assert np.abs(up.numpy() - g["st_up"]).max() < 1e-4
    dm, dl = o.decoder(torch.from_numpy(g["st_up"]), zl * 4)
    assert np.abs(dm.numpy() - g["st_dec_mel"]).max() < 2e-4
    y = o.vocos(torch.from_numpy(g["st_dec_mel"]))
    assert np.abs(y.numpy() - g["st_y"]).max() < WAV_TOL


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("name", ["single", "ragged", "zeros", "short", "chunked"])
def test_end_to_end(tag, name):
    if tag == "real" and name == "chunked" and os.environ.get("SWC_SLOW", "0") != "1":
        pytest.skip("real-config 22 s case takes ~1 min of CPU; run with SWC_SLOW=1")
    g = golden(tag, name)
    o = oracle(tag)
    wavs = golden_audio(g)
    enc = o.encode(wavs)
    for i, c in enumerate(enc["codes_list"]):
        assert c.shape == g[f"codes_{i}"].shape
        assert np.array_equal(c.numpy(), g[f"codes_{i}"]), f"utt {i}: codes differ"
    dec = o.decode([torch.from_numpy(g[f"codes_{i}"]).long() for i in range(len(wavs))])
    for i, w in enumerate(dec["syn_wav_list"]):
        w = w.numpy()
        assert w.shape[0] == (int(g["spec_n"][i]) // 1280) * 1280  # the length contract of docs/assets/codec
        if f"wav_{i}" in g:
            assert np.abs(w - g[f"wav_{i}"]).max() < WAV_TOL if w.size else True
        else:
            assert np.abs(w[::7] - g[f"wav_stride7_{i}"]).max() < WAV_TOL


@pytest.mark.parametrize("tag", TAGS)
def test_forward(tag):
    from simwhisper_codec_amd import synth
    g = golden(tag, "forward")
    T, lens = int(g["T"]), g["lens"]
    mel = torch.from_numpy(synth._uniform("forward/mel", len(lens) * 80 * T, 77).reshape(len(lens), 80, T) * 0.8 + 0.2)
    r = oracle(tag).forward({"mel_features": mel, "mel_lens": torch.from_numpy(lens)})
    assert np.array_equal(r["audio_lengths"].numpy(), g["audio_lengths"])
    assert np.abs(r["reconstructed_audio"][:, 0].numpy() - g["audio"]).max() < WAV_TOL


def test_fsq_known_answers():
    o = oracle("tiny")
    scale, offset, shift = o._fsq_consts()
    assert np.allclose(scale.view(-1).numpy(), [3.4965, 2.997, 2.4975, 2.4975], atol=1e-6)
    assert np.allclose(shift.view(-1).numpy(), [0.14398292, 0.0, 0.20291847, 0.20291847], atol=1e-7)
    assert torch.round(torch.tensor([0.5, 1.5, 2.5, -0.5])).tolist() == [0.0, 2.0, 2.0, -0.0]
    idx = torch.arange(2016).view(1, 1, -1).expand(8, 1, -1).contiguous()
    lens = torch.tensor([2016])
    zq = o.fsq_decode(idx, lens)
    scale_i = torch.tensor([4.0, 3.0, 3.0, 3.0]).view(1, 4, 1)
    back = (((zq[:, :4] * scale_i + scale_i) * o.base).sum(1)).to(torch.int64)
    assert torch.equal(back[0], torch.arange(2016))


def test_length_laws():
    o = oracle("tiny")
    from simwhisper_codec_amd import synth
    for n in (1280, 2000, 16000, 50001):
        r = o.tokenize(synth.synth_audio(n).view(1, 1, -1), torch.tensor([n]), trim=True)
        mel_len = -(-n // 160)
        assert int(r["codes_lengths"][0]) == -(-(mel_len // 2) // 4)
        assert r["codes"].shape == (8, 1, 375)


@pytest.mark.skipif(not os.path.isdir("/root/reference/audiocodec"), reason="the reference checkout exists in the build container only")
def test_golden_generator_reproduces_committed_fixture(tmp_path):
    """The recipe that pins the oracle must stay runnable: oracle/make_golden.py imports THE REFERENCE (not the repo's
    own `audiocodec` drop-in package, which once shadowed it) and regenerates tiny_single bit for bit."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "make_golden.py"), "--only", "tiny", "--cases", "single",
                        "--out", str(tmp_path)], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    new = np.load(tmp_path / "tiny_single.npz", allow_pickle=False)
    old = np.load(os.path.join(ROOT, "tests", "golden", "tiny_single.npz"), allow_pickle=False)
    assert set(new.files) == set(old.files)
    for k in old.files:
        assert np.array_equal(new[k], old[k]), k
