"""Generate tests/golden/*.npz by running THE REFERENCE ITSELF (build container only).

    python oracle/make_golden.py [--ref /root/reference] [--only tiny|real]

The reference (pure Python on PyTorch) is imported from its read-only checkout with one
in-memory shim: `audiocodec/nn/modules.py:21` imports two private torchaudio helpers that
only the never-instantiated IMDCTSymExpHead uses, and torchaudio is not installed, so
empty `torchaudio*` modules are registered before the import (recipe: SURVEY.md §8c).
Nothing of the reference is copied: the fixtures hold inputs specs and output tensors.

Weights: the closed-form synthetic checkpoint of simwhisper_codec_amd.synth (the trained
checkpoint needs a network fetch).  Loading it with strict=True also proves the key set.
The GPU box never runs this script and never sees /root/reference.
"""
import argparse
import os
import sys
import time
import types

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simwhisper_codec_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def import_reference(ref_root):
    """Import `audiocodec.model.AudioCodec` FROM THE REFERENCE CHECKOUT.  The repo root holds its own regular package
    `audiocodec/` (the drop-in import path of the product); the reference's `audiocodec/` is a namespace package and
    would lose to it whatever the order of sys.path.  So: forget any `audiocodec*` module already imported, take the
    repo root (and the cwd entry) off sys.path for the duration of the import, and check where the module came from."""
    ref_root = os.path.realpath(ref_root)
    import transformers  # noqa: F401
    import transformers.activations  # noqa: F401
    import transformers.audio_utils  # noqa: F401
    import transformers.feature_extraction_sequence_utils  # noqa: F401
    for name in ("torchaudio", "torchaudio.functional", "torchaudio.functional.functional"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["torchaudio.functional.functional"]._hz_to_mel = None
    sys.modules["torchaudio.functional.functional"]._mel_to_hz = None
    for name in [m for m in sys.modules if m == "audiocodec" or m.startswith("audiocodec.")]:
        del sys.modules[name]
    saved = list(sys.path)
    here = {os.path.realpath(ROOT), os.path.realpath(os.getcwd())}
    sys.path[:] = [ref_root] + [p for p in saved if os.path.realpath(p or os.getcwd()) not in here]
    try:
        import audiocodec.model as ref_model
    finally:
        sys.path[:] = saved
    origin = os.path.realpath(ref_model.__file__)
    if not origin.startswith(ref_root + os.sep):
        raise RuntimeError(f"audiocodec.model was imported from {origin}, not from the reference at {ref_root}")
    return ref_model.AudioCodec


def tiny_params():
    """A small but structurally complete configuration (the reference is fully config driven)."""
    gp = yaml.safe_load(open(os.path.join(ROOT, "config", "SimWhisperCodec.yaml")))["generator_params"]
    gp["acoustic_encoder"].update(d_model=128, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=256)
    gp["acoustic_decoder"].update(d_model=128, decoder_layers=2, decoder_attention_heads=2, decoder_ffn_dim=256)
    gp["downsample"].update(in_dim=128, hidden_dim=64)
    gp["upsample"].update(out_dim=128, hidden_dim=64)
    gp["vocos"].update(dim=64, intermediate_dim=128, num_layers=3)
    return gp


def real_params():
    return yaml.safe_load(open(os.path.join(ROOT, "config", "SimWhisperCodec.yaml")))["generator_params"]


def audio(spec):
    return [synth.synth_audio(n, index=i, kind=k) for (k, i, n) in spec]


def f32(t):
    return t.detach().to(torch.float32).cpu().numpy()


def run_case(model, name, spec, tag, stages=False):
    """encode -> decode through the reference's public surface; optionally per-stage tensors."""
    cpu = torch.device("cpu")
    wavs = audio(spec)
    t0 = time.time()
    enc = model.encode(wavs, overlap_seconds=10, device=cpu)
    dec = model.decode(enc["codes_list"], overlap_seconds=10, device=cpu)
    out = {"spec_kind": np.array([s[0] for s in spec]), "spec_index": np.array([s[1] for s in spec]),
           "spec_n": np.array([s[2] for s in spec])}
    for i, (c, w) in enumerate(zip(enc["codes_list"], dec["syn_wav_list"])):
        out[f"codes_{i}"] = c.cpu().numpy().astype(np.int32)
        w = f32(w)
        if w.size <= 60000:
            out[f"wav_{i}"] = w
        else:  # long outputs: every 7th sample + per-code-frame energy
            out[f"wav_stride7_{i}"] = w[::7].copy()
            out[f"wav_energy_{i}"] = (w.reshape(-1, 1280).astype(np.float64) ** 2).sum(1).astype(np.float32)
    if stages:
        n = torch.tensor([len(w) for w in wavs])
        L = int(n.max())
        x = torch.zeros(len(wavs), 1, L)
        for i, w in enumerate(wavs):
            x[i, 0, : len(w)] = w
        with torch.inference_mode():
            lx = [xi[:, :l].reshape(-1).numpy() for xi, l in zip(x, n)]
            feats = model.feature_extractor(lx, sampling_rate=16000, return_tensors="pt", return_attention_mask=True)
            mel = feats["input_features"]
            ml = feats["attention_mask"].sum(-1).long()
            eo, el = model.acoustic_encoder(mel, ml)
            z, zl = model.downsample(eo, el)
            zq, codes = model.quantizer(z, zl)
            cl = zl
            T = int(cl.max())
            zq2 = model.quantizer.decode(codes[:, :, :T].long(), cl)
            up, ul = model.upsample(zq2, cl)
            dm, dl = model.acoustic_decoder(up, ul)
            y, yl = model.vocos(dm, dl)
        mf = int(ml.max()) + 2
        out.update(st_mel=f32(mel[:, :, :mf]), st_mel_lens=ml.numpy(), st_mel_tail=f32(mel[:, :, -1]),
                   st_enc=f32(eo[:, :, : int(el.max())]), st_z=f32(z), st_zq=f32(zq),
                   st_codes=codes.numpy().astype(np.int32), st_code_lens=zl.numpy(), st_up=f32(up), st_dec_mel=f32(dm),
                   st_y=f32(y[:, 0]))
    np.savez_compressed(os.path.join(GOLD, f"{tag}_{name}.npz"), **out)
    print(f"  {tag}_{name}: {time.time() - t0:.1f}s codes {[c.shape for c in enc['codes_list']]}")


def run_forward_case(model, tag, T=402, lens=(402, 300)):
    B = len(lens)
    u = synth._uniform("forward/mel", B * 80 * T, 77).reshape(B, 80, T)
    mel = torch.from_numpy(u * 0.8 + 0.2)
    with torch.inference_mode():
        r = model.forward({"mel_features": mel, "mel_lens": torch.tensor(lens)})
    np.savez_compressed(os.path.join(GOLD, f"{tag}_forward.npz"), T=np.array(T), lens=np.array(lens),
                        audio=f32(r["reconstructed_audio"][:, 0]), audio_lengths=r["audio_lengths"].numpy())
    print(f"  {tag}_forward: audio {tuple(r['reconstructed_audio'].shape)} lens {r['audio_lengths'].tolist()}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default=None)
    ap.add_argument("--cases", default=None, help="'facts' to rewrite only *_facts.npz; a comma list of case names (single,ragged,zeros,short,chunked,forward) to write only those")
    ap.add_argument("--out", default=None, help="output directory (default tests/golden)")
    args = ap.parse_args()
    global GOLD
    if args.out:
        GOLD = os.path.abspath(args.out)
    os.makedirs(GOLD, exist_ok=True)
    want = None if args.cases in (None, "facts") else set(args.cases.split(","))
    AudioCodec = import_reference(args.ref)
    torch.manual_seed(0)
    for tag, gp in (("tiny", tiny_params()), ("real", real_params())):
        if args.only and args.only != tag:
            continue
        print(f"[{tag}] building reference model")
        model = AudioCodec(yaml.safe_load(yaml.safe_dump(gp))).eval()
        sd = synth.synth_state_dict(gp)
        missing = set(model.state_dict()) ^ set(sd)
        assert not missing, f"key mismatch: {sorted(missing)[:8]}"
        model.load_state_dict(sd, strict=True)
        if args.cases in (None, "facts") and want is None:
            # analytic facts about the reference that the tests restate.  The old-style weight-norm hook
            # refreshes `.weight` only inside forward, so run the module once before reading it.
            with torch.inference_mode():
                model.downsample(torch.zeros(1, gp["downsample"]["in_dim"], 8), torch.tensor([8]))
            np.savez(os.path.join(GOLD, f"{tag}_facts.npz"),
                     n_keys=np.array(len(sd)), n_params=np.array(sum(v.numel() for v in sd.values())),
                     mel_filters=np.asarray(model.feature_extractor.mel_filters, dtype=np.float64),
                     aa_filter=f32(model.downsample.res_blocks[0].block[0].upsample.filter.view(-1)),
                     wn_folded=f32(model.downsample.to_latent.weight))
        if args.cases == "facts":
            continue
        cases = [("single", [("speech", 0, 50000)], True), ("ragged", [("noise", 1, 48000), ("speech", 2, 32777)], True),
                 ("zeros", [("noise", 3, 20000), ("zero", 0, 16000)], False),
                 ("short", [("speech", 4, 1279), ("noise", 5, 1280), ("noise", 6, 2000)], False),
                 ("chunked", [("speech", 7, 352000), ("noise", 8, 48000)], False)]
        for name, spec, stages in cases:
            if want is None or name in want:
                run_case(model, name, spec, tag, stages=stages)
        if want is None or "forward" in want:
            run_forward_case(model, tag)


if __name__ == "__main__":
    main()
