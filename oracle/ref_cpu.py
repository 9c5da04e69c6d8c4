"""CPU ORACLE — test infrastructure, not product code.

A plain-PyTorch fp32 restatement, in this repository's own words, of the reference's
encode -> quantize -> vocode path (SimWhisper-Codec, `/root/reference` in the build
container).  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this module; the product path (simwhisper_codec_amd) never does.

PINNING: oracle/make_golden.py imports the reference itself in the build container,
loads the same synthetic checkpoint into both, and writes `tests/golden/*.npz`;
tests/test_oracle_cpu.py checks this file against those fixtures (and the analytic
known answers of SURVEY.md §8c).  So parity is pinned by outputs of the reference run
here; the trained checkpoint is not available offline, so parity is numerical /
architectural, not perceptual.

It is a functional restatement over a flat state_dict (no nn.Module tree): each
function cites the reference lines it follows (paths relative to the reference repo).
By default it reproduces the reference's cost model too (30 s padded log-mel, 1500
encoder tokens, materialised attention) so that timing it is a fair CPU baseline;
`trim=True` drops work that provably cannot change the result (see `encoder`).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# ---------------------------------------------------------------- constants

N_FFT, HOP, N_SAMPLES, N_FRAMES = 400, 160, 480000, 3000


def slaney_mel_filters(n_freq=201, n_mels=80, fmin=0.0, fmax=8000.0, sr=16000):
    """transformers.audio_utils.mel_filter_bank(norm="slaney", mel_scale="slaney") as called at
    audiocodec/nn/feature_extractor.py:50-58 (third party, transformers 4.53.3 pinned /
    5.15 here): slaney mel scale = linear below 1 kHz, log above; triangular filters on
    the linear-Hz FFT grid; area normalisation 2 / (f[m+2] - f[m])."""
    def hz2mel(f):
        f = np.asarray(f, dtype=np.float64)
        m = 3.0 * f / 200.0
        lg = f >= 1000.0
        m = np.where(lg, 15.0 + np.log(np.maximum(f, 1e-300) / 1000.0) * (27.0 / np.log(6.4)), m)
        return m

    def mel2hz(m):
        m = np.asarray(m, dtype=np.float64)
        f = 200.0 * m / 3.0
        lg = m >= 15.0
        return np.where(lg, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), f)

    pts = mel2hz(np.linspace(hz2mel(fmin), hz2mel(fmax), n_mels + 2))
    fft = np.linspace(0, sr // 2, n_freq)
    diff = np.diff(pts)
    slopes = pts[None, :] - fft[:, None]
    down = -slopes[:, :-2] / diff[:-1]
    up = slopes[:, 2:] / diff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    fb *= (2.0 / (pts[2:n_mels + 2] - pts[:n_mels]))[None, :]
    return fb  # (n_freq, n_mels) float64


def fold_weight_norm(g, v):
    """old-style torch.nn.utils.weight_norm (modules.py:30-31): w = g * v / ||v|| per out channel."""
    n = v.reshape(v.shape[0], -1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
    return g * v / n


class Oracle:
    def __init__(self, generator_params, state_dict, num_threads=None, dtype=torch.float32):
        """dtype=torch.float64 runs the same algorithm in double on the same float32 weights: the
        rounding-free yardstick tools/code_agreement.py measures every float32 implementation against."""
        self.gp = generator_params
        self.dtype = dtype
        if num_threads:
            torch.set_num_threads(num_threads)
        sd = {k: (v.detach().to(dtype, copy=True) if v.is_floating_point() else v.detach().clone()) for k, v in state_dict.items()}
        # fold weight norm once (the reference recomputes it every forward through the hook)
        for k in [k for k in sd if k.endswith(".weight_g")]:
            p = k[: -len("weight_g")]
            sd[p + "weight"] = fold_weight_norm(sd[k], sd[p + "weight_v"])
        self.sd = sd
        self.sr = generator_params["input_sample_rate"]
        self.rate = generator_params["encoder_downsample_rate"]
        q = generator_params["quantizer"]
        self.groups = q["num_groups"]
        self.levels = torch.tensor(q["num_levels_per_group"], dtype=torch.int32).view(1, -1, 1)
        self.base = torch.cumprod(torch.tensor([1] + q["num_levels_per_group"][:-1]), 0).to(torch.int32).view(1, -1, 1)
        self.eps = q.get("eps", 1e-3)
        self.heads_e = generator_params["acoustic_encoder"]["encoder_attention_heads"]
        self.heads_d = generator_params["acoustic_decoder"]["decoder_attention_heads"]
        self.mel_fb = torch.from_numpy(slaney_mel_filters()).to(torch.float32).to(dtype)  # (201, 80)
        self.stack = generator_params["downsample"]["stack_factor"]
        self.n_fft_v = generator_params["vocos"]["n_fft"]
        self.hop_v = generator_params["vocos"]["hop_size"]
        self.filt = sd["downsample.res_blocks.0.block.0.upsample.filter"].view(-1)

    # ------------------------------------------------------------ log-mel
    def logmel(self, wavs):
        """MelFeatureExtractor.__call__ + _torch_extract_fbank_features
        (feature_extractor.py:136-245, 86-112): zero-pad each utterance to 30 s, hann STFT
        n_fft 400 hop 160 (center, reflect), drop the last frame, power, slaney mel, log10,
        per-utterance floor at max-8, (x+4)/4.  Returns mel (B,80,3000), mel_lens (B,)
        = number of samples-mask entries at stride 160 = ceil(n/160) (:237, model.py:191)."""
        B = len(wavs)
        x = torch.zeros(B, N_SAMPLES, dtype=self.dtype)
        lens = []
        for i, w in enumerate(wavs):
            w = torch.as_tensor(w, dtype=torch.float32).reshape(-1)[:N_SAMPLES]
            x[i, : w.numel()] = w
            lens.append((w.numel() + HOP - 1) // HOP)
        st = torch.stft(x, N_FFT, HOP, window=torch.hann_window(N_FFT).to(self.dtype), return_complex=True)
        power = st[..., :-1].abs() ** 2
        mel = self.mel_fb.T @ power
        lg = torch.clamp(mel, min=1e-10).log10()
        mx = lg.amax(dim=(1, 2), keepdim=True)
        lg = torch.maximum(lg, mx - 8.0)
        return (lg + 4.0) / 4.0, torch.tensor(lens, dtype=torch.long)

    # ------------------------------------------------------- transformer
    def _layer(self, h, lens, p, heads):
        """OmniWhisperTransformerLayer.forward + VarLenAttention.forward (modules.py:214-232,
        145-187; mask :111-143): pre-LN, q scaled by d_head^-0.5 after bias, k without bias,
        additive mask = finfo.min wherever query or key is beyond the length."""
        sd = self.sd
        B, T, D = h.shape
        hd = D // heads
        x = F.layer_norm(h, (D,), sd[p + "self_attn_layer_norm.weight"], sd[p + "self_attn_layer_norm.bias"], 1e-5)
        q = F.linear(x, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"]) * hd ** -0.5
        k = F.linear(x, sd[p + "self_attn.k_proj.weight"])
        v = F.linear(x, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"])
        q, k, v = [t.view(B, T, heads, hd).transpose(1, 2) for t in (q, k, v)]
        s = q @ k.transpose(-1, -2)
        ok = torch.arange(T)[None, :] < lens[:, None]  # (B,T)
        both = (ok[:, None, :, None] & ok[:, None, None, :]).to(s.dtype)
        # the reference's additive mask is 1.0 (not 0) on valid pairs and finfo.min elsewhere (:142,174);
        # the +1 is a per-row constant, invisible to softmax except through fp32 rounding
        s = s + (both + (1.0 - both) * torch.finfo(s.dtype).min)
        a = torch.softmax(s, dim=-1) @ v
        a = a.transpose(1, 2).reshape(B, T, D)
        h = h + F.linear(a, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
        x = F.layer_norm(h, (D,), sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"], 1e-5)
        x = F.gelu(F.linear(x, sd[p + "fc1.weight"], sd[p + "fc1.bias"]))
        return h + F.linear(x, sd[p + "fc2.weight"], sd[p + "fc2.bias"])

    def encoder(self, mel, mel_lens, trim=False):
        """OmniAudioEncoder.forward, acoustic branch (modules.py:287-376, :314-319, no positional
        embedding :330): conv1 k3 p1, conv2 k3 s2 p1 (no GELU), 12 layers, LayerNorm, zero rows
        >= len, out_len = mel_len // 2.
        trim=True runs only ceil-to-even(max mel_len)+2 frames: rows beyond an utterance's
        length never reach valid rows (keys masked, everything else row-wise) and the conv
        stem's receptive field is 3 frames — checked against the untrimmed path in
        tests/test_oracle_cpu.py."""
        sd = self.sd
        T_full = mel.shape[-1]
        if trim:
            keep = min(T_full, int(mel_lens.max()) // 2 * 2 + 4)
            mel = mel[..., :keep]
        x = F.conv1d(mel, sd["acoustic_encoder.conv1.weight"], sd["acoustic_encoder.conv1.bias"], padding=1)
        x = F.conv1d(x, sd["acoustic_encoder.conv2.weight"], sd["acoustic_encoder.conv2.bias"], stride=2, padding=1)
        lens = mel_lens // 2
        h = x.transpose(1, 2)
        if trim:
            h = h[:, : max(1, int(lens.max()))]
        n_layers = self.gp["acoustic_encoder"]["encoder_layers"]
        for i in range(n_layers):
            h = self._layer(h, lens, f"acoustic_encoder.layers.{i}.", self.heads_e)
        D = h.shape[-1]
        h = F.layer_norm(h, (D,), sd["acoustic_encoder.layer_norm.weight"], sd["acoustic_encoder.layer_norm.bias"], 1e-5)
        ok = (torch.arange(h.shape[1])[None, :] < lens[:, None])[:, :, None]
        h = torch.where(ok, h, torch.zeros((), dtype=h.dtype))
        if trim:  # restore the reference's (B, D, T_full // 2) shape with exact zeros
            h = F.pad(h, (0, 0, 0, T_full // 2 - h.shape[1]))
        return h.transpose(1, 2), lens

    # ------------------------------------------- frame stack down / up
    def _act1d(self, x, p):
        """Activation1d(SnakeBeta(alpha_logscale)) (alias_free_torch/act.py:23-28, resample.py:25-33,
        46-49, filter.py:83-92, activations.py:107-120)."""
        C = x.shape[1]
        f = self.filt.view(1, 1, -1).expand(C, -1, -1)
        u = F.pad(x, (5, 5), mode="replicate")
        u = 2 * F.conv_transpose1d(u, f, stride=2, groups=C)[..., 15:-15]
        a = torch.exp(self.sd[p + "act.alpha"]).view(1, -1, 1)
        b = torch.exp(self.sd[p + "act.beta"]).view(1, -1, 1)
        u = u + (1.0 / (b + 1e-9)) * torch.sin(u * a) ** 2
        u = F.pad(u, (5, 6), mode="replicate")
        return F.conv1d(u, f, stride=2, groups=C)

    def _res_units(self, h, prefix):
        """3 x ResidualUnit, dilations 1/3/9 (modules.py:37-49)."""
        sd = self.sd
        for i, d in enumerate((1, 3, 9)):
            p = f"{prefix}.res_blocks.{i}.block."
            y = self._act1d(h, p + "0.")
            y = F.conv1d(y, sd[p + "1.weight"], sd[p + "1.bias"], dilation=d, padding=3 * d)
            y = self._act1d(y, p + "2.")
            y = F.conv1d(y, sd[p + "3.weight"], sd[p + "3.bias"])
            h = h + y
        return h

    def downsample(self, x, lens):
        """FrameStackDownConv.forward (modules.py:519-550): right-pad T to a multiple of 4, stack 4
        frames into channels with channel index d*4+s, in_proj, residual units, to_latent;
        out_len = ceil(len/4).  Nothing is masked."""
        s = self.stack
        B, D, T = x.shape
        if T % s:
            x = F.pad(x, (0, s - T % s))
        x = x.view(B, D, -1, s).permute(0, 1, 3, 2).reshape(B, D * s, -1)
        h = F.conv1d(x, self.sd["downsample.in_proj.weight"], self.sd["downsample.in_proj.bias"])
        h = self._res_units(h, "downsample")
        z = F.conv1d(h, self.sd["downsample.to_latent.weight"], self.sd["downsample.to_latent.bias"])
        return z, (lens + s - 1) // s

    def upsample(self, zq):
        """FrameStackUpConv.forward (modules.py:601-631): from_latent, residual units, to_stacked,
        unstack 'b (d s) t -> b d (t s)'.  Not masked."""
        s = self.stack
        h = F.conv1d(zq, self.sd["upsample.from_latent.weight"], self.sd["upsample.from_latent.bias"])
        h = self._res_units(h, "upsample")
        h = F.conv1d(h, self.sd["upsample.to_stacked.weight"], self.sd["upsample.to_stacked.bias"])
        B, DS, T = h.shape
        return h.view(B, DS // s, s, T).permute(0, 1, 3, 2).reshape(B, DS // s, T * s)

    # ------------------------------------------------------------- FSQ
    def _fsq_consts(self):
        """FiniteScalarQuantizer.compress constants (quantizer.py:131-137)."""
        scale = (self.levels - 1) / 2
        scale = scale * (1 - self.eps)
        offset = torch.where(self.levels % 2 == 0, 0.5, 0)
        shift = (offset / scale).tan()
        return scale, offset, shift

    @staticmethod
    def _mask_time(t, lens):
        """mask_sequence_tensor (quantizer.py:9-30) for (B, D, L) / (B, L)."""
        ok = torch.arange(t.shape[-1])[None, :] < lens[:, None]
        return t * (ok[:, None, :] if t.dim() == 3 else ok)

    def fsq_encode(self, z, lens):
        """GroupFiniteScalarQuantizer.forward (quantizer.py:273-290) over FiniteScalarQuantizer.forward
        (:181-200): tanh compression, round half-to-even, /(levels//2), mixed-radix index, masking."""
        scale, offset, shift = self._fsq_consts()
        half = self.levels // 2
        zq, idx = [], []
        for zg in z.chunk(self.groups, dim=1):
            c = torch.round(scale * torch.tanh(zg + shift) - offset)
            dq = c / half
            ix = torch.sum((half * dq + half) * self.base, dim=1).to(torch.int32)
            zq.append(self._mask_time(dq, lens))
            idx.append(self._mask_time(ix, lens).unsqueeze(0))
        return torch.cat(zq, dim=1), torch.cat(idx, dim=0)

    def fsq_decode(self, codes, lens):
        """GroupFiniteScalarQuantizer.decode (quantizer.py:306-318, 207-224)."""
        half = self.levels // 2
        out = []
        for cg in codes.chunk(self.groups, dim=0):
            nn_ = (cg.permute(1, 0, 2) // self.base) % self.levels
            out.append(self._mask_time((nn_ - half) / half, lens))
        return torch.cat(out, dim=1)

    # --------------------------------------------------------- decoder
    def decoder(self, x, lens):
        """OmniAudioDecoder.forward (modules.py:437-474): 12 masked layers (positional embedding is
        commented out :441-448), LayerNorm, zero rows >= len, ConvTranspose1d k3 s2, ConvTranspose1d
        k3 s1, crop to 2T; out_len = 2 len."""
        sd = self.sd
        h = x.transpose(1, 2)
        T = h.shape[1]
        for i in range(self.gp["acoustic_decoder"]["decoder_layers"]):
            h = self._layer(h, lens, f"acoustic_decoder.layers.{i}.", self.heads_d)
        D = h.shape[-1]
        h = F.layer_norm(h, (D,), sd["acoustic_decoder.layer_norm.weight"], sd["acoustic_decoder.layer_norm.bias"], 1e-5)
        ok = (torch.arange(T)[None, :] < lens[:, None])[:, :, None]
        h = torch.where(ok, h, torch.zeros((), dtype=h.dtype)).transpose(1, 2)
        y = F.conv_transpose1d(h, sd["acoustic_decoder.deconv1.weight"], sd["acoustic_decoder.deconv1.bias"], stride=2)
        y = F.conv_transpose1d(y, sd["acoustic_decoder.deconv2.weight"], sd["acoustic_decoder.deconv2.bias"], stride=1)
        return y[:, :, : 2 * T], lens * 2

    # ----------------------------------------------------------- Vocos
    def vocos(self, mel):
        """Vocos.forward (modules.py:1569-1573): VocosBackbone (:1492-1504) with 24 ConvNeXtBlock
        (:1229-1248), ISTFTHead (:1053-1082), ISTFT 'same' (:831-886)."""
        sd = self.sd
        p = "vocos.backbone."
        x = F.conv1d(mel, sd[p + "embed.weight"], sd[p + "embed.bias"], padding=3)
        C = x.shape[1]
        x = F.layer_norm(x.transpose(1, 2), (C,), sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6).transpose(1, 2)
        for i in range(self.gp["vocos"]["num_layers"]):
            q = f"{p}convnext.{i}."
            y = F.conv1d(x, sd[q + "dwconv.weight"], sd[q + "dwconv.bias"], padding=3, groups=C).transpose(1, 2)
            y = F.layer_norm(y, (C,), sd[q + "norm.weight"], sd[q + "norm.bias"], 1e-6)
            y = F.gelu(F.linear(y, sd[q + "pwconv1.weight"], sd[q + "pwconv1.bias"]))
            y = F.linear(y, sd[q + "pwconv2.weight"], sd[q + "pwconv2.bias"])
            x = x + (sd[q + "gamma"] * y).transpose(1, 2)
        h = F.layer_norm(x.transpose(1, 2), (C,), sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"], 1e-6)
        o = F.linear(h, sd["vocos.head.out.weight"], sd["vocos.head.out.bias"]).transpose(1, 2)
        mag, ph = o.chunk(2, dim=1)
        mag = torch.clip(torch.exp(mag), max=1e2)
        S = mag * (torch.cos(ph) + 1j * torch.sin(ph))
        n, hop = self.n_fft_v, self.hop_v
        win = sd["vocos.head.istft.window"]
        fr = torch.fft.irfft(S, n, dim=1, norm="backward") * win[None, :, None]
        T = fr.shape[-1]
        size = (T - 1) * hop + n
        pad = (n - hop) // 2
        y = F.fold(fr, output_size=(1, size), kernel_size=(1, n), stride=(1, hop))[:, 0, 0, pad:-pad]
        env = F.fold(win.square().expand(1, T, -1).transpose(1, 2), output_size=(1, size), kernel_size=(1, n),
                     stride=(1, hop)).squeeze()[pad:-pad]
        return y / env

    # ------------------------------------------------ model-level entry points
    @torch.inference_mode()
    def tokenize(self, x, lengths, trim=False):
        """AudioCodec.inference_tokenize (model.py:167-210). x (B,1,T<=480000), lengths (B,)."""
        wavs = [x[i, 0, : int(lengths[i])] for i in range(x.shape[0])]
        mel, mel_lens = self.logmel(wavs)
        h, l = self.encoder(mel, mel_lens, trim=trim)
        z, zl = self.downsample(h, l)
        zq, codes = self.fsq_encode(z, zl)
        return {"zq": zq, "codes": codes, "codes_lengths": zl, "z": z, "mel": mel, "enc": h}

    @torch.inference_mode()
    def detokenize(self, codes, lens):
        """AudioCodec.inference_detokenize (model.py:212-242)."""
        zq = self.fsq_decode(codes, lens)
        up = self.upsample(zq)
        mel, ml = self.decoder(up, lens * self.stack)
        y = self.vocos(mel)
        return {"y": y[:, None, :], "output_length": ml * self.hop_v, "zq": zq, "up": up, "mel": mel}

    @torch.inference_mode()
    def encode(self, wav_list, overlap_seconds=10, trim=False):
        """AudioCodec.encode (model.py:244-308): 30 s windows every 20 s, keep <= 250 codes per window,
        concatenate, trim each utterance to len // 1280."""
        dur = (30 - overlap_seconds) * self.sr
        chunk = 30 * self.sr
        keep = dur // self.rate
        B = len(wav_list)
        if B == 0:
            return {"codes_list": []}
        n = torch.tensor([len(w) for w in wav_list], dtype=torch.long)
        L = int(n.max())
        x = torch.zeros(B, 1, L, dtype=self.dtype)
        for i, w in enumerate(wav_list):
            x[i, 0, : len(w)] = torch.as_tensor(w, dtype=torch.float32)
        parts = []
        for c in range((L + dur - 1) // dur):
            s, e = c * dur, min(c * dur + chunk, L)
            cl = torch.clamp(n - s, 0, e - s)
            if int(cl.max()) == 0:
                continue
            r = self.tokenize(x[:, :, s:e], cl, trim=trim)
            vl = torch.clamp(r["codes_lengths"], 0, keep)
            blk = torch.zeros(self.groups, B, keep, dtype=r["codes"].dtype)
            for b in range(B):
                blk[:, b, : int(vl[b])] = r["codes"][:, b, : int(vl[b])]
            parts.append(blk)
        if not parts:
            return {"codes_list": [torch.zeros(self.groups, 0, dtype=torch.long) for _ in range(B)]}
        allc = torch.cat(parts, dim=-1)
        return {"codes_list": [allc[:, i, : int(n[i]) // self.rate] for i in range(B)]}

    @torch.inference_mode()
    def decode(self, codes_list, overlap_seconds=10):
        """AudioCodec.decode (model.py:310-373): 375-code windows every 250 codes, keep <= 320000
        samples per window, concatenate, trim to T*1280."""
        win = 30 * self.sr // self.rate
        step = (30 - overlap_seconds) * self.sr // self.rate
        keep = step * self.rate
        B = len(codes_list)
        if B == 0:
            return {"syn_wav_list": []}
        n = torch.tensor([c.shape[-1] for c in codes_list], dtype=torch.long)
        L = int(n.max())
        codes = torch.zeros(self.groups, B, L, dtype=torch.long)
        for i, c in enumerate(codes_list):
            codes[:, i, : c.shape[-1]] = c
        parts = []
        for c in range((L + step - 1) // step):
            s, e = c * step, min(c * step + win, L)
            cl = torch.clamp(n - s, 0, e - s)
            if int(cl.max()) == 0:
                continue
            r = self.detokenize(codes[:, :, s:e], cl)
            vl = torch.clamp(r["output_length"], 0, keep)
            blk = torch.zeros(B, 1, keep)
            for b in range(B):
                blk[b, :, : int(vl[b])] = r["y"][b, :, : int(vl[b])]
            parts.append(blk)
        if not parts:
            return {"syn_wav_list": [torch.zeros(0) for _ in range(B)]}
        wav = torch.cat(parts, dim=-1)
        return {"syn_wav_list": [wav[i, 0, : int(n[i]) * self.rate] for i in range(B)]}

    @torch.inference_mode()
    def forward(self, batch):
        """AudioCodec.forward (model.py:112-165): precomputed mel, no 30 s padding, no chunking,
        up-sampler sees the whole padded T."""
        mel, ml = batch["mel_features"], batch["mel_lens"]
        h, l = self.encoder(mel, ml)
        z, zl = self.downsample(h, l)
        zq, _ = self.fsq_encode(z, zl)
        up = self.upsample(zq)
        m, ol = self.decoder(up, zl * self.stack)
        y = self.vocos(m)
        return {"reconstructed_audio": y[:, None, :], "audio_lengths": ol * self.hop_v}
