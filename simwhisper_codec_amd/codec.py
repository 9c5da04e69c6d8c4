"""AudioCodec — the reference's model surface (audiocodec/model.py:15-396) on MI355X.

Same constructor argument (`generator_params` of the YAML), same checkpoint layout
(`state_dict()` keys and shapes, strict load), same methods and return dicts:
`load_from_checkpoint`, `encode`, `decode`, `inference_tokenize`,
`inference_detokenize`, `forward`, `remove_weight_norm`.  The computation is a sequence
of libswc_hip.so calls (include/swc.h) on frame-major [B, T, C] buffers; PyTorch only
owns the memory and the stream.  There is no CPU path: using the model on a non-HIP
device raises SwcError.

Work the reference does and this path provably does not need:
  * the 30 s padding of every tokenize call (feature_extractor.py:207-214): rows beyond an
    utterance's length never reach valid rows, so mel / encoder run on the valid frames
    (+2 halo frames for the conv stem) — reference measured identical codes, |diff| 3e-6;
  * the down-sampler over 375 frames: it runs on ceil(len/4)+64 frames (its receptive
    field is +-59 frames) of the zero-extended encoder output;
  * masks, permutes and per-utterance Python copy loops (model.py:293-295,358-360).
Ragged `decode` batches are padded to the batch maximum exactly like the reference,
because its un-masked up-sampler / Vocos make short utterances depend on that maximum.
"""
import functools
import logging
import math
import os
import threading

import torch
import torch.nn as nn
import yaml

from . import ops, packed, spec, trace
from ._lib import SwcError

# precision presets: (encode-side GEMM operands, decode-side GEMM operands)
#   f32  : exact-f32 MFMA (v_mfma_f32_16x16x4_f32)
#   f16s : split-f16, 3 f16 MFMAs per k-step, f32-class accuracy (SWC_F16S in include/swc.h)
#   bf16 : bf16 MFMA, f32 accumulate
#   fp8  : the `bf16` preset with the 48 encoder-transformer linears (qkv, out, fc1, fc2) on the fp8 MFMA (OCP e4m3fn,
#          f32 accumulate; BASELINE.json configs[4]); codes are no longer bit-exact, tolerance in DESIGN.md section 4
#   f16s (preset): split-f16 on BOTH sides: f32-class waveforms as well (gate 1e-4 of the peak) on the MFMA kernels that exist —
#          the strict-precision neighbour of `mixed` (bf16 decode, 1.8e-2) at a third of its decode rate instead of `fp32`'s tenth
#   fp8_fc1 : the `bf16` preset with ONLY the encoder's fc1 (+ GELU) on the fp8 MFMA: the one linear whose fp8 form keeps
#          bf16's class of FSQ-level agreement (round 3's CPU simulation: 95.4 % equal levels; q/k/v + out-proj in fp8 cost the rest)
PRECISIONS = {"fp32": ("f32", "f32"), "mixed": ("f16s", "bf16"), "mixed_f32": ("f32", "bf16"), "bf16": ("bf16", "bf16"),
              "fp8": ("bf16", "bf16"), "fp8_fc1": ("bf16", "bf16"), "f16s": ("f16s", "f16s")}
_TORCH_DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16s": torch.float16}


_TLS = threading.local()


def _on_model_device(fn):
    """Run a public entry point with the model's GPU as the current device.  Kernels are launched on the current
    device's stream (ops._stream), so a model on cuda:1 called while cuda:0 is current would otherwise enqueue on the
    wrong GPU; the reference dispatches by tensor device and has no such requirement."""
    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        dev = self._buffers_device()
        if dev.type != "cuda":  # empty inputs still return empty results (model.py:303-304); anything else raises in _packed()
            return fn(self, *a, **k)
        active = getattr(_TLS, "active", None)
        if active is None:
            active = _TLS.active = set()
        if id(self) in active:  # an entry point called from another one (encode -> inference_tokenize) on this thread:
            return fn(self, *a, **k)  # the outer guard is in force (current device, counter pointer are per thread)
        with torch.cuda.device(dev):
            # the counter pointer is per calling thread: a call into a second model from inside this one's entry point
            # (or around it) must find ITS counter again afterwards, or its later kernels would go uncounted
            prev = getattr(_TLS, "counter", None)
            buf = self._sat_state(dev)["buf"]
            ops.set_saturation_counter(buf)
            _TLS.counter = buf
            active.add(id(self))
            try:
                return fn(self, *a, **k)
            finally:
                active.discard(id(self))
                ops.set_saturation_counter(prev)
                _TLS.counter = prev
    return wrapped


class _Node(nn.Module):
    """Bare container so that buffers can sit at the reference's dotted paths."""


def _register(root, name, tensor):
    parts = name.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, _Node())
        mod = mod._modules[p]
    mod.register_buffer(parts[-1], tensor)


class _Layer:
    __slots__ = ("ln1", "wqkv", "bqkv", "wo", "bo", "ln2", "w1", "b1", "w2", "b2", "ws", "wts", "t16")  # ws / wts: operand streams of swc_mlp_block / swc_layer_tail; t16: wts holds plain-f16 weights


class _Packed:
    """Device-resident, GEMM-ready weights for one compute dtype per stage."""


class _PW:
    """One packed GEMM weight: tensor in the stage's operand format + the accumulator scale that
    undoes the power-of-two operand scales (1 except for split-f16)."""
    __slots__ = ("w", "alpha")

    def __init__(self, w, alpha=1.0):
        self.w, self.alpha = w, alpha

    @property
    def shape(self):
        return self.w.shape


_PACK_CLASSES = {"Packed": _Packed, "PW": _PW, "Layer": _Layer}


def _fold_wn(sd, p):
    g, v = sd[p + ".weight_g"], sd[p + ".weight_v"]
    nrm = v.reshape(v.shape[0], -1).norm(dim=1).view(-1, 1, 1)
    return g * v / nrm


def length_groups(units, overhead=5000.0, quad=0.11 / 500.0):
    """Partition rows SORTED BY LENGTH (descending `units`: encoder tokens / decoder tokens per row) into contiguous groups
    that are each padded to their own longest row.  A call costs about rows x T x (1 + quad x T) token-equivalents (linear
    layers + attention) plus a fixed `overhead` (launch-bound small kernels, under-filled grids); dynamic programme over
    the sorted list.  Returns [(start, end)].  A uniform batch is one group."""
    n = len(units)
    if n == 0:
        return []
    best = [0.0] * (n + 1)
    cut = [0] * (n + 1)
    for j in range(1, n + 1):
        best[j] = float("inf")
        for i in range(j):
            t = max(units[i], 1)
            c = best[i] + (j - i) * t * (1.0 + quad * t) + overhead
            if c < best[j]:
                best[j], cut[j] = c, i
    out, j = [], n
    while j > 0:
        out.append((cut[j], j))
        j = cut[j]
    return out[::-1]


def _config_digest(gp):
    import hashlib
    import json
    return hashlib.sha256(json.dumps(gp, sort_keys=True, default=str).encode()).hexdigest()[:16]


class AudioCodec(nn.Module):
    def __init__(self, generator_params, precision="mixed", _packed_file=None):
        super().__init__()
        gp = generator_params
        self.generator_params = gp
        self.input_sample_rate = gp["input_sample_rate"]
        self.output_sample_rate = gp["output_sample_rate"]
        self.max_audio_seconds = 30
        self.encoder_downsample_rate = gp["encoder_downsample_rate"]
        self.decoder_upsample_rate = gp["decoder_upsample_rate"]
        q = gp["quantizer"]
        self.num_groups = q["num_groups"]
        self.codebook_dim_per_group = len(q["num_levels_per_group"])
        self.fsq_levels = tuple(int(v) for v in q["num_levels_per_group"])
        if len(self.fsq_levels) != 4 or any(v < 2 or v > 1024 for v in self.fsq_levels) or math.prod(self.fsq_levels) >= 2 ** 31:
            raise SwcError("the FSQ kernels take groups of 4 channels with 2..1024 levels each and an int32 code per group "
                           f"(num_levels_per_group = {list(self.fsq_levels)})")
        enc = dict(gp["acoustic_encoder"])
        # keys the reference pops before building the encoder (model.py:35-39)
        self.freeze_acoustic_encoder_flag = enc.pop("freeze", False)
        self.whisper_model_path = enc.pop("whisper_model_path", None)
        self.init_from_whisper = enc.pop("init_from_whisper", False)
        if not enc.get("is_acoustic", False):
            raise SwcError("only the acoustic (no positional embedding, no GELU stem) encoder is on the hot path")
        for sect, key in (("acoustic_encoder", "encoder_attention_heads"), ("acoustic_decoder", "decoder_attention_heads")):
            if gp[sect]["d_model"] // gp[sect][key] != spec.HEAD_DIM:
                raise SwcError(f"{sect}: head_dim must be {spec.HEAD_DIM}")
        if gp["vocos"]["n_fft"] != 640 or gp["vocos"]["hop_size"] != 160 or gp["vocos"].get("padding", "same") != "same":
            raise SwcError("the ISTFT kernels are built for n_fft 640 / hop 160 / 'same' padding")
        self.precision = precision
        # a model built from a packed-operand file (tools/pack_checkpoint.py --fold) holds no state_dict tensors: its
        # operands go file -> device on first use; one anchor buffer keeps .to(device) / device tracking working
        self._packed_file = _packed_file
        if _packed_file is None:
            for name, (shape, dtype) in spec.state_shapes(gp).items():
                _register(self, name, torch.zeros(shape, dtype=dtype))
        else:
            self.register_buffer("_anchor", torch.zeros(1))
        self._pk = None
        self._pk_key = None

    # ------------------------------------------------------------ plumbing
    @property
    def precision(self):
        return self._precision

    @precision.setter
    def precision(self, p):
        if p not in PRECISIONS:
            raise ValueError(f"precision must be one of {PRECISIONS}")
        if getattr(self, "_packed_file", None) is not None and p != getattr(self, "_precision", p):
            raise SwcError(f"this model was loaded from operands packed for precision={self._precision!r} "
                           f"({self._packed_file}); pack the checkpoint again for {p!r} or load the .pt")
        self._precision = p
        self._pk = None

    def _apply(self, fn, *a, **k):
        self._pk = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        if self._packed_file is not None:
            raise SwcError("a model built from a packed-operand file has no state_dict tensors to load into")
        self._pk = None
        return super().load_state_dict(*a, **k)

    def remove_weight_norm(self):
        """Weight norm is folded once when the weights are packed; kept for API parity (model.py:101-110)."""
        return self

    @classmethod
    def load_from_checkpoint(cls, config_path: str, ckpt_path: str):
        """model.py:375-396: YAML -> model, `.pt` (bare state_dict or {'model': ...}) -> strict load.
        The file is read with weights_only=True (nothing in it is executed)."""
        logging.info(f"Loading model from {config_path} and {ckpt_path}")
        with open(config_path, "r") as f:
            config = yaml.safe_load(f)
        if str(ckpt_path).endswith(".safetensors"):
            meta = packed.peek(ckpt_path)
            if meta is not None:  # tools/pack_checkpoint.py --fold: GEMM-ready operands of one precision preset
                if meta["config"] != _config_digest(config["generator_params"]):
                    raise SwcError(f"{ckpt_path} was packed for another configuration than {config_path}")
                return cls(config["generator_params"], precision=meta["precision"], _packed_file=str(ckpt_path))
        model = cls(config["generator_params"])
        if str(ckpt_path).endswith(".safetensors"):  # tools/pack_checkpoint.py output: same keys, no pickle
            from safetensors.torch import load_file
            ckpt = load_file(ckpt_path, device="cpu")
        else:
            ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=True)
        model.load_state_dict(ckpt["model"] if "model" in ckpt else ckpt, strict=True)
        return model

    def _device(self):
        return self._buffers_device()

    def _buffers_device(self):
        return next(self.buffers()).device

    def _resolve_device(self, device):
        """`device` argument of encode / decode (model.py:245,311 default torch.device("cuda")): an un-indexed "cuda" means
        the GPU the weights live on; an explicit other GPU (or the CPU) is an error — the weights do not move per call."""
        mdev = self._buffers_device()
        if device is None:
            return mdev
        dev = torch.device(device)
        if dev.type == "cuda" and dev.index is None:
            return mdev if mdev.type == "cuda" else torch.device("cuda", torch.cuda.current_device())
        if dev != mdev:
            raise SwcError(f"device={dev} but the model is on {mdev}: move the model with .to() first")
        return dev

    # ------------------------------------------------- range guard (split-f16 / fp8 operands)
    # "fallback": an encode whose split-f16 operands clipped (|activation| >= 1023: possible with trained Whisper-style
    #             outlier channels, never seen with the synthetic checkpoint) is re-run on exact-f32 operands and the
    #             model stays on `mixed_f32` from then on — the codes equal the reference's either way;
    # "raise":    SwcError instead; "ignore": count only (saturation_count()); "off": the counters are not read back at
    #             all (no stream synchronisation inside encode; for checkpoints whose range is known).
    saturation_policy = "fallback"

    def _sat_state(self, dev):
        st = self.__dict__.get("_sat")
        if st is None or st["dev"] != dev:
            # (not inference tensors, whatever the caller's mode: saturation_count() updates the host copy in place later)
            with torch.inference_mode(False):
                st = {"dev": dev, "buf": torch.zeros(2, dtype=torch.int32, device=dev),
                      "host": torch.zeros(2, dtype=torch.int32).pin_memory(),
                      "snap": torch.zeros(2, dtype=torch.int32).pin_memory(), "snap_ev": None,
                      "seen": [0, 0], "warned": False}
            self.__dict__["_sat"] = st
        return st

    def saturation_count(self):
        """{"f16s": n, "fp8": n}: producer threads that clipped a split-f16 / fp8 activation since the model was moved
        to its device (include/swc.h swc_set_saturation_counter).  Synchronises the stream."""
        st = self.__dict__.get("_sat")
        if st is None:
            return {"f16s": 0, "fp8": 0}
        with torch.cuda.device(st["dev"]):
            st["host"].copy_(st["buf"], non_blocking=True)
            torch.cuda.current_stream().synchronize()
        n = st["host"].tolist()
        return {"f16s": int(n[0]), "fp8": int(n[1])}

    def _snapshot_counters(self):
        """enqueue a copy of the clip counters into pinned memory behind the kernels enqueued so far, with an event: the
        deferred check later waits for THAT point of the stream only, not for what was enqueued after it"""
        st = self._sat_state(self._buffers_device())
        with torch.cuda.device(st["dev"]), torch.inference_mode(False):
            st["snap"].copy_(st["buf"], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        st["snap_ev"] = ev

    def _lazy_fp8_check(self, ran_as):
        """account for the snapshot the PREVIOUS fp8-preset call left behind, if its event has completed (no waiting)"""
        st = self.__dict__.get("_sat")
        if st is not None and st["snap_ev"] is not None and st["snap_ev"].query():
            self._check_clipping(ran_as, snapshot=True)

    class _Deferred:
        """`with model.deferred_range_check() as chk:` — inside, encode-side calls do not read the clip counters back (no
        stream synchronisation between encode and the work enqueued after it); leaving the block reads them once — from a
        snapshot taken right behind the encode kernels (round 4): the host waits for the encode part of the block only and
        returns while the decode it enqueued afterwards is still running, so the next call's host work overlaps it.
        chk.clipped is then True when split-f16 operands clipped: with policy "fallback" the model has switched to exact-f32
        encoder operands and THE CALLER MUST REDO the block (its codes are not reliable); "raise" raises here."""

        def __init__(self, model):
            self.m, self.clipped = model, False

        def __enter__(self):
            self.m.__dict__["_defer"] = self.m.__dict__.get("_defer", 0) + 1
            return self

        def __exit__(self, et, ev, tb):
            self.m.__dict__["_defer"] -= 1
            if et is None and self.m.__dict__["_defer"] == 0 and self.m.__dict__.pop("_defer_pending", False):
                self.clipped = self.m._check_clipping(self.m.__dict__.pop("_defer_ran_as", None), snapshot=True)
            return False

    def deferred_range_check(self):
        return AudioCodec._Deferred(self)

    def _check_clipping(self, ran_as=None, snapshot=False):
        """Read the counters (synchronises; snapshot: waits for the event of the last _snapshot_counters() only), apply the policy.  True: split-f16 operands clipped since the last check and the
        model switched to exact-f32 encoder operands (policy "fallback"): the caller re-runs.  ran_as: the preset the checked
        kernels ran with (with batches in flight another thread may have switched this model's preset meanwhile)."""
        e = PRECISIONS[ran_as or self._precision][0]
        st = self._sat_state(self._buffers_device())
        if snapshot and st["snap_ev"] is not None:
            st["snap_ev"].synchronize()
            v = st["snap"].tolist()
            n, st["snap_ev"] = {"f16s": int(v[0]), "fp8": int(v[1])}, None
        else:
            n = self.saturation_count()
        new16, new8 = n["f16s"] - st["seen"][0], n["fp8"] - st["seen"][1]
        st["seen"] = [n["f16s"], n["fp8"]]
        if new8 and not st["warned"]:
            st["warned"] = True
            logging.warning("fp8 preset: %d producer threads clipped activations at |x| = 28 in this call "
                            "(part of this preset's stated tolerance)", new8)
        if new16 and e == "f16s":
            if self.saturation_policy == "raise":
                raise SwcError(f"split-f16 operands clipped ({new16} producer threads saw |activation| >= 1023): the codes of "
                               "this call are not reliable; use precision='mixed_f32'")
            if self.saturation_policy == "fallback":
                logging.warning("split-f16 operands clipped (%d producer threads saw |activation| >= 1023): re-running this "
                                "call on exact-f32 encoder operands; the model stays on precision='mixed_f32'", new16)
                self._switch_precision("mixed_f32" if (ran_as or self._precision) in ("mixed", "mixed_f32") else "fp32")
                return True
        return False

    _REPACK_LOCK = threading.RLock()

    def _switch_precision(self, p):
        """the range-guard fallback.  A replica has no weights of its own to pack from: its origin packs the new preset
        (once, under a lock: replicas run on other threads) and the replica adopts those operands."""
        origin = self.__dict__.get("_origin")
        if origin is None:
            with AudioCodec._REPACK_LOCK:
                if self._precision != p:
                    if self._packed_file is not None:
                        # a packed-operand file holds no f32 weights to pack from: it carries the exact-f32 encoder
                        # operands of its fallback preset beside the split-f16 ones (export_packed); adopt those
                        self._precision = p
                        self._pk = None
                    else:
                        self.precision = p
                self._packed()
            return
        with AudioCodec._REPACK_LOCK:
            if origin._precision != p:
                origin._switch_precision(p)
            P = origin._packed()
        self._precision, self._pk, self._pk_key = p, P, origin._pk_key

    def _follow_origin(self):
        """a replica whose origin has switched presets meanwhile (another thread's batch clipped) follows it"""
        origin = self.__dict__.get("_origin")
        if origin is not None and origin._precision != self._precision:
            with AudioCodec._REPACK_LOCK:
                P = origin._packed()
                self._precision, self._pk, self._pk_key = origin._precision, P, origin._pk_key

    def _guarded_encode(self, run):
        """run(P) enqueues the encode-side kernels of one call and returns its outputs.  After it, the clip counters are
        read back (one 8-byte copy + stream sync per call; presets without reduced-range encode operands skip it)."""
        self._follow_origin()
        with AudioCodec._REPACK_LOCK:  # (preset, operands) as one consistent pair: another thread may be switching presets
            ran_as = self._precision
            P = self._packed()  # raises for a model that is not on the HIP device
        e = PRECISIONS[ran_as][0]
        if (e != "f16s" and ran_as not in ("fp8", "fp8_fc1")) or self.saturation_policy == "off":
            return run(P)
        if e != "f16s":
            # fp8 presets: clipping is part of the preset's stated tolerance — counted and warned about, never re-run — so
            # nothing has to wait for the counters: they are snapshotted behind this call's kernels and looked at when the next
            # call comes in (round 4: the blocking read-back cost 0.4 ms of a 14.9 ms step, tools/probes/cmp_presets.py)
            self._lazy_fp8_check(ran_as)
            out = run(P)
            self._snapshot_counters()
            return out
        out = run(P)
        if self.__dict__.get("_defer", 0) > 0:  # inside deferred_range_check(): one read-back when the block ends
            self.__dict__["_defer_pending"] = True
            self.__dict__["_defer_ran_as"] = ran_as
            self._snapshot_counters()   # (counters are cumulative: the last snapshot of the block covers every encode in it)
            return out
        if self._check_clipping(ran_as):
            with AudioCodec._REPACK_LOCK:
                P = self._packed()
            out = run(P)
        return out

    # ------------------------------------------------------------- packing
    def _dtypes(self):
        e, d = PRECISIONS[self._precision]
        return _TORCH_DT[e], _TORCH_DT[d]

    # encode-side operand fields of _Packed that depend on the encode operand type: what a packed-operand file stores a
    # second time, in exact f32, for the range-guard fallback (mixed -> mixed_f32)
    _ENCODE_FIELDS = ("edt", "c1dt", "c1k", "c1w", "c1b", "c2w", "c2b", "enc_ldt", "enc_layers", "enc_ln", "inw", "inb",
                      "down_units", "tlw", "tlb")
    _FALLBACK_OF = {"mixed": "mixed_f32"}

    def _packed(self):
        # under the lock: the range-guard fallback of another thread (a batch in flight on a replica) replaces the
        # (preset, operands) pair of this object while inference_detokenize / forward of this thread fetch it
        with AudioCodec._REPACK_LOCK:
            dev = self._buffers_device()
            if dev.type != "cuda":
                raise SwcError("AudioCodec runs on the HIP device only (call .to('cuda')); there is no CPU fallback")
            key = (dev, self._precision)
            if self._pk is not None and self._pk_key == key:
                return self._pk
            if self._packed_file is not None:
                self._pk = self._load_packed_file(dev)
            else:
                self._pk = self._pack(dev)
            self._pk_key = key
            return self._pk

    def _load_packed_file(self, dev):
        """operands of a packed-operand file for the CURRENT preset: the preset it was packed for, or that preset's
        range-guard fallback (exact-f32 encoder operands stored in the same file, loaded only when needed)."""
        P, meta = packed.load(self._packed_file, dev, _PACK_CLASSES)
        if meta["abi"] != ops.abi_version():
            raise SwcError(f"{self._packed_file}: packed for library ABI {meta['abi']}, this is {ops.abi_version()}: "
                           "pack the checkpoint again")
        if not hasattr(P, "c1k"):   # a file written before conv1 had a padded form: n_mel bins per tap, as packed
            P.c1k = P.n_mel
        if meta["precision"] == self._precision:
            return P
        if self._FALLBACK_OF.get(meta["precision"]) == self._precision:
            fb = packed.load_extra(self._packed_file, dev, _PACK_CLASSES, "fallback_encode")
            if fb is None:
                raise SwcError(f"split-f16 operands clipped, and {self._packed_file} carries no exact-f32 fallback operands "
                               f"(written by an older tools/pack_checkpoint.py): pack the checkpoint again, pack it for "
                               f"precision={self._precision!r}, or load the .pt")
            for k in self._ENCODE_FIELDS:
                setattr(P, k, fb[k] if k != "c1k" else fb.get(k, P.n_mel))
            return P
        raise SwcError(f"{self._packed_file}: packed for precision {meta['precision']}, this is {self._precision}: "
                       "pack the checkpoint again")

    _TUNABLES = ("saturation_policy", "varlen_packing", "length_bucketing", "bucket_overhead_tokens", "trim_vocos", "ragged_vocos",
                 "vocos_streams", "vocos_phase_us", "vocos_split_override", "max_rows_per_call", "fused_mlp_min_rows",
                 "fused_layer_mlp_min_rows", "layer_fusion")

    def replica(self):
        """A second AudioCodec over the SAME device-resident operands (nothing is copied or re-packed), with its own
        staging ring, range counters, side stream and caches: what a second host thread needs to run batches on its own
        stream beside this one (pipeline.InFlight).  A replica cannot pack by itself: when its split-f16 operands clip, the
        origin packs the exact-f32 preset (range-guard fallback) and every replica follows."""
        P = self._packed()
        r = AudioCodec(self.generator_params, precision=self._precision, _packed_file="<replica of a loaded model>")
        r = r.to(self._buffers_device()).eval()
        r._pk, r._pk_key = P, self._pk_key
        r.__dict__["_origin"] = self.__dict__.get("_origin") or self
        for name in self._TUNABLES:
            if name in self.__dict__:
                setattr(r, name, self.__dict__[name])
        return r

    def export_packed(self, path):
        """Write this model's GEMM-ready operands (current precision preset, current device) to a packed-operand
        .safetensors file that load_from_checkpoint() maps straight to the device (tools/pack_checkpoint.py --fold)."""
        P = self._packed()
        dev = self._buffers_device()
        extra = {}
        fb = self._FALLBACK_OF.get(self._precision)
        if fb is not None and self._packed_file is None:
            # the exact-f32 encoder operands the range guard falls back to when split-f16 activations clip (trained
            # Whisper-style outlier channels): stored beside the preset's own, read only if that ever happens.
            # Under the repack lock: a replica / thread fetching (preset, operands) meanwhile must never see the
            # temporary (fallback preset, no operands) pair
            with AudioCodec._REPACK_LOCK:
                keep = (self._precision, self._pk, self._pk_key)
                try:
                    self._precision, self._pk = fb, None
                    Pf = self._pack(dev)
                finally:
                    self._precision, self._pk, self._pk_key = keep
            extra["fallback_encode"] = {k: getattr(Pf, k) for k in self._ENCODE_FIELDS}
        elif fb is not None and os.path.exists(str(self._packed_file)):
            # a model that was itself loaded from a packed file: its fallback part travels on (dropping it silently would
            # leave the re-exported file unable to recover from clipping)
            part = packed.load_extra(self._packed_file, dev, _PACK_CLASSES, "fallback_encode")
            if part is None:
                raise SwcError(f"export_packed: {self._packed_file} carries no exact-f32 fallback operands to copy; export "
                               "from the .pt checkpoint instead")
            extra["fallback_encode"] = part
        elif fb is not None:
            raise SwcError("export_packed: a replica holds no checkpoint to pack the exact-f32 fallback operands from; export "
                           "from its origin")
        torch.cuda.synchronize(dev)
        meta = {"precision": self._precision, "config": _config_digest(self.generator_params), "abi": ops.abi_version(),
                "fallback": fb if extra else None}
        return packed.save(path, P, _PACK_CLASSES, meta, extra=extra)

    @torch.no_grad()
    def _pack(self, dev):
        gp = self.generator_params
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        edt, ddt = self._dtypes()
        P = _Packed()
        P.edt, P.ddt = edt, ddt

        def W(t, dt):
            """[N, taps*K] f32 -> packed operand.  split-f16 weights are scaled by the power of two that puts
            max|w| in [8192, 16384): lo halves stay normal fp16 numbers, nothing overflows."""
            t = t.to(dev, torch.float32).contiguous()
            if dt == torch.bfloat16:
                return _PW(ops.cast_bf16(t))
            if dt == ops.FP8_T:  # per-tensor power-of-two scale: max|w| lands in [224, 448)
                mx = float(t.abs().max())
                sw = 2.0 ** math.floor(math.log2(448.0 / mx)) if mx > 0 else 1.0
                return _PW(ops.cast_fp8(t, sw), 1.0 / (ops.FP8_ACT_SCALE * sw))
            if dt == torch.float16:
                mx = float(t.abs().max())
                sw = 2.0 ** math.floor(math.log2(16384.0 / mx)) if mx > 0 else 1.0
                return _PW(ops.cast_f16s(t.view(t.shape[0], -1), t.shape[-1], scale=sw), 1.0 / (ops.F16S_ACT_SCALE * sw))
            return _PW(t)

        def V(t):
            return t.to(dev, torch.float32).contiguous()

        def conv_w(w):  # (Cout, Cin, k) -> [Cout][k][Cin]
            return w.permute(0, 2, 1).reshape(w.shape[0], -1)

        def layers(prefix, n, dt, fc1_dt=None):
            """fc1_dt: operand type of fc1 alone when it differs from the other linears' (preset fp8_fc1)"""
            out = []
            for i in range(n):
                p = f"{prefix}.layers.{i}."
                L = _Layer()
                s = spec.HEAD_DIM ** -0.5  # q = (Wq x + bq) * s (modules.py:159): folded, exact (power of two)
                wq, wk, wv = sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.v_proj.weight"]
                L.wqkv = W(torch.cat([wq * s, wk, wv], 0), dt)
                L.bqkv = V(torch.cat([sd[p + "self_attn.q_proj.bias"] * s, torch.zeros_like(sd[p + "self_attn.q_proj.bias"]),
                                      sd[p + "self_attn.v_proj.bias"]]))
                L.wo, L.bo = W(sd[p + "self_attn.out_proj.weight"], dt), V(sd[p + "self_attn.out_proj.bias"])
                L.ln1 = (V(sd[p + "self_attn_layer_norm.weight"]), V(sd[p + "self_attn_layer_norm.bias"]))
                L.ln2 = (V(sd[p + "final_layer_norm.weight"]), V(sd[p + "final_layer_norm.bias"]))
                L.w1, L.b1 = W(sd[p + "fc1.weight"], fc1_dt or dt), V(sd[p + "fc1.bias"])
                L.w2, L.b2 = W(sd[p + "fc2.weight"], dt), V(sd[p + "fc2.bias"])
                # the fused MLP sub-block kernel (swc_mlp_block) exists for the shipped geometry with bf16 operands
                ok = dt == torch.bfloat16 and ops.mlp_supported(L.w1.w.shape[1], L.w1.w.shape[0])
                f8fc1 = fc1_dt == ops.FP8_T   # (preset fp8_fc1: swc_layer_tail runs fc1 on the fp8 MFMA; swc_mlp_block is bf16 only)
                # (one stream per layer and form, ~10 MB each: only the form `layer_fusion` selects at pack time is built)
                L.ws = ops.mlp_pack(L.w1.w, L.w2.w) if ok and not f8fc1 and self.layer_fusion == 1 else None
                L.wts, L.t16 = None, False
                if ok and self.layer_fusion >= 2:
                    f32w = [sd[p + k] for k in ("self_attn.out_proj.weight", "fc1.weight", "fc2.weight")]
                    # plain f16 inside the kernel (11 significand bits for weights, LayerNorm output, GELU output) when the weights
                    # are representable; the kernel saturates its conversions at +-65504
                    L.t16 = bool(self.layer_tail_f16 and not f8fc1 and
                                 all(bool(torch.isfinite(t).all()) and float(t.abs().max()) < 8192.0 for t in f32w))
                    if L.t16:
                        L.wts = ops.layer_tail_pack(*[V(t).to(torch.float16) for t in f32w])
                    else:
                        L.wts = ops.layer_tail_pack(L.wo.w, L.w1.w, L.w2.w)
                out.append(L)
            return out

        def res_units(prefix, dt):
            units = []
            for i, d in enumerate((1, 3, 9)):
                b = f"{prefix}.res_blocks.{i}.block."
                u = {"dil": d}
                for a in ("0", "2"):
                    fu, fd = sd[b + a + ".upsample.filter"].reshape(-1), sd[b + a + ".downsample.lowpass.filter"].reshape(-1)
                    if not torch.equal(fu.cpu(), fd.cpu()):
                        raise SwcError("anti-alias up/down filters differ; the fused kernel assumes one 12-tap filter")
                    u["f" + a] = [float(x) for x in fu.cpu()]
                    u["a" + a] = V(torch.exp(sd[b + a + ".act.alpha"].cpu().float()))
                    u["b" + a] = V(torch.exp(sd[b + a + ".act.beta"].cpu().float()))
                u["w1"], u["c1"] = W(conv_w(_fold_wn(sd, b + "1")), dt), V(sd[b + "1.bias"])
                u["w3"], u["c3"] = W(conv_w(_fold_wn(sd, b + "3")), dt), V(sd[b + "3.bias"])
                units.append(u)
            return units

        # ---- encode side
        e = gp["acoustic_encoder"]
        P.D, P.He, P.n_mel = e["d_model"], e["encoder_attention_heads"], e["num_mel_bins"]
        P.dft = _PW(V(spec.dft_basis_400()))                               # [402][400] f32 always
        fb = torch.from_numpy(spec.slaney_mel_filters(n_mels=P.n_mel)).float().T.contiguous()  # [80][201]
        P.melw = _PW(V(torch.nn.functional.pad(fb, (0, 208 - 201))))      # [80][208]
        # conv1 contracts K = 80 mel bins per tap: not a multiple of the 32-element split-f16 block.  The log-mel stays f32
        # (P.c1dt: what _logmel returns and forward() is given); for the split-f16 preset _encoder_impl zero-pads the bins to
        # 96 and converts, and the weights are packed [D][3][96]: conv1 then runs on the split-f16 MFMA path like every other
        # encoder GEMM (151 -> ~55 us per step at 32 x 10 s; it was the last exact-f32 GEMM of the `mixed` encode side)
        c1dt = torch.float32 if edt == torch.float16 else edt
        P.c1dt = c1dt
        P.c1k = P.n_mel
        w1c = sd["acoustic_encoder.conv1.weight"]                       # (D, n_mel, 3)
        if edt == torch.float16 and self.conv1_split_f16:
            P.c1k = spec.cdiv(P.n_mel, 32) * 32
            w1c = torch.nn.functional.pad(w1c, (0, 0, 0, P.c1k - P.n_mel))
            P.c1w = W(conv_w(w1c), torch.float16)
        else:
            P.c1w = W(conv_w(w1c), c1dt)
        P.c1b = V(sd["acoustic_encoder.conv1.bias"])
        P.c2w, P.c2b = W(conv_w(sd["acoustic_encoder.conv2.weight"]), edt), V(sd["acoustic_encoder.conv2.bias"])
        P.enc_ldt = ops.FP8_T if self._precision == "fp8" else edt  # operand type of the encoder-transformer linears
        P.enc_layers = layers("acoustic_encoder", e["encoder_layers"], P.enc_ldt,
                              fc1_dt=ops.FP8_T if self._precision == "fp8_fc1" else None)
        P.enc_ln = (V(sd["acoustic_encoder.layer_norm.weight"]), V(sd["acoustic_encoder.layer_norm.bias"]))
        ds = gp["downsample"]
        P.stack, P.hid, P.lat = ds["stack_factor"], ds["hidden_dim"], ds["latent_dim"]
        w = _fold_wn(sd, "downsample.in_proj")[:, :, 0]                     # [hid][d*s + s_idx]
        w = w.view(P.hid, ds["in_dim"], P.stack).permute(0, 2, 1).reshape(P.hid, -1)  # -> [hid][s_idx*D + d]
        P.inw, P.inb = W(w, edt), V(sd["downsample.in_proj.bias"])
        P.down_units = res_units("downsample", edt)
        P.tlw, P.tlb = W(_fold_wn(sd, "downsample.to_latent")[:, :, 0], edt), V(sd["downsample.to_latent.bias"])
        q = gp["quantizer"]
        P.fsq = spec.fsq_constants(q["num_levels_per_group"], q.get("eps", 1e-3))

        # ---- decode side
        us = gp["upsample"]
        P.uhid = us["hidden_dim"]
        # 16-bit decode presets keep their 16-bit operands where the flops are (12 decoder layers, 24 ConvNeXt blocks: 97 % of the
        # decode side) and run the small stages around them on split-f16 operands: the up-sampler (P.udt), the decoder's output
        # stage + Vocos' embed conv (P.io16) and the ISTFT head (P.tail16).  tools/probes/decode_error_attribution.py
        P.udt = torch.float16 if (ddt == torch.bfloat16 and self.upsample_split_f16) else ddt
        P.flw, P.flb = W(_fold_wn(sd, "upsample.from_latent")[:, :, 0], P.udt), V(sd["upsample.from_latent.bias"])
        P.up_units = res_units("upsample", P.udt)
        w = _fold_wn(sd, "upsample.to_stacked")[:, :, 0]                    # [(d*s + s_idx)][hid]
        od = us["out_dim"]
        w = w.view(od, P.stack, -1).permute(1, 0, 2).reshape(od * P.stack, -1)
        b = sd["upsample.to_stacked.bias"].view(od, P.stack).T.reshape(-1)
        P.tsw, P.tsb = W(w, P.udt), V(b)
        dc = gp["acoustic_decoder"]
        P.Dd, P.Hd = dc["d_model"], dc["decoder_attention_heads"]
        P.dec_layers = layers("acoustic_decoder", dc["decoder_layers"], ddt)
        P.dec_ln = (V(sd["acoustic_decoder.layer_norm.weight"]), V(sd["acoustic_decoder.layer_norm.bias"]))
        P.io16 = ddt == torch.float16 or (ddt == torch.bfloat16 and self.decoder_io_split_f16 and dc["d_model"] % 32 == 0)
        iodt = torch.float16 if P.io16 else ddt
        w1 = sd["acoustic_decoder.deconv1.weight"]                          # (ci, co, j)
        P.d1w, P.d1b = W(w1.permute(2, 1, 0).reshape(-1, w1.shape[0]), iodt), V(sd["acoustic_decoder.deconv1.bias"])
        w2 = sd["acoustic_decoder.deconv2.weight"]                          # (ci, co, j): conv with flipped taps, pad 2
        P.d2w = W(w2.flip(2).permute(1, 2, 0).reshape(w2.shape[1], -1), iodt)
        P.d2b = V(sd["acoustic_decoder.deconv2.bias"])
        v = gp["vocos"]
        P.vdim, P.vint, P.vin = v["dim"], v["intermediate_dim"], v["input_channels"]
        p = "vocos.backbone."
        # split-f16 decode (preset f16s): K must be a multiple of the 32-element split block: the decoder's mel (80 channels) is
        # zero-padded to 96 per tap for the embed conv and the ISTFT spectrum (642 -> 648 columns) to 672, as conv1 does on the
        # encode side
        dd16 = ddt == torch.float16
        P.vin_k = spec.cdiv(P.vin, 32) * 32 if P.io16 else P.vin
        # the ISTFT head (final LayerNorm -> Linear 512 -> 642 -> exp / cos / sin -> inverse DFT) in split-f16 while the backbone
        # keeps 16-bit operands: 0.7 % of Vocos' flops, and the stage whose rounding the waveform sees undamped (log-magnitudes
        # go through exp(), phases through cos / sin)
        P.tail16 = dd16 or (ddt == torch.bfloat16 and self.vocos_head_split_f16 and v["dim"] % 32 == 0)
        P.idft_k = 672 if P.tail16 else 648
        emw = sd[p + "embed.weight"]                                        # (C, vin, 7)
        if P.vin_k != P.vin:
            emw = torch.nn.functional.pad(emw, (0, 0, 0, P.vin_k - P.vin))
        P.emw, P.emb = W(conv_w(emw), iodt), V(sd[p + "embed.bias"])
        P.vnorm = (V(sd[p + "norm.weight"]), V(sd[p + "norm.bias"]))
        P.blocks = []
        # the fused MLP kernel (swc_convnext_mlp) exists for the shipped geometry with bf16 operands
        P.fused_mlp = ddt == torch.bfloat16 and ops.convnext_supported(P.vdim, P.vint)
        # the fused block's internal operands in plain f16 when every pointwise weight is representable (finite, and the block's
        # LayerNorm cannot leave the range: sqrt(C) max|ln_w| + max|ln_b| < 65504 / 8)
        P.cx_f16 = bool(P.fused_mlp and self.convnext_f16 and self._cx_f16_safe(sd, p, v["num_layers"], P.vdim))
        for i in range(v["num_layers"]):
            b_ = f"{p}convnext.{i}."
            P.blocks.append(dict(
                dw=V(sd[b_ + "dwconv.weight"][:, 0].T), db=V(sd[b_ + "dwconv.bias"]),
                ln=(V(sd[b_ + "norm.weight"]), V(sd[b_ + "norm.bias"])),
                w1=W(sd[b_ + "pwconv1.weight"], ddt), b1=V(sd[b_ + "pwconv1.bias"]),
                w2=W(sd[b_ + "pwconv2.weight"], ddt), b2=V(sd[b_ + "pwconv2.bias"]), g=V(sd[b_ + "gamma"])))
            if P.fused_mlp:
                blk = P.blocks[-1]
                if P.cx_f16:   # plain half precision inside the fused block kernel: the same f32 weights rounded to 11 bits, not 8
                    blk["ws"] = ops.convnext_pack(V(sd[b_ + "pwconv1.weight"]).to(torch.float16),
                                                  V(sd[b_ + "pwconv2.weight"]).to(torch.float16), blk["g"])
                else:
                    blk["ws"] = ops.convnext_pack(blk["w1"].w, blk["w2"].w, blk["g"])
        # frames a kept sample can depend on: embed k7 (+-3), one depthwise k7 per block (+-3 each), ISTFT overlap (+-3)
        if self.VOCOS_HALO_FRAMES < 3 * (v["num_layers"] + 1) + 3:
            raise SwcError(f"VOCOS_HALO_FRAMES = {self.VOCOS_HALO_FRAMES} is too small for {v['num_layers']} ConvNeXt blocks "
                           "(halo-trimmed / tile-skipping Vocos would no longer be bit-exact)")
        P.vfin = (V(sd[p + "final_layer_norm.weight"]), V(sd[p + "final_layer_norm.bias"]))
        tdt = torch.float16 if P.tail16 else ddt
        P.hw, P.hb = W(sd["vocos.head.out.weight"], tdt), V(sd["vocos.head.out.bias"])
        win = sd["vocos.head.istft.window"].float()
        P.idft = W(spec.idft_basis(640, win, P.idft_k), tdt)                 # [640][648] ([640][672] for split-f16 operands)
        P.wsq = V(win.square())
        torch.cuda.synchronize(dev)
        return P

    # --------------------------------------------------------- sub-graphs
    def _cast(self, x, dt, K=None):
        """f32 -> the stage's GEMM operand format (no-op for f32)."""
        if dt == torch.bfloat16 and x.dtype != dt:
            return ops.cast_bf16(x)
        if dt == torch.float16 and x.dtype != dt:
            return ops.cast_f16s(x, x.shape[-1] if K is None else K)
        return x

    @staticmethod
    def _mm(A, pw, M, N, K, out_dtype=None, **kw):
        """swc_gemm against a packed weight; split-f16 outputs are written at the activation scale."""
        if out_dtype == torch.float16:
            kw["out_scale"] = ops.F16S_ACT_SCALE
        elif out_dtype == ops.FP8_T:
            kw["out_scale"] = ops.FP8_ACT_SCALE
        return ops.gemm(A, pw.w, M, N, K, alpha=pw.alpha, out_dtype=out_dtype, **kw)

    def _transformer(self, h, lens, B, T, layers, H, dt, row_start=None):
        """12 x OmniWhisperTransformerLayer (modules.py:214-232). h: f32 residual stream (updated in place), padded
        [B*T, D] or — row_start given — packed [sum(len), D] (every row-wise kernel just sees fewer rows; attention finds
        utterance b at row_start[b]).
        bf16 operands at the shipped geometry: the MLP sub-block is ONE kernel (swc_mlp_block) that also applies the
        LayerNorm in front of it and emits the normalised operand of the NEXT layer's q/k/v projection, so a layer is
        qkv GEMM -> attention -> out-proj GEMM -> swc_mlp_block: no LayerNorm launch except the first of the stack."""
        D = h.shape[-1]
        M = h.shape[0]
        fp8 = dt == ops.FP8_T
        adt = torch.bfloat16 if fp8 else dt  # fp8 linears feed a bf16 attention
        lnB, lnT = (B, T) if row_start is None else (1, M)
        fused = self._mlp_fused(layers, M, dt)
        x = None  # the normalised operand of this layer's q/k/v projection when the previous layer's MLP kernel produced it
        for i, L in enumerate(layers):
            if x is None:
                x = ops.layernorm(h, L.ln1[0], L.ln1[1], 1e-5, B=lnB, t_in=lnT, C_=D, out_dtype=dt)
            qkv = self._mm(x, L.wqkv, M, 3 * D, D, lda=D, bias=L.bqkv, out_dtype=adt)
            a = ops.attention(qkv, lens, B, T, H, row_start=row_start, rows=M)
            if fp8:
                a = ops.cast_fp8(a)
            F_ = L.b1.shape[0]
            nxt = layers[i + 1].ln1 if i + 1 < len(layers) else None
            if fused and self.layer_fusion >= 2 and getattr(L, "wts", None) is not None:
                # out-proj + residual + LayerNorm + MLP + residual + next LayerNorm: one kernel
                _, x = ops.layer_tail(a, h, L.wts, L.bo, L.ln2[0], L.ln2[1], 1e-5, L.b1, L.b2, M=M, D=D, F=F_, next_ln=nxt,
                                      fc1_dtype=L.w1.w.dtype, fc1_alpha=L.w1.alpha,
                                      operands=torch.float16 if getattr(L, "t16", False) else torch.bfloat16)
                continue
            self._mm(a, L.wo, M, D, D, lda=D, bias=L.bo, residual=h, out=h)
            if fused and self.layer_fusion == 1:
                _, x = ops.mlp_block(h, L.ln2[0], L.ln2[1], 1e-5, L.ws, L.b1, L.b2, M=M, D=D, F=F_, next_ln=nxt)
                continue
            dt1 = L.w1.w.dtype if L.w1.w.dtype == ops.FP8_T else dt  # (preset fp8_fc1: fc1 alone reads e4m3 rows, writes bf16)
            x = ops.layernorm(h, L.ln2[0], L.ln2[1], 1e-5, B=lnB, t_in=lnT, C_=D, out_dtype=dt1)
            f = self._mm(x, L.w1, M, F_, D, lda=D, bias=L.b1, act=ops.ACT_GELU, out_dtype=dt)
            self._mm(f, L.w2, M, D, F_, lda=F_, bias=L.b2, residual=h, out=h)
            x = None
        return h

    # tokens from which the transformer's MLP sub-block runs as swc_mlp_block (64-token tiles, one per CU: below ~160 busy
    # CUs the two-GEMM form, whose 64/128-row tiles of N = 3072 spread over more CUs, is faster)
    fused_layer_mlp_min_rows = 64 * 160
    layer_fusion = 2  # 2: swc_layer_tail (out-proj + MLP sub-block in one kernel); 1: swc_mlp_block behind an out-proj GEMM; 0: off
    #                   (read at pack time for the operand streams: set it before the first call, or drop `_pk` to re-pack)

    def _mlp_fused(self, layers, M, dt):
        if dt != torch.bfloat16 or M < self.fused_layer_mlp_min_rows or self.layer_fusion <= 0:
            return False
        if any(getattr(L, "wts" if self.layer_fusion >= 2 else "ws", None) is None for L in layers):
            return False  # (operands packed for another `layer_fusion`, an older packed file, another geometry: the unfused layers)
        if self.fused_layer_mlp_min_rows <= 0:  # forced (tests run the kernel on the few-second fixtures)
            return True
        tiles = spec.cdiv(M, 64)
        rounds = spec.cdiv(tiles, self.CUS)
        return tiles >= 0.6 * rounds * self.CUS  # a poorly filled last round of 64-token tiles costs a whole round

    # True: the fused ConvNeXt block keeps its internal operands (LayerNorm output, GELU output, pointwise weights) in plain f16
    # instead of bf16 (read at pack time).  Measured (tools/probes/decode_error_budget.py, profiles/r04_decode_error_budget.txt):
    # the Vocos stage error does not move (2.90e-3 against 2.92e-3 once the ISTFT head runs on split-f16 operands: the blocks'
    # operand rounding is not what the waveform sees) and the step is 1 - 2 % slower (f16 MFMAs draw more than bf16 ones): off
    convnext_f16 = False

    @staticmethod
    def _cx_f16_safe(sd, p, n_blocks, C):
        lim = 65504.0 / 8
        for i in range(n_blocks):
            b_ = f"{p}convnext.{i}."
            w1, w2 = sd[b_ + "pwconv1.weight"], sd[b_ + "pwconv2.weight"]
            if not (bool(torch.isfinite(w1).all()) and bool(torch.isfinite(w2).all())):
                return False
            if float(w1.abs().max()) >= lim or float(w2.abs().max()) >= lim:
                return False
            if math.sqrt(C) * float(sd[b_ + "norm.weight"].abs().max()) + float(sd[b_ + "norm.bias"].abs().max()) >= lim:
                return False
        return True

    # 16-bit decode presets (read at pack time): small stages around the decoder layers and the ConvNeXt blocks on split-f16
    # operands — f32-class arithmetic where the waveform sees rounding undamped (measured at 32 x 10 s,
    # waveform error against the reference / decode time: profiles/r04_decode_error_budget.txt):
    #   none 1.5 - 1.8e-2 | head 8.5e-3, +0.05 ms | head + io 6.5e-3, +0.30 ms | head + io + up 5.8e-3, +0.55 ms (the 12 bf16 decoder
    #   layers alone: 5.3 - 6.0e-3) | with layer_tail_f16 (+0.07 ms): 7.0e-3 / 5.2e-3 / 3.6e-3.
    # The head and the f16 layer tail are on: 2.5 x less error for 0.6 % of the step; the other two are there to be switched on
    layer_tail_f16 = True          # swc_layer_tail's internal operands (attention tile, LayerNorm output, GELU output, weights) in plain f16
    #                                instead of bf16 (8 x smaller operand rounding at the same MFMA rate; conversions saturate at +-65504)
    vocos_head_split_f16 = True    # final LayerNorm -> Linear 512 -> 642 -> exp / cos / sin -> inverse DFT
    decoder_io_split_f16 = False   # decoder: final LayerNorm -> deconv1 -> deconv2 -> mel; Vocos: embed conv
    upsample_split_f16 = False     # FrameStackUpConv: from_latent, 3 residual units (snake + k7 / k1 convs), to_stacked
    conv1_split_f16 = True  # `mixed`: conv1 on the split-f16 MFMA path (mel bins padded 80 -> 96); False: exact f32 (read at pack time)
    varlen_packing = True   # ragged calls: the transformers run on the valid tokens only (packed rows), not on B x longest
    PACK_BELOW = 0.9        # ... when the valid tokens are under this fraction of the padded rows

    def _pack_plan(self, lens_host, T, dt, dev):
        """(row_start device tensor, total rows) when the transformer of this call should run packed, else (None, B*T).
        Packing is exact (rows are independent; attention reads an utterance's own rows only) and exists for the 16-bit
        operand attention kernel (split-f16 / bf16 / fp8 presets)."""
        B = len(lens_host)
        total = sum(lens_host)
        if (not self.varlen_packing or dt == torch.float32 or total == 0 or total >= self.PACK_BELOW * B * T):
            return None, B * T
        cu, acc = [], 0
        for v in lens_host:
            cu.append(acc)
            acc += v
        return self._dev_ints(cu, dev), total

    def _res_units(self, h, units, B, T, C, dt):
        """3 x ResidualUnit (modules.py:37-49). h: [B*T, C] f32, updated in place."""
        M = B * T
        for u in units:
            y = ops.snake_aa(h, u["a0"], u["b0"], u["f0"], B=B, T=T, C_=C, out_dtype=dt)
            y = self._mm(y, u["w1"], M, C, C, lda=C, ldw=7 * C, bias=u["c1"], taps=7, dil=u["dil"], pad=3 * u["dil"],
                         t_in=T, t_out=T)
            y = ops.snake_aa(y, u["a2"], u["b2"], u["f2"], B=B, T=T, C_=C, out_dtype=dt)
            self._mm(y, u["w3"], M, C, C, lda=C, bias=u["c3"], residual=h, out=h)
        return h

    def _logmel(self, wav, n_dev, n_host, P):
        with trace.stage("mel"):
            return self._logmel_impl(wav, n_dev, n_host, P)

    def _logmel_impl(self, wav, n_dev, n_host, P):
        """Whisper log-mel of the valid frames (+2 halo) on the zero-extended, reflect-padded signal
        (feature_extractor.py:86-112). wav: [B, >=max n] f32. Returns mel [B, Tm, 80] (conv1 operand dtype), Tm."""
        B = wav.shape[0]
        Tm = min(spec.MEL_FRAMES, max(spec.mel_len(n) for n in n_host) + 2)
        fr = ops.mel_frames(wav, n_dev, spec.CHUNK_SAMPLES, B=B, T=Tm)
        M = B * Tm
        dft = self._mm(fr, P.dft, M, 402, 400, lda=400)
        pw = ops.mel_power(dft, 402, M, 208)
        mp = self._mm(pw, P.melw, M, P.n_mel, 208, lda=208)
        umax = torch.full((B,), -10.0 if Tm < spec.MEL_FRAMES else float("-inf"), device=wav.device, dtype=torch.float32)
        ops.mel_logmax(mp, P.n_mel, umax, B=B, T=Tm, n_mel=P.n_mel)
        # (ldo = P.c1k: the split-f16 conv1 reads 96 bins per frame; swc_mel_final zeroes the columns beyond n_mel)
        mel = ops.mel_final(mp, P.n_mel, umax, B=B, T=Tm, n_mel=P.n_mel, ldo=P.c1k, out_dtype=P.c1dt)
        return mel, Tm

    def _encoder(self, mel, Tm, t_full, tok_host, P):
        with trace.stage("encoder"):
            return self._encoder_impl(mel, Tm, t_full, tok_host, P)

    def _encoder_impl(self, mel, Tm, t_full, tok_host, P):
        """conv stem + 12 transformer layers + final LayerNorm with the length mask (modules.py:287-376) on frame-major
        mel [B, Tm, n_mel].  t_full: the number of encoder tokens the reference would run (1500 for a 30 s padded call).
        Returns hn [B, Tds*s, D] in the encode operand format, zero beyond each length (the zero-extended encoder output
        the down-sampler reads), Tds."""
        B, dt, D = mel.shape[0], P.edt, P.D
        dev = mel.device
        Ttok = max(1, min(t_full, max(tok_host)))
        K1 = P.c1k
        if K1 != P.n_mel:  # split-f16 conv1: bins zero-padded to a multiple of 32 (by _logmel already), one conversion pass
            if mel.shape[-1] != K1:   # a caller's own [B, T, n_mel] mel (forward(), stage tests)
                mel = torch.nn.functional.pad(mel.to(torch.float32), (0, K1 - mel.shape[-1]))
            mel = ops.cast_f16s(mel.view(B * Tm, K1), K1)
        c1 = self._mm(mel, P.c1w, B * Tm, D, K1, lda=K1, ldw=3 * K1, bias=P.c1b, taps=3, pad=1, t_in=Tm,
                      t_out=Tm, out_dtype=dt)
        h = self._mm(c1, P.c2w, B * Ttok, D, D, lda=D, ldw=3 * D, bias=P.c2b, taps=3, stride=2, pad=1, t_in=Tm, t_out=Ttok)
        tok_c = [min(t, Ttok) for t in tok_host]
        lens = self._dev_ints(tok_c, dev)
        row_start, total = self._pack_plan(tok_c, Ttok, P.enc_ldt, dev)
        if row_start is not None:  # ragged call: the 12 layers see the valid tokens only
            h = ops.pack_rows(h, row_start, lens, B=B, T=Ttok, total=total)
        self._transformer(h, lens, B, Ttok, P.enc_layers, P.He, P.enc_ldt, row_start=row_start)
        s = P.stack
        tds_full = spec.cdiv(t_full, s)
        Tds = min(tds_full, spec.cdiv(Ttok, s) + 64)
        # encoder output is exactly zero beyond each length: the LayerNorm kernel writes those rows as zeros
        # (zero halves are a zero in split-f16 too), so no separate fill pass is needed; a packed stream returns to
        # the padded layout here
        hn = ops.layernorm(h, P.enc_ln[0], P.enc_ln[1], 1e-5, B=B, t_in=Ttok, t_out=Tds * s, C_=D, lens=lens, out_dtype=dt,
                           row_start=row_start)
        return hn, Tds

    def _downsample(self, hn, B, Tds, P):
        with trace.stage("downsample"):
            return self._downsample_impl(hn, B, Tds, P)

    def _downsample_impl(self, hn, B, Tds, P):
        """FrameStackDownConv (modules.py:519-550) on the zero-extended encoder output hn [B, Tds*s, D] (operand format).
        Returns z [B, Tds, lat] f32."""
        dt, s, D = P.edt, P.stack, P.D
        hd = self._mm(hn, P.inw, B * Tds, P.hid, s * D, lda=s * D, bias=P.inb)
        self._res_units(hd, P.down_units, B, Tds, P.hid, dt)
        z = self._mm(self._cast(hd, dt), P.tlw, B * Tds, P.lat, P.hid, lda=P.hid, bias=P.tlb)
        return z.view(B, Tds, P.lat)

    def _encode_mel(self, mel, Tm, t_full, tok_host, P):
        """conv stem + encoder + down-sampler.  Returns z [B, Tds, lat] f32, Tds, latent lengths (host list)."""
        hn, Tds = self._encoder(mel, Tm, t_full, tok_host, P)
        z = self._downsample(hn, mel.shape[0], Tds, P)
        return z, Tds, [spec.cdiv(t, P.stack) for t in tok_host]

    # Vocos receptive field: embed k7 (+-3) + 24 depthwise k7 (+-72) frames, ISTFT overlap +-2 frames (SURVEY.md 8a row V
    # measured -73 ... +74): frames computed this far beyond the kept ones make the kept samples bit-identical
    VOCOS_HALO_FRAMES = 80

    def _upsample(self, zq, B, T, P):
        with trace.stage("upsample"):
            return self._upsample_impl(zq, B, T, P)

    def _upsample_impl(self, zq, B, T, P):
        """FrameStackUpConv (modules.py:601-631, not masked) on zq [B, T, lat] f32.  Returns x [B * s*T, D] f32: to_stacked
        with re-ordered rows makes [B*T, s*D] the un-stacked [B, s*T, D] token stream."""
        dt, s, D = getattr(P, "udt", P.ddt), P.stack, P.Dd
        h = self._mm(self._cast(zq, dt), P.flw, B * T, P.uhid, P.lat, lda=P.lat, bias=P.flb)
        self._res_units(h, P.up_units, B, T, P.uhid, dt)
        x = self._mm(self._cast(h, dt), P.tsw, B * T, s * D, P.uhid, lda=P.uhid, bias=P.tsb)
        return x.view(B * s * T, D)

    def _decoder(self, x, lat_host, B, Tt, P):
        with trace.stage("decoder"):
            return self._decoder_impl(x, lat_host, B, Tt, P)

    def _decoder_impl(self, x, lat_host, B, Tt, P):
        """OmniAudioDecoder (modules.py:437-474): 12 layers (masked), LayerNorm + mask, deconv1 (k3, s2), deconv2 (k3),
        crop to 2*Tt.  x [B*Tt, D] f32 (updated in place).  Returns mel [B, 2*Tt, 80] in the decode operand format."""
        dt, dev, D = P.ddt, x.device, P.Dd
        lens_h = [min(l * P.stack, Tt) for l in lat_host]
        lens = self._dev_ints(lens_h, dev)
        row_start, total = self._pack_plan(lens_h, Tt, dt, dev)
        if row_start is not None:  # tokens beyond a length are masked keys and zeroed outputs: they need not exist
            x = ops.pack_rows(x, row_start, lens, B=B, T=Tt, total=total)
        self._transformer(x, lens, B, Tt, P.dec_layers, P.Hd, dt, row_start=row_start)
        io16 = getattr(P, "io16", dt == torch.float16)
        hn = ops.layernorm(x, P.dec_ln[0], P.dec_ln[1], 1e-5, B=B, t_in=Tt, C_=D, lens=lens,
                           out_dtype=torch.float16 if io16 else dt, row_start=row_start)
        y3 = self._mm(hn.view(B * Tt, -1), P.d1w, B * Tt, 3 * D, D, lda=D)
        Tv = 2 * Tt
        if io16:  # split-f16 operands: the col2im kernel writes f32 / bf16; one conversion pass each
            d1 = ops.deconv_col2im(y3, P.d1b, B=B, T=Tt, C_=D, s=2, t_out=Tv + 1, out_dtype=torch.float32)
            d1 = ops.cast_f16s(d1.view(B * (Tv + 1), D), D)
            mel = torch.zeros((B * Tv, P.vin_k), device=dev, dtype=torch.float32)   # 80 channels + zero columns up to 96
            self._mm(d1, P.d2w, B * Tv, P.vin, D, lda=D, ldw=3 * D, bias=P.d2b, taps=3, pad=2, t_in=Tv + 1, t_out=Tv, out=mel,
                     ldc=P.vin_k)
            return ops.cast_f16s(mel, P.vin_k).view(B, Tv, 2 * P.vin_k)
        d1 = ops.deconv_col2im(y3, P.d1b, B=B, T=Tt, C_=D, s=2, t_out=Tv + 1, out_dtype=dt)
        return self._mm(d1, P.d2w, B * Tv, P.vin, D, lda=D, ldw=3 * D, bias=P.d2b, taps=3, pad=2, t_in=Tv + 1, t_out=Tv,
                        out_dtype=dt).view(B, Tv, P.vin)

    def _decode_latent(self, zq, lat_host, B, T, P, keep_frames=None, ragged=False):
        """up-sampler + decoder + Vocos on zq [B, T, lat] f32 (already masked). Returns wav [B, T*1280] f32, or
        [B, keep_frames*160] when only the first keep_frames Vocos frames are wanted (long-form windows keep 2000
        of their 3000 frames: the local-receptive-field vocoder then runs on 2000 + halo frames).
        ragged (decode() only): samples beyond a row's own length lat_i * 1280 are never looked at, so the ConvNeXt
        blocks skip the 128-frame tiles that lie beyond lat_i * 8 + VOCOS_HALO_FRAMES frames of row i."""
        Tt = P.stack * T
        x = self._upsample(zq, B, T, P)
        mel = self._decoder(x, lat_host, B, Tt, P)
        Tv = 2 * Tt
        if keep_frames is not None and keep_frames < Tv:
            mel = mel[:, :keep_frames].contiguous()
            Tv = keep_frames
        limits = None
        if ragged and self.ragged_vocos:
            lim = [min(Tv, 2 * P.stack * v + self.VOCOS_HALO_FRAMES) if v > 0 else 0 for v in lat_host]
            if sum(spec.cdiv(v, 128) for v in lim) < 0.9 * B * spec.cdiv(Tv, 128):  # worth a per-tile test in every block
                limits = lim
        return self._vocos(mel, B, Tv, P, limits)

    ragged_vocos = True  # decode(): skip ConvNeXt tiles beyond a row's kept frames + halo (bit-identical kept samples)

    def _vocos(self, mel, B, Tv, P, limits=None):
        with trace.stage("vocos"):
            return self._vocos_impl(mel, B, Tv, P, limits)

    def _vocos_impl(self, mel, B, Tv, P, limits=None):
        """Vocos backbone + ISTFT head (modules.py:1492-1504, 1229-1248, 1053-1082, 831-886). mel [B, Tv, 80].
        limits (host ints per row, or None): frames at or beyond limits[b] need not be right (ragged decode)."""
        dt, C, M = P.ddt, P.vdim, B * Tv
        dd16 = dt == torch.float16
        if getattr(P, "io16", dd16) and mel.dtype != torch.float16:  # a caller's own [B, Tv, 80] mel (stage tests, forward()): pad the channels, convert
            mel = ops.cast_f16s(torch.nn.functional.pad(mel.float(), (0, P.vin_k - mel.shape[-1])).reshape(M, P.vin_k), P.vin_k)
        x = self._mm(mel, P.emw, M, C, P.vin_k, lda=P.vin_k, ldw=7 * P.vin_k, bias=P.emb, taps=7, pad=3, t_in=Tv, t_out=Tv)
        x = ops.layernorm(x, P.vnorm[0], P.vnorm[1], 1e-6, B=B, t_in=Tv, C_=C)
        # one fused kernel per block when the grid fills the chip (128-frame tiles, one per CU); small batches keep the
        # two-GEMM form, whose 128 x 128 tiles spread over more CUs
        fused = P.fused_mlp and M >= self.fused_mlp_min_rows
        # the fused block is not in place: two buffers alternate.  With per-row limits the skipped tiles are never
        # written: zero-filled once, so that what lies beyond a row's limit is deterministic (the reference zero-fills
        # beyond the valid length, model.py:356-360) instead of uninitialised memory
        x2 = (torch.zeros_like(x) if limits is not None else torch.empty_like(x)) if fused else None
        lim_dev = self._dev_ints(limits, mel.device) if (fused and limits is not None) else None
        split = self._vocos_split(B, Tv) if (fused and lim_dev is None) else None
        if split is not None:
            x = self._blocks_two_streams(x, x2, B, Tv, C, P, *split)
        else:
            for blk in P.blocks if fused else ():  # the whole block (depthwise conv + LayerNorm + MLP + residual) is one kernel
                ops.convnext_block(x, x2, blk["dw"], blk["db"], blk["ln"][0], blk["ln"][1], 1e-6, blk["ws"], blk["b1"],
                                   blk["b2"], blk["g"], B=B, T=Tv, C_=C, I=P.vint, t_limit=lim_dev,
                                   operands=torch.float16 if getattr(P, "cx_f16", False) else torch.bfloat16)
                x, x2 = x2, x
        for blk in P.blocks:
            if fused:
                break
            y = ops.dwconv7_ln(x, blk["dw"], blk["db"], blk["ln"][0], blk["ln"][1], 1e-6, B=B, T=Tv, C_=C,
                               out_dtype=torch.float32 if dd16 else dt)
            if dd16:
                y = ops.cast_f16s(y.view(M, C), C)
            y = self._mm(y, blk["w1"], M, P.vint, C, lda=C, bias=blk["b1"], act=ops.ACT_GELU, out_dtype=dt)
            self._mm(y, blk["w2"], M, C, P.vint, lda=P.vint, bias=blk["b2"], gamma=blk["g"], residual=x, out=x)
        t16 = getattr(P, "tail16", dd16)
        hn = ops.layernorm(x, P.vfin[0], P.vfin[1], 1e-6, B=B, t_in=Tv, C_=C, out_dtype=torch.float16 if t16 else dt)
        ho = torch.empty((M, 648), device=mel.device, dtype=torch.float32)  # ld 648: 16-byte rows for vector stores
        self._mm(hn.view(M, -1), P.hw, M, 642, C, lda=C, bias=P.hb, out=ho, ldc=648)
        sp = ops.istft_spec(ho, 648, M, P.idft_k, out_dtype=torch.float16 if t16 else dt)   # (split-f16 written directly)
        fr = self._mm(sp, P.idft, M, 640, P.idft_k, lda=P.idft_k)
        return ops.istft_ola(fr, P.wsq, B=B, T=Tv)

    length_bucketing = True  # encode() / decode(): rows of similar length share a call (exact; False: one call per batch)
    bucket_overhead_tokens = 5000.0  # fixed cost of one more call, in row-tokens (length_groups)
    DECODE_HALO_CODES = 64   # code frames of the batch padding a row's kept samples can depend on (up-sampler +-59)
    vocos_streams = 1  # 1: one launch per ConvNeXt block over the whole batch; 2: two out-of-phase half-batch chains on two
    #                    streams (bit-identical; measured -1.3 % decode time at B = 32 x 10 s, tools/ab_streams.py: opt-in)
    vocos_phase_us = 130  # start of the second half-batch chain after the first (half a block launch)

    CUS = 256  # one 128-frame workgroup of the fused ConvNeXt kernel per CU and round
    vocos_split_override = None

    def _vocos_split(self, B, Tv):
        """(utterances in the first chain, phase shift in us, chip share of a launch) for running the ConvNeXt blocks as two
        chains on two streams, or None for one launch per block over the whole batch.
          * grid quantisation: when the last round of 128-frame tiles is poorly filled (32 x 2080 frames = 520 tiles = two
            full rounds + 8 tiles: 3 rounds of workgroup time for 2.03 rounds of work), the batch is cut in two halves
            whose chains of launches run beside each other: utterances are independent through the whole backbone, so
            the chains never wait for each other and the dispatcher always has workgroups of one of them to place
            (B = 32 x 30 s decode 46.5 -> 41.5 ms; peeling off only the last utterance gave 44.2, tools/ab_split.py);
          * vocos_streams = 2 (opt-in): two out-of-phase half-batch chains."""
        if self.vocos_split_override is not None:  # tuning (tools/ab_streams.py): utterances in the first chain, or 0
            h = int(self.vocos_split_override)
            return (h, 0, 1.0) if 0 < h < B else None
        tiles = spec.cdiv(B * Tv, 128)
        rounds = spec.cdiv(tiles, self.CUS)
        if rounds >= 2 and tiles < 0.85 * rounds * self.CUS and B >= 2:
            return B // 2, 0, 0.5
        if self.vocos_streams == 2 and B % 2 == 0 and (B // 2) * Tv >= 64 * 128:
            return B // 2, self.vocos_phase_us, 0.5
        return None

    def _blocks_two_streams(self, x, x2, B, Tv, C, P, h, phase_us, share):
        """ConvNeXt blocks of utterances [0, h) on the current stream and of [h, B) on a side stream (two independent chains
        of launches over disjoint rows of the same two residual buffers)."""
        dev = x.device
        st = self.__dict__.get("_side")
        if st is None or st["dev"] != dev:
            st = {"dev": dev, "stream": torch.cuda.Stream(device=dev)}
            self.__dict__["_side"] = st
        side, main = st["stream"], torch.cuda.current_stream(dev)
        xf, x2f = x.view(B * Tv, C), x2.view(B * Tv, C)
        xa, xb = (xf[: h * Tv], xf[h * Tv:]), (x2f[: h * Tv], x2f[h * Tv:])  # parts of the two buffers; roles swap per block
        nb = (h, B - h)
        # profiling only: two half-batch chains share the chip evenly; a small side chain runs inside the main chain's time
        shares = (share, share) if share < 1.0 else (1.0, 0.0)
        side.wait_stream(main)  # the embed / LayerNorm outputs exist

        def run(blk, cur, part):
            src, dst = (xa[part], xb[part]) if cur == 0 else (xb[part], xa[part])
            ops.convnext_block(src, dst, blk["dw"], blk["db"], blk["ln"][0], blk["ln"][1], 1e-6, blk["ws"], blk["b1"],
                               blk["b2"], blk["g"], B=nb[part], T=Tv, C_=C, I=P.vint, chip_share=shares[part],
                               operands=torch.float16 if getattr(P, "cx_f16", False) else torch.bfloat16)
        # an out-of-phase second chain starts half a launch late (events only fire at launch boundaries, which would put the
        # chains back in phase: a one-wave delay kernel shifts it) and stays out of phase: equal launches, in-order streams
        cur = 0
        with torch.cuda.stream(side):
            if phase_us:
                ops.delay_us(phase_us)
            for blk in P.blocks:
                run(blk, cur, 1)
                cur ^= 1
        cur = 0
        for blk in P.blocks:
            run(blk, cur, 0)
            cur ^= 1
        main.wait_stream(side)
        return x if len(P.blocks) % 2 == 0 else x2

    # short int lists (lengths) go to the device through a ring of pinned staging rows with non-blocking copies:
    # torch.tensor(list, device=...) copies from pageable memory and synchronises the stream every time
    _PIN_SLOTS, _PIN_LEN = 32, 4096

    def _dev_ints(self, values, dev, dtype=torch.int32, cache_it=True):
        n = len(values)
        dev = torch.device(dev)
        per = 1 if dtype == torch.int32 else 2  # int64 entries take two int32 slots of the staging row
        if n == 0 or n * per > self._PIN_LEN or dev.type != "cuda":
            return torch.tensor(values, dtype=dtype, device=dev)
        # recurring shapes (serving, benchmarks, graph capture) re-use the uploaded tensor: nothing is copied at all
        cache = self.__dict__.setdefault("_ints_cache", {})
        key = (dev, dtype, tuple(values)) if cache_it else None
        hit = cache.get(key) if cache_it else None
        if hit is not None:
            return hit
        if torch.cuda.is_current_stream_capturing():
            raise SwcError("a length list first seen during graph capture: run the same shapes once before capturing")
        if len(cache) >= 512:
            cache.clear()
        st = self.__dict__.get("_pin")
        if st is None or st["dev"] != dev:
            st = {"dev": dev, "buf": torch.empty((self._PIN_SLOTS, self._PIN_LEN), dtype=torch.int32).pin_memory(),
                  "ev": [None] * self._PIN_SLOTS, "i": 0}
            self.__dict__["_pin"] = st
        i = st["i"]
        st["i"] = (i + 1) % self._PIN_SLOTS
        if st["ev"][i] is not None:
            st["ev"][i].synchronize()  # the copy that last used this row finished 32 uploads ago
        row = st["buf"][i, :n * per].view(dtype)
        row.copy_(torch.tensor(values, dtype=dtype))
        out = torch.empty(n, dtype=dtype, device=dev)
        out.copy_(row, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        st["ev"][i] = ev
        if cache_it:
            cache[key] = out
        return out

    # ------------------------------------------------- reference entry points
    @_on_model_device
    @torch.inference_mode()
    def inference_tokenize(self, x, input_lengths):
        """model.py:167-210. x (B, 1, T<=480000) on the device, input_lengths (B,).
        Returns zq (B, D, 375), codes (G, B, 375) int32, codes_lengths (B,) int64."""
        B = x.shape[0]
        wav = x.reshape(B, -1).to(torch.float32)
        # the reference slices xi[:, :x_len] (model.py:180): a length beyond the row is the row
        n_host = [min(int(v), spec.CHUNK_SAMPLES, wav.shape[1]) for v in
                  (input_lengths.tolist() if torch.is_tensor(input_lengths) else input_lengths)]
        if wav.stride(-1) != 1:
            wav = wav.contiguous()
        dev = wav.device
        n_dev = self._dev_ints(n_host, dev)
        tok = [spec.token_len(n) for n in n_host]

        def run(P):
            mel, Tm = self._logmel(wav, n_dev, n_host, P)
            z, Tds, lat = self._encode_mel(mel, Tm, spec.MEL_FRAMES // 2, tok, P)
            t_pad = spec.cdiv(spec.MEL_FRAMES // 2, P.stack)  # 375: the reference always returns the padded length
            lat_dev = self._dev_ints(lat, dev)
            zq, codes = ops.fsq_encode(z, P.lat, lat_dev, P.fsq, B=B, T=Tds, t_pad=t_pad, G=self.num_groups, levels=self.fsq_levels)
            return {"zq": zq.transpose(1, 2), "codes": codes, "codes_lengths": lat_dev.long()}
        return self._guarded_encode(run)

    @_on_model_device
    @torch.inference_mode()
    def inference_detokenize(self, codes, codes_lengths, _keep_samples=None, _ragged=False):
        """model.py:212-242. codes (G, B, T) integer, codes_lengths (B,). Returns y (B, 1, T*1280), output_length.
        _keep_samples (internal, decode()): only the first _keep_samples samples of every row are needed; y is then
        shorter than T*1280 but those samples are bit-identical.  _ragged (internal, decode()): samples of row i beyond
        codes_lengths[i] * 1280 are never looked at (they may then differ from the padded computation)."""
        P = self._packed()
        G, B, T = codes.shape
        lat = [int(v) for v in (codes_lengths.tolist() if torch.is_tensor(codes_lengths) else codes_lengths)]
        dev = codes.device
        lat_dev = self._dev_ints(lat, dev)
        zq = ops.fsq_decode(codes.to(torch.int64).contiguous(), lat_dev, B=B, T=T, G=G, levels=self.fsq_levels)
        kf = None
        if _keep_samples is not None:
            hop = self.generator_params["vocos"]["hop_size"]
            kf = spec.cdiv(_keep_samples, hop) + self.VOCOS_HALO_FRAMES
        wav = self._decode_latent(zq, lat, B, T, P, keep_frames=kf, ragged=_ragged)
        return {"y": wav[:, None, :], "output_length": lat_dev.long() * self.decoder_upsample_rate}

    # long-form scheduling: windows of one call are independent rows, so several 30 s windows are batched into
    # one tokenize / detokenize call (the reference loops them serially, model.py:275,340)
    max_rows_per_call = 64
    fused_mlp_min_rows = 128 * 160  # Vocos frames from which the fused ConvNeXt MLP kernel is used (>= 160 of 256 CUs busy)
    trim_vocos = True  # decode(): run Vocos only on the kept frames (+ halo) of each window; bit-identical output

    def _stack(self, tensors, lens, dev, dtype, min_len=0):
        """list of 1-D tensors -> zero-padded [B, max(len, min_len)] on dev.  Device-resident 4-byte rows (the normal case)
        are assembled by one gather kernel from an uploaded address list instead of one copy per utterance."""
        L = max(max(lens), 1, int(min_len))
        if (dev.type == "cuda" and dtype in (torch.float32, torch.int32) and len(tensors) <= 65535
                and all(t.dtype == dtype and t.device == dev and t.is_contiguous() for t in tensors)):
            addr = [t.data_ptr() for t in tensors]
            if not any(a & 3 for a in addr):
                ptrs = self._dev_ints(addr, dev, torch.int64, cache_it=False)
                nbytes = self._dev_ints([4 * n for n in lens], dev, torch.int64)
                return ops.gather_rows(ptrs, nbytes, len(tensors), L, dtype, dev)
        if len(set(lens)) == 1 and lens[0] == L and all(t.device == dev and t.dtype == dtype for t in tensors):
            return torch.stack([t.reshape(-1) for t in tensors])
        out = torch.zeros(len(tensors), L, device=dev, dtype=dtype)
        for i, t in enumerate(tensors):
            if lens[i]:
                out[i, : lens[i]] = t.reshape(-1)
        return out

    @_on_model_device
    @torch.inference_mode()
    def encode(self, wav_list, overlap_seconds=10, device=torch.device("cuda")):
        """model.py:244-308: 30 s windows every (30 - overlap) s, keep the first 250 codes of each window,
        concatenate, trim to len // 1280.  Returns {"codes_list": [IntTensor(G, T_i)]}."""
        B = len(wav_list)
        if B == 0:
            return {"codes_list": []}
        n = [int(w.shape[-1]) if w.dim() else 0 for w in wav_list]
        dev = self._resolve_device(device)
        # Rows are independent, so the batch is assembled longest utterance first: workgroups are dispatched row by row
        # (the batch index is the slowest grid dimension of the attention / LayerNorm / snake grids), and with the long
        # rows first the short ones fill in behind them (longest-processing-time order) instead of leaving the chip to
        # the last long row.  Results are handed back in the caller's order.
        order = sorted(range(B), key=lambda i: -n[i])
        ns = [n[i] for i in order]
        rate = self.encoder_downsample_rate
        out = [None] * B
        # Ragged batches: every call runs all its rows at the longest row's token count (as the reference does, at 30 s
        # always), so rows of similar length are encoded together (length_groups; exact: rows are independent).  Windows
        # beyond the first of a long recording are full 30 s windows for every row that has them: only the length inside
        # the first window enters the grouping.
        chunk = int(self.max_audio_seconds * self.input_sample_rate)
        # (with valid-token packing the 12 layers of a ragged call cost their valid tokens anyway, and what stays padded —
        # log-mel, conv stem, down-sampler — is 5 % of the encoder: one call then beats several)
        packs = self.varlen_packing and PRECISIONS[self._precision][0] != "f32"
        groups = (length_groups([spec.token_len(min(v, chunk)) for v in ns], self.bucket_overhead_tokens)
                  if self.length_bucketing and not packs else [(0, B)])
        for a, b in groups:
            wav = self._stack([wav_list[i] for i in order[a:b]], ns[a:b], dev, torch.float32)
            allc = self._encode_padded(wav, ns[a:b], overlap_seconds)
            for k in range(a, b):
                i = order[k]
                out[i] = (torch.zeros(self.num_groups, 0, device=dev, dtype=torch.long) if allc is None
                          else allc[:, k - a, : n[i] // rate])
        return {"codes_list": out}

    @_on_model_device
    @torch.inference_mode()
    def encode_padded(self, wav, n, overlap_seconds=10):
        """The batch form of encode() (the data-parallel wrapper's fast path): wav [B, L] f32 on the model's GPU, zero
        padded, n = list of valid lengths.  Returns codes (G, B, Lc) int32, zero beyond n_i // 1280 (Lc >= max of them),
        or None when no utterance reaches one code frame."""
        return self._encode_padded(wav, [int(v) for v in n], overlap_seconds)

    def _encode_padded(self, wav, n, overlap_seconds):
        sr, rate = self.input_sample_rate, self.encoder_downsample_rate
        chunk = int(self.max_audio_seconds * sr)
        dur = int((self.max_audio_seconds - overlap_seconds) * sr)
        keep = dur // rate
        B, dev = wav.shape[0], wav.device
        L = max(n)
        wins = []
        for c in range(spec.cdiv(L, dur) if L else 0):
            s0, e0 = c * dur, min(c * dur + chunk, L)
            cl = [min(max(v - s0, 0), e0 - s0) for v in n]
            if max(cl) > 0:
                wins.append((s0, e0, cl))
        if not wins:
            return None
        # Windows are independent rows, so several of them share one tokenize call (rows = windows x utterances) — but
        # only windows of the same padded length: the encoder of a call runs every row at the longest row's token count,
        # and the last window of a recording is short (32 x 30 s: 1500-token windows + 500-token tails; batched together
        # the tails would be computed at 1500 tokens, 50 % more encoder work than the two calls cost)
        parts = [None] * len(wins)
        per_call = max(1, self.max_rows_per_call // B)
        # ... unless the transformer of the call runs on packed valid tokens anyway (varlen_packing): then all windows go
        # together again (one call of 64 000 valid tokens instead of 48 000 + 16 000) and only log-mel / conv stem /
        # down-sampler (5 % of the encoder) see the padding
        packs = self.varlen_packing and PRECISIONS[self._precision][0] != "f32"
        by_len = {}
        for i, wdw in enumerate(wins):
            by_len.setdefault(0 if packs else wdw[1] - wdw[0], []).append(i)
        for _, idx in by_len.items():
            for w0 in range(0, len(idx), per_call):
                grp = idx[w0:w0 + per_call]
                wl = max(wins[i][1] - wins[i][0] for i in grp)
                if len(grp) == 1:
                    s0, e0, cl = wins[grp[0]]
                    x, lens = wav[:, None, s0:e0], cl
                else:
                    if all(wins[i][1] - wins[i][0] == wl for i in grp):
                        x = torch.stack([wav[:, wins[i][0]:wins[i][1]] for i in grp]).view(len(grp) * B, 1, wl)
                    else:
                        x = torch.zeros(len(grp) * B, 1, wl, device=dev)
                        for k, i in enumerate(grp):
                            x[k * B:(k + 1) * B, 0, : wins[i][1] - wins[i][0]] = wav[:, wins[i][0]:wins[i][1]]
                    lens = [v for i in grp for v in wins[i][2]]
                r = self.inference_tokenize(x, lens)
                # codes beyond an utterance's length are already zero (FSQ kernel masks them, quantizer.py:193-196);
                # keeping the first `keep` frames of every window reproduces model.py:291-297 without the copy loop
                for k, i in enumerate(grp):
                    parts[i] = r["codes"][:, k * B:(k + 1) * B, :keep]
        return torch.cat(parts, dim=-1) if len(parts) > 1 else parts[0]

    @_on_model_device
    @torch.inference_mode()
    def decode(self, codes_list, overlap_seconds=10, device=torch.device("cuda"), pad_to_length=None):
        """model.py:310-373: 375-code windows every 250 codes, keep the first 320000 samples of each,
        concatenate, trim to T_i * 1280.  Returns {"syn_wav_list": [FloatTensor(T_i * 1280)]}.
        pad_to_length (extension): pad the batch to at least this many code frames — a shard of a larger
        batch passes the global maximum so that its results equal the un-sharded call (the reference's
        up-sampler / Vocos are not masked, so outputs depend on the padded batch length)."""
        B = len(codes_list)
        if B == 0:
            return {"syn_wav_list": []}
        n_in = [int(c.shape[-1]) for c in codes_list]
        L = max(max(n_in), int(pad_to_length or 0))
        dev = self._resolve_device(device)
        if L == 0:
            return {"syn_wav_list": [torch.zeros(0, device=dev) for _ in range(B)]}
        order = sorted(range(B), key=lambda i: -n_in[i])  # longest first, as in encode(); rows are independent
        codes_list = [codes_list[i] for i in order]
        n = [n_in[i] for i in order]
        G = self.num_groups
        dt0 = codes_list[0].dtype
        if (dev.type == "cuda" and dt0 in (torch.int32, torch.int64) and B * G <= 65535
                and all(c.device == dev and c.dtype == dt0 and c.dim() == 2 and c.shape[0] == G
                        and (c.shape[1] == 0 or c.stride(1) == 1) for c in codes_list)):
            # one gather kernel over the B x G code rows (they may be strided views of encode()'s output)
            es = 4 if dt0 == torch.int32 else 8
            ptrs = [c.data_ptr() + g * c.stride(0) * es for c in codes_list for g in range(G)]
            nb = [es * v for v in n for _ in range(G)]
            rows = ops.gather_rows(self._dev_ints(ptrs, dev, torch.int64, cache_it=False),
                                   self._dev_ints(nb, dev, torch.int64), B * G, L, dt0, dev)
            codes = rows.view(B, G, L).permute(1, 0, 2).to(torch.long)
        elif len(set(n)) == 1 and n[0] == L and all(c.device == dev and c.dtype == dt0 for c in codes_list):
            codes = torch.stack(list(codes_list), dim=1).to(torch.long)  # one conversion, not one per utterance
        else:
            codes = torch.zeros(self.num_groups, B, L, device=dev, dtype=torch.long)
            for i, c in enumerate(codes_list):
                codes[:, i, : n[i]] = c.to(dev)
        # Ragged batches.  The reference pads every row to the batch maximum and its un-masked up-sampler / Vocos read that
        # padding — but only within their receptive fields: the up-sampler reaches +-59 code frames, the decoder
        # transformer is masked by length and its output is zeroed beyond it, Vocos + ISTFT reach +-10 code frames.  A row
        # decoded at min(L, n_i + 64) code frames therefore gives bit-identical samples below n_i * 1280, and rows of
        # similar length share a call (length_groups) instead of all running at the longest row's length.
        up = self.decoder_upsample_rate
        out = [None] * B
        need = [min(L, v + self.DECODE_HALO_CODES) for v in n]
        # (measured, tools/bench_ragged.py: once the decoder transformer runs packed, one call beats the grouped calls —
        # what stays padded, up-sampler and Vocos, loses more to small calls than it saves; presets without the packed
        # attention kernel keep the grouping)
        packs = self.varlen_packing and PRECISIONS[self._precision][1] != "f32"
        bucket = self.length_bucketing and not packs
        groups = length_groups([4 * v for v in need], self.bucket_overhead_tokens) if bucket else [(0, B)]
        for a, b in groups:
            Lg = L if not bucket else max(need[a:b])
            wav = self._decode_padded(codes[:, a:b, :Lg], n[a:b], overlap_seconds)
            for k in range(a, b):
                out[order[k]] = wav[k - a, : n[k] * up]
        return {"syn_wav_list": out}

    @_on_model_device
    @torch.inference_mode()
    def decode_padded(self, codes, n, overlap_seconds=10, pad_to_length=None):
        """The batch form of decode(): codes (G, B, L) integer on the model's GPU, zero padded, n = valid code counts;
        pad_to_length as in decode().  Returns wav [B, max(L, pad_to_length) * 1280] f32 (row i valid up to n_i * 1280)."""
        L = max(int(codes.shape[-1]), int(pad_to_length or 0))
        codes = codes.to(torch.long)
        if L > codes.shape[-1]:
            codes = torch.nn.functional.pad(codes, (0, L - codes.shape[-1]))
        return self._decode_padded(codes, [int(v) for v in n], overlap_seconds)

    def _decode_padded(self, codes, n, overlap_seconds):
        sr, rate = self.input_sample_rate, self.encoder_downsample_rate
        win = int(self.max_audio_seconds * sr // rate)
        step = int((self.max_audio_seconds - overlap_seconds) * sr // rate)
        keep = step * self.decoder_upsample_rate
        B, L = codes.shape[1], codes.shape[2]
        wins = []
        for c in range(spec.cdiv(L, step)):
            s0, e0 = c * step, min(c * step + win, L)
            cl = [min(max(v - s0, 0), e0 - s0) for v in n]
            if max(cl) > 0 or c == 0:
                wins.append((c, s0, e0, cl))
        # windows of EQUAL padded length are independent rows of one call; a shorter (last) window must keep its
        # own padded length, because the un-masked up-sampler / Vocos see that boundary
        parts = {}
        per_call = max(1, self.max_rows_per_call // B)
        by_len = {}
        for wdw in wins:
            by_len.setdefault(wdw[2] - wdw[1], []).append(wdw)
        for tlen, ws in by_len.items():
            for w0 in range(0, len(ws), per_call):
                grp = ws[w0:w0 + per_call]
                if len(grp) == 1:
                    cg, lens = codes[:, :, grp[0][1]:grp[0][2]].contiguous(), grp[0][3]
                else:
                    cg = torch.cat([codes[:, :, s0:e0] for _, s0, e0, _ in grp], dim=1)
                    lens = [v for *_, cl in grp for v in cl]
                y = self.inference_detokenize(cg, lens, _keep_samples=keep if self.trim_vocos else None, _ragged=True)["y"]
                # samples beyond an utterance's valid length fall after its final trim (n_i * 1280), so the
                # zero-fill of model.py:356-360 is unobservable: keep the first `keep` samples of each window
                for k, wdw in enumerate(grp):
                    parts[wdw[0]] = y[k * B:(k + 1) * B, 0, :keep]
        order = sorted(parts)
        return torch.cat([parts[c] for c in order], dim=-1) if len(order) > 1 else parts[order[0]]

    @_on_model_device
    @torch.inference_mode()
    def forward(self, batch):
        """model.py:112-165: mel (B, n_mel, T) + mel_lens -> {'reconstructed_audio' (B,1,T_audio), 'audio_lengths'}.
        No 30 s padding and no chunking; the un-masked up-sampler sees the whole padded length."""
        mel_in, ml = batch["mel_features"], batch["mel_lens"]
        B, _, T = mel_in.shape
        ml_host = [int(v) for v in ml.tolist()]
        mel32 = mel_in.to(torch.float32).transpose(1, 2).contiguous()
        dev = mel32.device
        t_full = (T - 1) // 2 + 1  # Conv1d(k3, s2, p1) output length
        tok = [m // 2 for m in ml_host]
        t_lat = spec.cdiv(t_full, self.generator_params["downsample"]["stack_factor"])

        def run(P):
            # the trimmed stem needs frames up to 2*max(tok): the given T is the true right boundary
            z, Tds, lat = self._encode_mel(self._cast(mel32, P.c1dt), T, t_full, tok, P)
            lat_dev = self._dev_ints(lat, dev)
            zq, _ = ops.fsq_encode(z, P.lat, lat_dev, P.fsq, B=B, T=Tds, t_pad=t_lat, G=self.num_groups, levels=self.fsq_levels)
            return zq, lat, lat_dev
        zq, lat, lat_dev = self._guarded_encode(run)
        P = self._packed()
        wav = self._decode_latent(zq, lat, B, t_lat, P)
        return {"reconstructed_audio": wav[:, None, :],
                "audio_lengths": lat_dev.long() * (P.stack * 2 * self.generator_params["vocos"]["hop_size"])}
