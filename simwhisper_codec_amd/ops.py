"""Torch-tensor front ends of the C-ABI kernels.  PyTorch is plumbing here: it owns
device memory and the stream; every computation is a libswc_hip.so call.

All activations are frame-major: a reference (B, C, T) tensor lives as [B, T, C].
"""
import ctypes as C

import torch

from . import _lib
from ._lib import ACT_GELU, ACT_NONE, BF16, F16S, F16S_ACT_SCALE, F32, FP8, FP8_ACT_SCALE  # noqa: F401

# torch.float16 tensors carry the split-f16 format (SWC_F16S): last dim = 2 x logical columns
# torch.float8_e4m3fn tensors carry SWC_FP8 bytes (OCP e4m3fn, gfx950's native fp8)
FP8_T = torch.float8_e4m3fn
_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16S, FP8_T: FP8}


def _w(dtype, cols):
    """storage width (tensor columns) of `cols` logical columns"""
    return 2 * cols if dtype == torch.float16 else cols

# Optional per-launch timing hook (bench.py): an object with begin(kind, work) / end(), called
# around every swc_gemm launch on the launching stream.  None in normal operation.
PROFILER = None
LEGACY_ATTENTION = False  # tests only: route bf16 attention through the f32-MFMA kernel


def _stream():
    # raw handle of torch's current stream on the current device (torch.cuda.current_stream() costs ~8 us per call)
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _chk(t, name, dtype=None):
    if not t.is_cuda:
        raise _lib.SwcError(f"{name}: expected a device tensor (the HIP path has no CPU fallback)")
    if t.device.index != torch._C._cuda_getDevice():
        raise _lib.SwcError(f"{name}: tensor on cuda:{t.device.index} but cuda:{torch._C._cuda_getDevice()} is current; kernels "
                            "run on the current device's stream (AudioCodec's entry points set it; direct ops callers "
                            "use torch.cuda.device)")
    if dtype is not None and t.dtype != dtype:
        raise _lib.SwcError(f"{name}: expected {dtype}, got {t.dtype}")
    return t


def gemm(A, W, M, N, K, *, out=None, out_dtype=None, lda=None, ldw=None, ldc=None, bias=None, gamma=None,
         residual=None, ldr=None, act=ACT_NONE, taps=1, dil=1, stride=1, pad=0, t_in=None, t_out=None, alpha=1.0,
         out_scale=1.0):
    """C[M, N] = epi(A (*) W^T); see include/swc.h swc_gemm.  A: [.., lda], W: [N, ldw].
    float16 tensors are split-f16 (2 halves per logical column); lda/ldw/ldc are LOGICAL columns."""
    lib = _lib.load()
    _chk(A, "gemm A"); _chk(W, "gemm W")
    if A.dtype != W.dtype:
        raise _lib.SwcError(f"gemm: A is {A.dtype} but W is {W.dtype}")
    sa = 2 if A.dtype == torch.float16 else 1
    lda = A.stride(-2) // sa if lda is None else lda
    ldw = W.stride(-2) // sa if ldw is None else ldw
    if out is None:
        od = out_dtype or torch.float32
        out = torch.empty((M, _w(od, N)), device=A.device, dtype=od)
    ldc = out.stride(-2) // (2 if out.dtype == torch.float16 else 1) if ldc is None else ldc
    a = _lib.GemmArgs()
    a.A, a.W, a.C = A.data_ptr(), W.data_ptr(), out.data_ptr()
    a.bias = bias.data_ptr() if bias is not None else None
    a.gamma = gamma.data_ptr() if gamma is not None else None
    a.residual = residual.data_ptr() if residual is not None else None
    a.lda, a.ldw, a.ldc = lda, ldw, ldc
    a.ldr = (residual.stride(-2) if ldr is None else ldr) if residual is not None else 0
    a.M, a.N, a.K = M, N, K
    a.taps, a.dil, a.stride, a.pad = taps, dil, stride, pad
    a.t_in = M if t_in is None else t_in
    a.t_out = M if t_out is None else t_out
    if M == 0:
        return out
    a.a_dtype, a.c_dtype, a.act = _DT[A.dtype], _DT[out.dtype], act
    a.alpha, a.out_scale = alpha, out_scale
    prof = PROFILER
    if prof is not None:
        prof.begin({torch.bfloat16: "gemm_bf16", torch.float16: "gemm_f16s", FP8_T: "gemm_fp8"}.get(A.dtype, "gemm_f32"),
                   2.0 * M * N * K * taps)
    _lib.check(lib.swc_gemm(C.byref(a), _stream()), "swc_gemm")
    if prof is not None:
        prof.end()
    return out


def attention(qkv, lens, B, T, H, out=None, out_dtype=None, row_start=None, rows=None):
    """out_dtype None: same element type as qkv.  torch.float16: f32 qkv in, split-f16 out.
    row_start (int32 device tensor, 16-bit operands only): packed layout, `rows` = total number of packed rows."""
    lib = _lib.load()
    _chk(qkv, "attention qkv"); _chk(lens, "attention lens", torch.int32)
    od = out_dtype or qkv.dtype
    if out is None:
        shape = (B, T, _w(od, H * 64)) if row_start is None else (rows, _w(od, H * 64))
        out = torch.empty(shape, device=qkv.device, dtype=od)
    if row_start is not None and not (qkv.dtype in (torch.bfloat16, torch.float16) and od == qkv.dtype and not LEGACY_ATTENTION):
        raise _lib.SwcError("attention: the packed layout exists for the 16-bit operand kernel only")
    if qkv.dtype in (torch.bfloat16, torch.float16) and od == qkv.dtype and not LEGACY_ATTENTION:
        _lib.check(lib.swc_attention16(_ptr(qkv), _ptr(out), _ptr(lens), B, T, H, _DT[qkv.dtype], _ptr(row_start), _stream()),
                   "swc_attention16")
    elif od == qkv.dtype:
        _lib.check(lib.swc_attention(_ptr(qkv), _ptr(out), _ptr(lens), B, T, H, _DT[qkv.dtype], _stream()),
                   "swc_attention")
    else:
        _chk(qkv, "attention qkv", torch.float32)
        _lib.check(lib.swc_attention_ex(_ptr(qkv), _ptr(out), _ptr(lens), B, T, H, _DT[od], _stream()),
                   "swc_attention_ex")
    return out


def layernorm(x, w, b, eps, *, B, t_in, C_, t_out=None, lens=None, out=None, out_dtype=torch.float32, row_start=None):
    """row_start (int32 device tensor, with lens): x is packed (utterance b's rows start at row_start[b]); out is padded."""
    lib = _lib.load()
    _chk(x, "layernorm x", torch.float32)
    t_out = t_in if t_out is None else t_out
    if out is None:
        out = torch.empty((B, t_out, _w(out_dtype, C_)), device=x.device, dtype=out_dtype)
    _lib.check(lib.swc_layernorm(_ptr(x), _ptr(out), _ptr(w), _ptr(b), _ptr(lens), B, t_in, t_out, C_, eps,
                                 _DT[out.dtype], _ptr(row_start), _stream()), "swc_layernorm")
    return out


def pack_rows(x, row_start, lens, *, B, T, total):
    """padded [B*T, C] (4-byte elements) -> packed [total, C]: the first lens[b] rows of every utterance, back to back."""
    lib = _lib.load()
    _chk(x, "pack_rows x"); _chk(row_start, "pack_rows row_start", torch.int32); _chk(lens, "pack_rows lens", torch.int32)
    Cw = x.shape[-1]
    out = torch.empty((total, Cw), device=x.device, dtype=x.dtype)
    _lib.check(lib.swc_pack_rows(_ptr(x), _ptr(out), _ptr(row_start), _ptr(lens), B, T, Cw * x.element_size(), _stream()),
               "swc_pack_rows")
    return out


def dwconv7_ln(x, w7, bias, ln_w, ln_b, eps, *, B, T, C_, out=None, out_dtype=torch.float32):
    lib = _lib.load()
    _chk(x, "dwconv7_ln x", torch.float32)
    if out is None:
        out = torch.empty((B, T, C_), device=x.device, dtype=out_dtype)
    _lib.check(lib.swc_dwconv7_ln(_ptr(x), _ptr(out), _ptr(w7), _ptr(bias), _ptr(ln_w), _ptr(ln_b), B, T, C_, eps,
                                  _DT[out.dtype], _stream()), "swc_dwconv7_ln")
    return out


def snake_aa(x, alpha, beta, filt12, *, B, T, C_, out=None, out_dtype=torch.float32):
    """filt12: python sequence of the 12 kaiser-sinc taps (host)."""
    lib = _lib.load()
    _chk(x, "snake_aa x", torch.float32)
    if out is None:
        out = torch.empty((B, T, _w(out_dtype, C_)), device=x.device, dtype=out_dtype)
    f = (C.c_float * 12)(*[float(v) for v in filt12])
    _lib.check(lib.swc_snake_aa(_ptr(x), _ptr(out), _ptr(alpha), _ptr(beta), f, B, T, C_, _DT[out.dtype],
                                _stream()), "swc_snake_aa")
    return out


FSQ_LEVELS = (8, 7, 6, 6)  # config/SimWhisperCodec.yaml; any four levels work (swc_fsq_*_levels)


def fsq_encode(z, ldz, lens, consts12, *, B, T, t_pad, G, levels=FSQ_LEVELS):
    lib = _lib.load()
    _chk(z, "fsq_encode z", torch.float32); _chk(lens, "fsq_encode lens", torch.int32)
    zq = torch.empty((B, t_pad, 4 * G), device=z.device, dtype=torch.float32)
    codes = torch.empty((G, B, t_pad), device=z.device, dtype=torch.int32)
    k = (C.c_float * 12)(*[float(v) for v in consts12])
    lv = (C.c_int32 * 4)(*[int(v) for v in levels])
    _lib.check(lib.swc_fsq_encode_levels(_ptr(z), ldz, _ptr(zq), _ptr(codes), _ptr(lens), k, lv, B, T, t_pad, G, _stream()),
               "swc_fsq_encode_levels")
    return zq, codes


def fsq_decode(codes, lens, *, B, T, G, ldq=None, levels=FSQ_LEVELS):
    lib = _lib.load()
    _chk(codes, "fsq_decode codes", torch.int64); _chk(lens, "fsq_decode lens", torch.int32)
    ldq = 4 * G if ldq is None else ldq
    zq = torch.empty((B, T, ldq), device=codes.device, dtype=torch.float32)
    lv = (C.c_int32 * 4)(*[int(v) for v in levels])
    _lib.check(lib.swc_fsq_decode_levels(_ptr(codes), _ptr(zq), ldq, _ptr(lens), lv, B, T, G, _stream()), "swc_fsq_decode_levels")
    return zq


def mel_frames(wav, n, n_pad, *, B, T):
    lib = _lib.load()
    _chk(wav, "mel_frames wav", torch.float32); _chk(n, "mel_frames n", torch.int32)
    frames = torch.empty((B, T, 400), device=wav.device, dtype=torch.float32)
    _lib.check(lib.swc_mel_frames(_ptr(wav), wav.stride(0), _ptr(n), n_pad, _ptr(frames), B, T, _stream()),
               "swc_mel_frames")
    return frames


def mel_power(dft, ld, rows, ldp):
    lib = _lib.load()
    pw = torch.empty((rows, ldp), device=dft.device, dtype=torch.float32)
    _lib.check(lib.swc_mel_power(_ptr(dft), ld, _ptr(pw), ldp, rows, _stream()), "swc_mel_power")
    return pw


def mel_logmax(mel, ld, umax, *, B, T, n_mel):
    lib = _lib.load()
    _lib.check(lib.swc_mel_logmax(_ptr(mel), ld, _ptr(umax), B, T, n_mel, _stream()), "swc_mel_logmax")


def mel_final(mel, ld, umax, *, B, T, n_mel, ldo, out_dtype=torch.float32):
    lib = _lib.load()
    out = torch.empty((B, T, ldo), device=mel.device, dtype=out_dtype)
    _lib.check(lib.swc_mel_final(_ptr(mel), ld, _ptr(umax), _ptr(out), ldo, B, T, n_mel, _DT[out_dtype], _stream()),
               "swc_mel_final")
    return out


def deconv_col2im(y3, bias, *, B, T, C_, s, t_out, ldo=None, out_dtype=torch.float32):
    lib = _lib.load()
    ldo = C_ if ldo is None else ldo
    out = torch.empty((B, t_out, ldo), device=y3.device, dtype=out_dtype)
    _lib.check(lib.swc_deconv_col2im(_ptr(y3), _ptr(bias), _ptr(out), ldo, B, T, C_, s, t_out, _DT[out_dtype],
                                     _stream()), "swc_deconv_col2im")
    return out


def istft_spec(h, ldh, rows, lds, out_dtype=torch.float32):
    lib = _lib.load()
    s = torch.empty((rows, _w(out_dtype, lds)), device=h.device, dtype=out_dtype)   # (float16 = split-f16: 2 halves per column)
    _lib.check(lib.swc_istft_spec(_ptr(h), ldh, _ptr(s), lds, rows, _DT[out_dtype], _stream()), "swc_istft_spec")
    return s


def istft_ola(frames, window_sq, *, B, T):
    lib = _lib.load()
    wav = torch.empty((B, T * 160), device=frames.device, dtype=torch.float32)
    _lib.check(lib.swc_istft_ola(_ptr(frames), _ptr(window_sq), _ptr(wav), B, T, _stream()), "swc_istft_ola")
    return wav


def cast_bf16(x):
    lib = _lib.load()
    _chk(x, "cast_bf16 x", torch.float32)
    x = x.contiguous()
    y = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    _lib.check(lib.swc_cast_f32_bf16(_ptr(x), _ptr(y), x.numel(), _stream()), "swc_cast_f32_bf16")
    return y


def cast_fp8(x, scale=FP8_ACT_SCALE):
    """f32 / bf16 tensor * scale -> e4m3 bytes (torch.float8_e4m3fn storage), saturating."""
    lib = _lib.load()
    _chk(x, "cast_fp8 x")
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise _lib.SwcError(f"cast_fp8: expected f32 or bf16, got {x.dtype}")
    x = x.contiguous()
    y = torch.empty(x.shape, device=x.device, dtype=FP8_T)
    _lib.check(lib.swc_cast_fp8(_ptr(x), _DT[x.dtype], _ptr(y), x.numel(), scale, _stream()), "swc_cast_fp8")
    return y


def cast_f16s(x, K, scale=F16S_ACT_SCALE, ldx=None):
    """f32 [rows, >=K] -> split-f16 [rows, 2K] (float16 storage), x * scale split into hi/lo halves."""
    lib = _lib.load()
    _chk(x, "cast_f16s x", torch.float32)
    rows = x.numel() // x.shape[-1]
    ldx = x.stride(-2) if ldx is None else ldx
    y = torch.empty((rows, 2 * K), device=x.device, dtype=torch.float16)
    _lib.check(lib.swc_cast_f32_f16s(_ptr(x), ldx, _ptr(y), rows, K, scale, _stream()), "swc_cast_f32_f16s")
    return y


def gather_rows(ptrs_dev, nbytes_dev, n_rows, ld_elems, dtype, device):
    """n_rows device buffers (int64 device arrays of addresses / byte counts) -> zero-padded [n_rows, ld_elems]."""
    lib = _lib.load()
    out = torch.empty((n_rows, ld_elems), device=device, dtype=dtype)
    _lib.check(lib.swc_gather_rows(_ptr(ptrs_dev), _ptr(nbytes_dev), _ptr(out), ld_elems * out.element_size(), n_rows,
                                   _stream()), "swc_gather_rows")
    return out


def pcm16_to_f32(pcm):
    """int16 device tensor -> f32 of the same shape, pcm * 2^-15 (include/swc.h swc_pcm16_to_f32)."""
    lib = _lib.load()
    _chk(pcm, "pcm16_to_f32 pcm", torch.int16)
    if not pcm.is_contiguous():
        raise _lib.SwcError("pcm16_to_f32: contiguous input expected")
    out = torch.empty(pcm.shape, device=pcm.device, dtype=torch.float32)
    _lib.check(lib.swc_pcm16_to_f32(_ptr(pcm), _ptr(out), pcm.numel(), _stream()), "swc_pcm16_to_f32")
    return out


def f32_to_pcm16(x):
    """f32 device tensor -> int16 of the same shape, round(clip(x, -1, 1) * 32767) (include/swc.h swc_f32_to_pcm16)."""
    lib = _lib.load()
    _chk(x, "f32_to_pcm16 x", torch.float32)
    if not x.is_contiguous():
        raise _lib.SwcError("f32_to_pcm16: contiguous input expected")
    out = torch.empty(x.shape, device=x.device, dtype=torch.int16)
    _lib.check(lib.swc_f32_to_pcm16(_ptr(x), _ptr(out), x.numel(), _stream()), "swc_f32_to_pcm16")
    return out


def set_saturation_counter(counters):
    """counters: int32/uint32 device tensor of 2 elements (or None): see swc_set_saturation_counter in include/swc.h.
    The pointer is per calling thread; the tensor must outlive its use."""
    lib = _lib.load()
    if counters is not None:
        _chk(counters, "saturation counters")
        if counters.numel() < 2 or counters.element_size() != 4:
            raise _lib.SwcError("saturation counters: need 2 x 32-bit elements")
    _lib.check(lib.swc_set_saturation_counter(_ptr(counters)), "swc_set_saturation_counter")


def convnext_supported(C_, I):
    return _lib.load().swc_convnext_stream_bytes(C_, I) > 0


def convnext_pack(w1, w2, gamma):
    """pwconv1.weight [I, C], pwconv2.weight [C, I] (bf16, device) and the block's gamma [C] (f32) -> the packed operand stream of
    swc_convnext_mlp / swc_convnext_block (gamma is folded into pwconv2's rows by CX_RES_ACC builds).
    Both weights may instead be PLAIN float16 [I, C] / [C, I] (not the split-f16 layout): the stream then serves
    convnext_block(..., operands=torch.float16)."""
    lib = _lib.load()
    _chk(w1, "convnext_pack w1"); _chk(w2, "convnext_pack w2", w1.dtype)
    if w1.dtype not in (torch.bfloat16, torch.float16):
        raise _lib.SwcError(f"convnext_pack: weights must be bf16 or plain f16, got {w1.dtype}")
    _chk(gamma, "convnext_pack gamma", torch.float32)
    I, C_ = w1.shape
    if tuple(w2.shape) != (C_, I):
        raise _lib.SwcError(f"convnext_pack: w2 is {tuple(w2.shape)}, expected {(C_, I)}")
    n = lib.swc_convnext_stream_bytes(C_, I)
    if n <= 0:
        raise _lib.SwcError(f"convnext_pack: unsupported geometry C={C_} I={I}")
    out = torch.empty(n, dtype=torch.uint8, device=w1.device)
    _lib.check(lib.swc_convnext_pack(_ptr(w1.contiguous()), _ptr(w2.contiguous()), _ptr(gamma.contiguous()), _ptr(out), C_, I,
                                     _stream()), "swc_convnext_pack")
    return out


def convnext_mlp(y, w_stream, b1, b2, gamma, x, *, M, C_, I):
    """x[M, C] += gamma * (GELU(y W1^T + b1) W2^T + b2) in one kernel; y bf16 [M, C], x f32 [M, C] (in place)."""
    lib = _lib.load()
    _chk(y, "convnext_mlp y", torch.bfloat16); _chk(x, "convnext_mlp x", torch.float32)
    prof = PROFILER
    if prof is not None:
        prof.begin("convnext_bf16", 4.0 * M * C_ * I)
    _lib.check(lib.swc_convnext_mlp(_ptr(y), _ptr(w_stream), _ptr(b1), _ptr(b2), _ptr(gamma), _ptr(x), M, C_, I, _stream()),
               "swc_convnext_mlp")
    if prof is not None:
        prof.end()
    return x


def convnext64_pack(w1, w2):
    """EXPERIMENT (swc_convnext64_mlp): pwconv1.weight [I, C] and pwconv2.weight [C, I] (bf16, device) -> its operand stream."""
    lib = _lib.load()
    _chk(w1, "convnext64_pack w1", torch.bfloat16); _chk(w2, "convnext64_pack w2", torch.bfloat16)
    I, C_ = w1.shape
    n = lib.swc_convnext64_stream_bytes(C_, I)
    if n <= 0 or tuple(w2.shape) != (C_, I):
        raise _lib.SwcError(f"convnext64_pack: unsupported geometry C={C_} I={I}")
    out = torch.empty(n, dtype=torch.uint8, device=w1.device)
    _lib.check(lib.swc_convnext64_pack(_ptr(w1.contiguous()), _ptr(w2.contiguous()), _ptr(out), C_, I, _stream()), "swc_convnext64_pack")
    return out


def convnext64_mlp(y, w_stream, b1, b2, gamma, x, *, M, C_, I, stagger_cycles=0):
    """EXPERIMENT: swc_convnext_mlp on 64-frame tiles, two workgroups per CU, optional start stagger (shader cycles)."""
    lib = _lib.load()
    _chk(y, "convnext64_mlp y", torch.bfloat16); _chk(x, "convnext64_mlp x", torch.float32)
    _lib.check(lib.swc_convnext64_mlp(_ptr(y), _ptr(w_stream), _ptr(b1), _ptr(b2), _ptr(gamma), _ptr(x), M, C_, I, int(stagger_cycles),
                                      _stream()), "swc_convnext64_mlp")
    return x


def abi_version():
    return int(_lib.load().swc_version())


SWC_F16 = 4  # include/swc.h: plain half precision, the internal operand type of swc_convnext_block (operands=torch.float16)


def convnext_block(x, x_out, w7, dw_bias, ln_w, ln_b, eps, w_stream, b1, b2, gamma, *, B, T, C_, I, chip_share=1.0,
                   t_limit=None, operands=torch.bfloat16):
    """One whole ConvNeXt block: residual stream x [B, T, C] f32 -> x_out (a different buffer; swc_convnext_block).
    operands: torch.bfloat16, or torch.float16 = PLAIN half precision inside the kernel (w_stream packed from float16 weights).
    t_limit (int32 device tensor [B], ragged batches): frames at or beyond t_limit[b] need not be computed.
    chip_share (profiling only): the fraction of the chip this launch runs on when another chain runs beside it on a
    second stream; the timing hook charges duration x chip_share, so that TFLOP/s stays a whole-chip rate."""
    lib = _lib.load()
    _chk(x, "convnext_block x", torch.float32); _chk(x_out, "convnext_block x_out", torch.float32)
    prof = PROFILER
    if prof is not None:
        prof.begin("convnext_bf16", 4.0 * B * T * C_ * I, chip_share)
    _lib.check(lib.swc_convnext_block(_ptr(x), _ptr(x_out), _ptr(w7), _ptr(dw_bias), _ptr(ln_w), _ptr(ln_b), eps, _ptr(w_stream),
                                      _ptr(b1), _ptr(b2), _ptr(gamma), B, T, C_, I, _ptr(t_limit),
                                      SWC_F16 if operands == torch.float16 else _DT[operands], _stream()), "swc_convnext_block")
    if prof is not None:
        prof.end()
    return x_out


def mlp_supported(D, F):
    return _lib.load().swc_mlp_stream_bytes(D, F) > 0


def mlp_pack(w1, w2):
    """fc1.weight [F, D] and fc2.weight [D, F] (bf16, device) -> the packed operand stream of swc_mlp_block."""
    lib = _lib.load()
    _chk(w1, "mlp_pack w1", torch.bfloat16); _chk(w2, "mlp_pack w2", torch.bfloat16)
    F_, D = w1.shape
    if tuple(w2.shape) != (D, F_):
        raise _lib.SwcError(f"mlp_pack: w2 is {tuple(w2.shape)}, expected {(D, F_)}")
    n = lib.swc_mlp_stream_bytes(D, F_)
    if n <= 0:
        raise _lib.SwcError(f"mlp_pack: unsupported geometry D={D} F={F_}")
    out = torch.empty(n, dtype=torch.uint8, device=w1.device)
    _lib.check(lib.swc_mlp_pack(_ptr(w1.contiguous()), _ptr(w2.contiguous()), _ptr(out), D, F_, _stream()), "swc_mlp_pack")
    return out


def mlp_block(x, ln_w, ln_b, eps, w_stream, b1, b2, *, M, D, F, x_out=None, next_ln=None, y_next=None):
    """One transformer MLP sub-block in one kernel (swc_mlp_block): x [M, D] f32 residual stream -> x_out (default: in place).
    next_ln = (weight, bias) of the LayerNorm that follows (the next layer's self_attn_layer_norm): its bf16 output is
    returned as the second value (None without next_ln)."""
    lib = _lib.load()
    _chk(x, "mlp_block x", torch.float32)
    x_out = x if x_out is None else _chk(x_out, "mlp_block x_out", torch.float32)
    nw = nb = None
    if next_ln is not None:
        nw, nb = next_ln
        if y_next is None:
            y_next = torch.empty((M, D), device=x.device, dtype=torch.bfloat16)
        _chk(y_next, "mlp_block y_next", torch.bfloat16)
    else:
        y_next = None
    prof = PROFILER
    if prof is not None:
        prof.begin("mlp_bf16", 4.0 * M * D * F)
    _lib.check(lib.swc_mlp_block(_ptr(x), _ptr(x_out), _ptr(ln_w), _ptr(ln_b), eps, _ptr(w_stream), _ptr(b1), _ptr(b2), _ptr(nw),
                                 _ptr(nb), _ptr(y_next), M, D, F, _stream()), "swc_mlp_block")
    if prof is not None:
        prof.end()
    return x_out, y_next


def layer_tail_pack(wo, w1, w2):
    """out_proj.weight [D, D] (bf16), fc1.weight [F, D] (bf16, or e4m3 at a per-tensor scale: preset fp8_fc1), fc2.weight [D, F]
    (bf16), all on the device -> the operand stream of swc_layer_tail for that fc1 operand type."""
    lib = _lib.load()
    _chk(wo, "layer_tail_pack wo"); _chk(w2, "layer_tail_pack w2", wo.dtype)
    _chk(w1, "layer_tail_pack w1")
    if wo.dtype not in (torch.bfloat16, torch.float16):
        raise _lib.SwcError(f"layer_tail_pack: weights must be bf16 (or PLAIN float16 for layer_tail(operands=torch.float16)), got {wo.dtype}")
    if w1.dtype not in (wo.dtype, FP8_T) or (w1.dtype == FP8_T and wo.dtype != torch.bfloat16):
        raise _lib.SwcError(f"layer_tail_pack: fc1 weights must be {wo.dtype} (or e4m3 beside bf16), got {w1.dtype}")
    F_, D = w1.shape
    if tuple(w2.shape) != (D, F_) or tuple(wo.shape) != (D, D):
        raise _lib.SwcError(f"layer_tail_pack: shapes {tuple(wo.shape)} {tuple(w1.shape)} {tuple(w2.shape)}")
    f1 = FP8 if w1.dtype == FP8_T else BF16   # (BF16 = "16-bit elements": the pack step re-orders them whatever their format)
    n = lib.swc_layer_tail_stream_bytes(D, F_, f1)
    if n <= 0:
        raise _lib.SwcError(f"layer_tail_pack: unsupported geometry D={D} F={F_}")
    out = torch.empty(n, dtype=torch.uint8, device=w1.device)
    _lib.check(lib.swc_layer_tail_pack(_ptr(wo.contiguous()), _ptr(w1.contiguous()), _ptr(w2.contiguous()), _ptr(out), D, F_,
                                       f1, _stream()), "swc_layer_tail_pack")
    return out


def layer_tail(attn, x, w_stream, bo, ln_w, ln_b, eps, b1, b2, *, M, D, F, x_out=None, next_ln=None, y_next=None,
               fc1_dtype=torch.bfloat16, fc1_alpha=1.0, operands=torch.bfloat16):
    """Everything of a transformer layer behind the attention in one kernel (swc_layer_tail): attn [M, D] bf16 (attention
    output), x [M, D] f32 residual stream -> x_out (default: in place) and, with next_ln = (weight, bias), the bf16
    LayerNorm output the next layer's q/k/v projection reads (second return value).  fc1_dtype = FP8_T: the stream was packed
    from e4m3 fc1 weights (scale sw) and fc1_alpha = 1 / (FP8_ACT_SCALE * sw).  operands = torch.float16: PLAIN half precision
    inside the kernel (the stream packed from float16 weights); attn and y_next stay bf16."""
    lib = _lib.load()
    _chk(attn, "layer_tail attn", torch.bfloat16); _chk(x, "layer_tail x", torch.float32)
    x_out = x if x_out is None else _chk(x_out, "layer_tail x_out", torch.float32)
    if w_stream.numel() != lib.swc_layer_tail_stream_bytes(D, F, _DT[fc1_dtype]):
        raise _lib.SwcError(f"layer_tail: the operand stream ({w_stream.numel()} bytes) was not packed for fc1 operands {fc1_dtype}")
    nw = nb = None
    if next_ln is not None:
        nw, nb = next_ln
        if y_next is None:
            y_next = torch.empty((M, D), device=x.device, dtype=torch.bfloat16)
        _chk(y_next, "layer_tail y_next", torch.bfloat16)
    else:
        y_next = None
    prof = PROFILER
    if prof is not None:
        prof.begin("mlp_bf16" if fc1_dtype == torch.bfloat16 else "mlp_fp8fc1", 4.0 * M * D * F + 2.0 * M * D * D)
    _lib.check(lib.swc_layer_tail(_ptr(attn), _ptr(x), _ptr(x_out), _ptr(w_stream), _ptr(bo), _ptr(ln_w), _ptr(ln_b), eps, _ptr(b1),
                                  _ptr(b2), _ptr(nw), _ptr(nb), _ptr(y_next), M, D, F, _DT[fc1_dtype], float(fc1_alpha),
                                  SWC_F16 if operands == torch.float16 else _DT[operands], _stream()),
               "swc_layer_tail")
    if prof is not None:
        prof.end()
    return x_out, y_next


def proj_ln_pack(w):
    """A split-f16 weight [N, 2K] (fp16 tensor, SWC_F16S rows) on the device -> the operand stream of swc_proj_ln."""
    lib = _lib.load()
    _chk(w, "proj_ln_pack w", torch.float16)
    N, K = w.shape[0], w.shape[1] // 2
    n = lib.swc_proj_ln_stream_bytes(N, K)
    if n <= 0:
        raise _lib.SwcError(f"proj_ln_pack: unsupported geometry N={N} K={K}")
    out = torch.empty(n, dtype=torch.uint8, device=w.device)
    _lib.check(lib.swc_proj_ln_pack(_ptr(w.contiguous()), _ptr(out), N, K, _stream()), "swc_proj_ln_pack")
    return out


def proj_ln(a, w_stream, bias, alpha, x, *, M, N, K, lda=None, x_out=None, ln=None, eps=1e-5, y_next=None):
    """x_out = x + alpha * (a W^T) + bias (default: in place) and, with ln = (weight, bias), LayerNorm(x_out) as the split-f16
    operand of the next GEMM (second return value) in one kernel (swc_proj_ln).  a: split-f16 [M, 2 lda]."""
    lib = _lib.load()
    _chk(a, "proj_ln a", torch.float16); _chk(x, "proj_ln x", torch.float32)
    x_out = x if x_out is None else _chk(x_out, "proj_ln x_out", torch.float32)
    lda = a.stride(-2) // 2 if lda is None else lda
    if w_stream.numel() != lib.swc_proj_ln_stream_bytes(N, K):
        raise _lib.SwcError(f"proj_ln: the operand stream ({w_stream.numel()} bytes) was not packed for N={N} K={K}")
    lw = lb = None
    if ln is not None:
        lw, lb = ln
        if y_next is None:
            y_next = torch.empty((M, 2 * N), device=x.device, dtype=torch.float16)
        _chk(y_next, "proj_ln y_next", torch.float16)
    else:
        y_next = None
    prof = PROFILER
    if prof is not None:
        prof.begin("gemm_f16s", 2.0 * M * N * K)
    _lib.check(lib.swc_proj_ln(_ptr(a), lda, _ptr(w_stream), _ptr(bias), float(alpha), _ptr(x), _ptr(x_out), _ptr(lw), _ptr(lb),
                               float(eps), _ptr(y_next), M, N, K, _stream()), "swc_proj_ln")
    if prof is not None:
        prof.end()
    return x_out, y_next


def delay_us(us):
    """occupy the current stream for `us` microseconds (phase shift between two streams)"""
    _lib.check(_lib.load().swc_delay_us(int(us), _stream()), "swc_delay_us")
