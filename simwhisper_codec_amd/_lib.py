"""ctypes binding of libswc_hip.so (the C-ABI declared in include/swc.h).

The product path has no fallback: if the library is missing or a call fails, an
exception is raised (`SwcError`).  `import torch` happens first so that the HIP
runtime the library resolves (`libamdhip64.so.7`) is the one PyTorch already loaded —
streams and device pointers are then shared between the two.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede CDLL: shares PyTorch's HIP runtime)

HERE = os.path.dirname(os.path.abspath(__file__))
# SWC_LIB: an alternative build of the same ABI (A/B benchmarking of compiler flags / kernel variants)
LIB_PATH = os.environ.get("SWC_LIB") or os.path.join(HERE, "libswc_hip.so")

F32, BF16, F16S, FP8 = 0, 1, 2, 3
F16S_ACT_SCALE = 64.0  # SWC_F16S_ACT_SCALE in include/swc.h
FP8_ACT_SCALE = 16.0   # SWC_FP8_ACT_SCALE
ACT_NONE, ACT_GELU = 0, 1


class SwcError(RuntimeError):
    pass


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("W", C.c_void_p), ("C", C.c_void_p),
        ("bias", C.c_void_p), ("gamma", C.c_void_p), ("residual", C.c_void_p),
        ("lda", C.c_int64), ("ldw", C.c_int64), ("ldc", C.c_int64), ("ldr", C.c_int64),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("taps", C.c_int32), ("dil", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("t_in", C.c_int32), ("t_out", C.c_int32),
        ("a_dtype", C.c_int32), ("c_dtype", C.c_int32), ("act", C.c_int32),
        ("alpha", C.c_float), ("out_scale", C.c_float),
    ]


_P, _I, _L, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_float

# name -> argtypes (all return int); mirrors include/swc.h one to one
SIGNATURES = {
    "swc_gemm": [C.POINTER(GemmArgs), _P],
    "swc_attention": [_P, _P, _P, _I, _I, _I, _I, _P],
    "swc_attention_ex": [_P, _P, _P, _I, _I, _I, _I, _P],
    "swc_attention16": [_P, _P, _P, _I, _I, _I, _I, _P, _P],
    "swc_layernorm": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P, _P],
    "swc_dwconv7_ln": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _P],
    "swc_snake_aa": [_P, _P, _P, _P, C.POINTER(_F), _I, _I, _I, _I, _P],
    "swc_fsq_encode": [_P, _L, _P, _P, _P, C.POINTER(_F), _I, _I, _I, _I, _P],
    "swc_fsq_decode": [_P, _P, _L, _P, _I, _I, _I, _P],
    "swc_fsq_encode_levels": [_P, _L, _P, _P, _P, C.POINTER(_F), C.POINTER(C.c_int32), _I, _I, _I, _I, _P],
    "swc_fsq_decode_levels": [_P, _P, _L, _P, C.POINTER(C.c_int32), _I, _I, _I, _P],
    "swc_mel_frames": [_P, _L, _P, _I, _P, _I, _I, _P],
    "swc_mel_power": [_P, _L, _P, _L, _L, _P],
    "swc_mel_logmax": [_P, _L, _P, _I, _I, _I, _P],
    "swc_mel_final": [_P, _L, _P, _P, _L, _I, _I, _I, _I, _P],
    "swc_deconv_col2im": [_P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P],
    "swc_istft_spec": [_P, _L, _P, _L, _L, _I, _P],
    "swc_istft_ola": [_P, _P, _P, _I, _I, _P],
    "swc_codes_pack": [_P, _L, _P, _I, _P],
    "swc_codes_unpack": [_P, _P, _L, _I, _P],
    "swc_cast_f32_bf16": [_P, _P, _L, _P],
    "swc_cast_f32_f16s": [_P, _L, _P, _L, _I, _F, _P],
    "swc_cast_fp8": [_P, _I, _P, _L, _F, _P],
    "swc_gather_rows": [_P, _P, _P, _L, _I, _P],
    "swc_pcm16_to_f32": [_P, _P, _L, _P],
    "swc_f32_to_pcm16": [_P, _P, _L, _P],
    "swc_set_saturation_counter": [_P],
    "swc_delay_us": [_I, _P],
    "swc_pack_rows": [_P, _P, _P, _P, _I, _I, _L, _P],
    "swc_convnext_pack": [_P, _P, _P, _P, _I, _I, _P],
    "swc_convnext_mlp": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "swc_convnext_block": [_P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _I, _I, _I, _I, _P, _I, _P],
    "swc_convnext64_pack": [_P, _P, _P, _I, _I, _P],
    "swc_convnext64_mlp": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "swc_mlp_pack": [_P, _P, _P, _I, _I, _P],
    "swc_mlp_block": [_P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "swc_layer_tail_pack": [_P, _P, _P, _P, _I, _I, _I, _P],
    "swc_layer_tail": [_P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P],
    "swc_proj_ln_pack": [_P, _P, _I, _I, _P],
    "swc_proj_ln": [_P, _L, _P, _P, _F, _P, _P, _P, _P, _F, _P, _I, _I, _I, _P],
}
PLAIN = {"swc_version": ([], C.c_int), "swc_last_error": ([], C.c_char_p), "swc_device_count": ([], C.c_int),
         "swc_convnext_stream_bytes": ([_I, _I], C.c_int64),
         "swc_mlp_stream_bytes": ([_I, _I], C.c_int64), "swc_convnext64_stream_bytes": ([_I, _I], C.c_int64), "swc_layer_tail_stream_bytes": ([_I, _I, _I], C.c_int64),
         "swc_proj_ln_stream_bytes": ([_I, _I], C.c_int64)}

_lib = None


def load():
    """Load the shared library once; raise SwcError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SwcError(
            f"{LIB_PATH} is missing: build it with `python -m simwhisper_codec_amd.build` "
            "(there is no CPU or PyTorch fallback for the hot path)")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    for name, (argtypes, restype) in PLAIN.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().swc_last_error().decode("utf-8", "replace")
        raise SwcError(f"{what} failed (rc={rc}): {msg}")


def exported_symbols():
    return list(SIGNATURES) + list(PLAIN)
