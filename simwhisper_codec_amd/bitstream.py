"""On-disk / on-wire code format.  The reference keeps codes only in memory
(`(8, T)` int32, 11 of 32 bits used, model.py:302); this is the packed form that makes the codec a
codec: 8 x 11 bits = 11 bytes per 12.5 Hz frame = 1100 bit/s.

File layout (little endian):  b"SWC1" | u32 n_frames | u8 n_groups (8) | u8 bits (11) | u16 reserved |
                              11 * n_frames payload bytes (frame-major, group g at bits 11g..11g+10, LSB first).
Packing / unpacking run on the device (swc_codes_pack / swc_codes_unpack).
"""
import ctypes as C
import struct

import torch

from . import _lib
from .ops import _ptr, _stream

MAGIC = b"SWC1"
GROUPS, BITS, FRAME_BYTES = 8, 11, 11


def pack_codes(codes):
    """codes: device IntTensor (8, T) with values < 2048 (11 bits: the shipped [8, 7, 6, 6] levels give 2016 codes per group; a
    config with a larger codebook or another group count needs another container) -> device ByteTensor (11 * T,)."""
    lib = _lib.load()
    if codes.dim() != 2 or codes.shape[0] != GROUPS:
        raise _lib.SwcError(f"pack_codes: expected ({GROUPS}, T) codes, got {tuple(codes.shape)}")
    if not codes.is_cuda:
        raise _lib.SwcError("pack_codes: expected a device tensor")
    c = codes.to(torch.int32).contiguous()
    T = c.shape[1]
    out = torch.empty(FRAME_BYTES * T, device=c.device, dtype=torch.uint8)
    _lib.check(lib.swc_codes_pack(_ptr(c), c.stride(0) if T else 0, _ptr(out), T, _stream()), "swc_codes_pack")
    return out


def unpack_codes(payload, n_frames):
    """device ByteTensor (11 * T,) -> device IntTensor (8, T)."""
    lib = _lib.load()
    if not payload.is_cuda or payload.dtype != torch.uint8 or payload.numel() < FRAME_BYTES * n_frames:
        raise _lib.SwcError("unpack_codes: expected a device uint8 tensor of at least 11 * n_frames bytes")
    codes = torch.empty(GROUPS, n_frames, device=payload.device, dtype=torch.int32)
    _lib.check(lib.swc_codes_unpack(_ptr(payload.contiguous()), _ptr(codes), n_frames, n_frames, _stream()),
               "swc_codes_unpack")
    return codes


def write_codes(path, codes):
    payload = pack_codes(codes).cpu().numpy().tobytes()
    with open(path, "wb") as f:
        f.write(MAGIC + struct.pack("<IBBH", codes.shape[1], GROUPS, BITS, 0) + payload)


def read_codes(path, device="cuda"):
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != MAGIC:
        raise ValueError(f"{path}: not a SWC1 code file")
    n, g, b, _ = struct.unpack("<IBBH", data[4:12])
    if g != GROUPS or b != BITS or len(data) < 12 + FRAME_BYTES * n:
        raise ValueError(f"{path}: unsupported or truncated code file")
    payload = torch.frombuffer(bytearray(data[12:12 + FRAME_BYTES * n]), dtype=torch.uint8).to(device)
    return unpack_codes(payload, n)
