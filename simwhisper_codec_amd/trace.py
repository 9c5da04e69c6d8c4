"""Per-stage ranges for profilers and per-stage device timers (tracing row).

    SWC_TRACE=roctx   roctx ranges around every stage of the path (mel, encoder, downsample, fsq, upsample, decoder,
                      vocos, istft, ...): `rocprofv3 --marker-trace --kernel-trace` then attributes kernels to stages
                      without matching kernel names.
    SWC_TRACE=time    device-event pairs around every stage on the launching stream; trace.report() returns the
                      accumulated milliseconds per stage (it synchronises the device).
    SWC_TRACE=roctx,time  both.   Unset (default): stage() is a shared no-op context manager.

The reference has no tracing at all; stage names follow its sub-modules (model.py:181-237).
"""
import contextlib
import ctypes
import os

import torch

_MODE = {m for m in os.environ.get("SWC_TRACE", "").replace(" ", "").split(",") if m}
_roctx = None
_events = {}  # stage -> [(start event, end event)]


def _load_roctx():
    global _roctx
    if _roctx is None:
        for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
            try:
                lib = ctypes.CDLL(name)
                lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                lib.roctxRangePushA.restype = ctypes.c_int
                lib.roctxRangePop.restype = ctypes.c_int
                _roctx = lib
                break
            except (OSError, AttributeError):
                continue
        else:
            raise RuntimeError("SWC_TRACE=roctx: neither librocprofiler-sdk-roctx.so nor libroctx64.so could be loaded")
    return _roctx


def configure(mode):
    """Set the mode programmatically: "", "roctx", "time" or "roctx,time" (tools and tests; the env var is the default)."""
    global _MODE
    _MODE = {m for m in (mode or "").replace(" ", "").split(",") if m}
    _events.clear()


_NULL = contextlib.nullcontext()


class _Stage:
    __slots__ = ("name", "e0")

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if "roctx" in _MODE:
            _load_roctx().roctxRangePushA(("swc/" + self.name).encode())
        if "time" in _MODE:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if "time" in _MODE:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            _events.setdefault(self.name, []).append((self.e0, e1))
        if "roctx" in _MODE:
            _load_roctx().roctxRangePop()
        return False


def stage(name):
    return _Stage(name) if _MODE else _NULL


def report(reset=True):
    """{stage: {"ms": total, "calls": n}} of the time mode; synchronises the device."""
    torch.cuda.synchronize()
    out = {k: {"ms": round(sum(a.elapsed_time(b) for a, b in v), 4), "calls": len(v)} for k, v in _events.items()}
    if reset:
        _events.clear()
    return out
