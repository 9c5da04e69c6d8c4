"""Several batches in flight on one GPU (the serving loop around the hot path; the reference's loop is serial,
inference.py:38-64).

A batch's kernels leave the chip idle in places no single launch can fill: the store tails of GEMM epilogues, the HBM phases
at both ends of a ConvNeXt block, kernels whose grid covers a fraction of the 256 CUs, the host read-back of the range
counters between encode and decode.  A second, independent batch running on its own HIP stream falls into exactly those
holes — two host threads, two streams, two AudioCodec objects over one set of device-resident operands
(AudioCodec.replica()).  Results are those of the serial loop, bit for bit: every batch runs the same kernels on the same
data, only beside another batch.
"""
import threading
from concurrent.futures import ThreadPoolExecutor

import torch


class InFlight:
    """`depth` batches in flight (default 2).  map(fn, items) calls fn(model, item) for every item, on `depth` worker threads
    with their own stream and model replica, and returns the results in order; each result is complete (its stream has been
    synchronised) when it is handed back."""

    def __init__(self, model, depth=2):
        self.device = model._buffers_device()
        if self.device.type != "cuda":
            raise RuntimeError("InFlight needs a model on the HIP device")
        self.models = [model] + [model.replica() for _ in range(depth - 1)]
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(depth)]
        self._free = list(range(depth))
        self._lock = threading.Lock()
        self._pool = ThreadPoolExecutor(max_workers=depth, thread_name_prefix="swc-inflight")

    def _run(self, fn, item, ready):
        with self._lock:
            k = self._free.pop()
        try:
            with torch.cuda.device(self.device), torch.cuda.stream(self.streams[k]):
                self.streams[k].wait_event(ready)  # the caller's stream has produced the item
                out = fn(self.models[k], item)
                self.streams[k].synchronize()
            return out
        finally:
            with self._lock:
                self._free.append(k)

    def submit(self, fn, item):
        """-> Future of fn(model, item); at most `depth` run at a time, the rest wait in submission order"""
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        return self._pool.submit(self._run, fn, item, ready)

    def map(self, fn, items):
        futs = [self.submit(fn, it) for it in items]
        return [f.result() for f in futs]

    def close(self):
        self._pool.shutdown(wait=True)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
