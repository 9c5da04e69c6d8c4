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


class HostStager:
    """One host <-> device copy per batch instead of one per file (the file loop of inference.py).

    to_device(): the utterances of a batch are packed back to back into a pinned staging buffer (one per calling thread, grown on
    demand, re-used) and cross PCIe as ONE asynchronous copy on the current stream; the model gets views of that one device buffer
    (AudioCodec.encode gathers its rows by address, so views cost nothing).  32 separate pinned allocations + 32 copies took
    14 ms + 5 ms per 32 x 10 s batch on the GPU box, one copy takes under 2 ms.  The caller must have synchronised the stream
    of its previous batch before it stages the next one (InFlight does: a result is complete when it is handed back).
    to_host(): decode() returns rows of one padded buffer; that buffer is copied once and the same views are taken on the host.
    """

    def __init__(self):
        self._tls = threading.local()

    def to_device(self, cpu_tensors, device):
        lens = [int(t.numel()) for t in cpu_tensors]
        total = sum(lens)
        if total == 0 or any(t.dtype != torch.float32 or t.device.type != "cpu" for t in cpu_tensors):
            return [t.to(device, non_blocking=True) for t in cpu_tensors]
        # every utterance starts on a 16-byte boundary of the device buffer (the vector loads of the framing kernel)
        offs, pos = [], 0
        for n in lens:
            offs.append(pos)
            pos += (n + 3) // 4 * 4
        buf = getattr(self._tls, "buf", None)
        if buf is None or buf.numel() < pos:
            buf = torch.empty(max(pos, 1 << 20), dtype=torch.float32).pin_memory()
            self._tls.buf = buf
        for t, o, n in zip(cpu_tensors, offs, lens):
            buf[o:o + n].copy_(t.reshape(-1))
        dev = torch.empty(pos, dtype=torch.float32, device=device)
        dev.copy_(buf[:pos], non_blocking=True)
        return [dev[o:o + n] for o, n in zip(offs, lens)]

    def to_device_pcm16(self, pcm_tensors, device):
        """int16 CPU tensors (wavio.read_pcm16) -> f32 device views, samples * 2^-15: the 16-bit samples cross PCIe (half the
        bytes) and are converted by swc_pcm16_to_f32; what load_audio + to_device give for the same files, bit for bit."""
        from . import ops
        lens = [int(t.numel()) for t in pcm_tensors]
        offs, pos = [], 0
        for n in lens:
            offs.append(pos)
            pos += (n + 7) // 8 * 8            # 16-byte boundaries on both sides of the conversion
        if pos == 0:
            return [torch.empty(0, dtype=torch.float32, device=device) for _ in lens]
        buf = getattr(self._tls, "buf16", None)
        if buf is None or buf.numel() < pos:
            buf = torch.empty(max(pos, 1 << 20), dtype=torch.int16).pin_memory()
            self._tls.buf16 = buf
        for t, o, n in zip(pcm_tensors, offs, lens):
            buf[o:o + n].copy_(t.reshape(-1))
        dev16 = torch.empty(pos, dtype=torch.int16, device=device)
        dev16.copy_(buf[:pos], non_blocking=True)
        with torch.cuda.device(device):
            dev = ops.pcm16_to_f32(dev16)
        return [dev[o:o + n] for o, n in zip(offs, lens)]

    @staticmethod
    def _by_buffer(tensors):
        """tensors grouped by the buffer they are views of: [(base or None, [indices])].  decode() returns rows of ONE padded
        buffer, DataParallelCodec.encode_decode rows of one buffer per rank: a few groups, not one per utterance."""
        groups, where = [], {}
        for i, t in enumerate(tensors):
            b = t._base
            ok = b is not None and b.device.type == "cuda" and b.is_contiguous()
            k = id(b) if ok else ("single", i)
            if k not in where:
                where[k] = len(groups)
                groups.append((b if ok else None, []))
            groups[where[k]][1].append(i)
        return groups

    @staticmethod
    def pcm16_on_device(tensors):
        """f32 device tensors (decode()'s rows of a padded buffer) -> int16 device tensors round(clip(x, -1, 1) * 32767), again
        rows of one buffer per source buffer (swc_f32_to_pcm16 over the whole padded buffer; launched on the current stream,
        nothing is copied).  to_host() then moves half the bytes in one copy per buffer; the samples are those
        wavio.save_audio writes, bit for bit."""
        from . import ops
        out = [None] * len(tensors)
        for base, idx in HostStager._by_buffer(tensors):
            if base is None or base.dtype != torch.float32:
                for i in idx:
                    with torch.cuda.device(tensors[i].device):
                        out[i] = ops.f32_to_pcm16(tensors[i].contiguous())
                continue
            with torch.cuda.device(base.device):
                b16 = ops.f32_to_pcm16(base)
            for i in idx:
                t = tensors[i]
                out[i] = b16.as_strided(t.size(), t.stride(), t.storage_offset() - base.storage_offset())
        return out

    @staticmethod
    def to_host(tensors):
        """device tensors -> host tensors, one copy per source buffer (see _by_buffer) instead of one per utterance.  The copies
        run on the current stream: the producing stream must have been synchronised, or be this one."""
        out = [None] * len(tensors)
        for base, idx in HostStager._by_buffer(tensors):
            if base is None:
                for i in idx:
                    out[i] = tensors[i].cpu()
                continue
            host = base.cpu()
            for i in idx:
                t = tensors[i]
                out[i] = host.as_strided(t.size(), t.stride(), t.storage_offset() - base.storage_offset())
        return out
