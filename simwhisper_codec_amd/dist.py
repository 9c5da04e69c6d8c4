"""Utterance-level data parallelism over the GPUs of one node (BASELINE.json configs[3]).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI; "gloo" on CPU for tests).  The path has no
exchange step: utterances are independent, so rank 0 scatters the audio with point-to-point sends grouped into one
RCCL group call (its 7 xGMI links work concurrently, nothing is reduced, no ring), every rank runs the unchanged
single-GPU codec on its shard, and codes / waveforms are gathered the same way.  This replaces the reference's serial
batch loop (inference.py:38-64) for batches larger than one GPU's share.

  * Shapes travel as ONE fixed-size int64 tensor broadcast ([count, lengths ...], `_INTS_CAP` entries; the maximum code
    length follows from the lengths on every rank) — no pickling, no object collectives, one host read-back per step.
  * Failures are collective: after its local work every rank contributes one status int to a MIN all-reduce; if any rank
    failed (SwcError under policy "raise", out of memory, ...) every rank raises before the gathers are posted, instead
    of the healthy ranks blocking in a receive that will never be matched.
  * Per encode_decode() step the collectives are therefore: 1 broadcast (lengths), 1 all-reduce (status, 8 bytes),
    and the point-to-point scatter + two gathers (batch_isend_irecv groups).
  * Rank 0 assembles the padded [B, L] batch once (one gather kernel) and sends each rank its ROW SLICE of that
    buffer: no per-shard concatenation, no staging copy.  Receivers use row views of what arrives.
  * The only global quantity is the decode padding length: the reference's un-masked up-sampler / Vocos make a short
    utterance depend on the longest one of its batch (model.py:327-333), so shards pad to the global maximum code
    length and sharded results equal the single-GPU batch bit for bit.
  * encode_decode(): the gather of the codes (tiny) is posted before the local decode starts and completes under it.
"""
import contextlib

import torch
import torch.distributed as dist


def partition(weights, world):
    """Contiguous split of range(len(weights)) into `world` parts with balanced total weight.
    Returns a list of (start, end) per rank (possibly empty)."""
    n = len(weights)
    total = float(sum(weights))
    bounds, acc, start = [], 0.0, 0
    for r in range(world):
        if r == world - 1:
            end = n
        else:
            target = total * (r + 1) / world
            end = start
            while end < n and (acc + weights[end] <= target or end == start) and (n - end) > (world - 1 - r):
                acc += weights[end]
                end += 1
            if end == start and start < n and (n - start) > (world - 1 - r):
                acc += weights[end]
                end += 1
        bounds.append((start, end))
        start = end
    return bounds


class _Pending:
    """Outstanding point-to-point work + the buffers it fills."""

    def __init__(self, works, bufs):
        self.works, self.bufs = works, bufs

    def wait(self):
        for w in self.works:
            w.wait()
        return self.bufs


class DataParallelCodec:
    """Wraps a codec object (AudioCodec surface: encode/decode returning the reference's dicts)."""

    def __init__(self, codec, device, group=None, comm_device=None):
        """comm_device: where the point-to-point buffers live (default: `device`, i.e. RCCL sends straight from HBM
        over xGMI; "cpu" for a gloo group driving HIP codecs, as the single-card test does)."""
        self.codec, self.device, self.group = codec, torch.device(device), group
        if self.device.type == "cuda" and self.device.index is None:
            # an un-indexed "cuda" compares unequal to every tensor's device ("cuda:0"): the one-kernel batch assembly would
            # silently fall back to one copy per utterance (1.5 ms of rank 0's host time per step at 256 utterances)
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.comm = torch.device(comm_device) if comm_device is not None else self.device
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    # ------------------------------------------------------------------ small integers: one tensor broadcast
    _INTS_CAP = 1024  # entries of the fixed-size broadcast (8 KiB): count + up to 1023 lengths in ONE collective

    def _share_ints(self, values, error=None):
        """rank 0 passes a list of ints; every rank returns it.  One int64 broadcast of fixed size ([count, values ...]) and
        one host read-back; lists longer than the buffer take a second broadcast for the remainder.
        error (rank 0 only): rank 0 could not produce the list (an unreadable file, a failed batch assembly): the count
        travels as -1 and EVERY rank raises here, at once, instead of the others waiting in this broadcast for the
        process-group timeout."""
        if self.world == 1:
            if error is not None:
                raise error
            return [int(v) for v in values]
        cap = self._INTS_CAP
        # (RCCL: from the side stream — the lengths are host data on rank 0 and nothing on the compute stream produces them, so
        # reading them back must not wait for the previous step's kernels still queued there; the host then runs a step ahead
        # and back-pressure comes from the data dependencies of the sends alone)
        side = self._status_stream()
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            if self.rank == 0:
                vals = [int(v) for v in values] if error is None else []
                head = ([len(vals)] + vals[: cap - 1]) if error is None else [-1]
                t = torch.tensor(head + [0] * (cap - len(head)), dtype=torch.int64, device=self.comm)
            else:
                t = torch.empty(cap, dtype=torch.int64, device=self.comm)
            dist.broadcast(t, src=0, group=self.group)
            got = t.tolist()
            k = int(got[0])
            if k < 0:
                if error is not None:
                    raise error
                raise RuntimeError("DataParallelCodec: rank 0 could not assemble this batch (see its log); the step is abandoned "
                                   "on every rank")
            out = got[1: 1 + min(k, cap - 1)]
            if k > cap - 1:
                rest = torch.tensor(vals[cap - 1:], dtype=torch.int64, device=self.comm) if self.rank == 0 else \
                    torch.empty(k - (cap - 1), dtype=torch.int64, device=self.comm)
                dist.broadcast(rest, src=0, group=self.group)
                out += rest.tolist()
        return [int(v) for v in out]

    def _agree(self, error):
        """Collective failure handling: every rank passes the exception its local work raised (or None).  If any rank
        failed, EVERY rank raises here — the failing one its own exception, the others a RuntimeError naming the situation —
        before another collective or point-to-point call is posted."""
        if self.world > 1:
            # the flag is known on the HOST once the local work is enqueued; on RCCL the all-reduce runs from a side stream so
            # that reading it back waits for the peers, not for this rank's own decode kernels (round 4: the step no longer ends
            # in a drain of the compute stream; the gathers posted next are stream-ordered behind the decode as before)
            side = self._status_stream()
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                t = torch.tensor([0 if error is not None else 1], dtype=torch.int64, device=self.comm)
                dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
                ok = int(t.item()) == 1
        else:
            ok = error is None
        if error is not None:
            raise error
        if not ok:
            raise RuntimeError("DataParallelCodec: another rank failed in its local encode / decode; this step is abandoned "
                               "on every rank")

    def _status_stream(self):
        if self.comm.type != "cuda":
            return None
        st = self.__dict__.get("_side")
        if st is None:
            st = self.__dict__["_side"] = torch.cuda.Stream(device=self.comm)
        return st

    def _guarded(self, fn):
        """fn() = this rank's local work; returns its result after all ranks have agreed that nobody failed.  Rank 0's
        scatter sends are waited for HERE, after its own local work has been enqueued: its encode runs beside the sends
        instead of behind them (rank 0 is on the critical path of every step)."""
        err, out = None, None
        try:
            out = fn()
        except Exception as e:  # noqa: BLE001 - re-raised by _agree on this rank, reported on the others
            err = e
        try:
            self._finish_scatter()
        except Exception as e:  # noqa: BLE001
            err = err or e
        self._agree(err)
        return out

    def abort(self, error):
        """rank 0: the batch could not be loaded / assembled.  Takes the place of the step's first collective (the lengths
        broadcast) and raises `error` here and a RuntimeError on every other rank (their next encode / decode /
        encode_decode call returns by raising)."""
        self._share_ints(None, error=error)

    def _rank0(self, fn):
        """rank 0's host-side preparation of a step (lengths, batch assembly): a failure travels as the abort marker of the
        lengths broadcast instead of leaving the other ranks inside it."""
        try:
            return fn(), None
        except Exception as e:  # noqa: BLE001
            return None, e

    _sends = None

    def _finish_scatter(self):
        p, self._sends = self._sends, None
        if p is not None:
            p.wait()

    # ------------------------------------------------------------------ point-to-point scatter / gather of row slices
    def _scatter_rows(self, batch, parts, width, dtype):
        """rank 0: batch is the padded [B, width] tensor; rank r receives rows parts[r].  Returns this rank's [rows, width]."""
        a, b = parts[self.rank]
        if self.rank == 0:
            ops, keep = [], []
            for r in range(1, self.world):
                ra, rb = parts[r]
                if rb > ra and width > 0:
                    sl = batch[ra:rb]  # contiguous row slice of the gathered batch: sent as it is
                    if sl.device != self.comm:
                        sl = sl.to(self.comm)
                    keep.append(sl)
                    ops.append(dist.P2POp(dist.isend, sl, r, self.group))
            # not waited for here: _guarded() waits once this rank's own local work is enqueued (the slices stay referenced)
            self._finish_scatter()
            self._sends = _Pending(dist.batch_isend_irecv(ops), keep + [batch]) if ops else None
            return batch[a:b]
        buf = torch.empty((b - a, width), device=self.comm, dtype=dtype)
        if b > a and width > 0:
            for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, buf, 0, self.group)]):
                w.wait()
        return buf if buf.device == self.device else buf.to(self.device)

    def _gather_rows_async(self, mine, parts, width, dtype):
        """every rank contributes its [rows, width] tensor (contiguous); rank 0 gets a _Pending whose wait() returns the
        list of per-rank tensors on self.device; other ranks get a _Pending to wait on before `mine` may be freed."""
        if self.rank == 0:
            bufs, ops = [mine], []
            for r in range(1, self.world):
                ra, rb = parts[r]
                t = torch.empty((rb - ra, width), device=self.comm, dtype=dtype)
                bufs.append(t)
                if rb > ra and width > 0:
                    ops.append(dist.P2POp(dist.irecv, t, r, self.group))
            works = dist.batch_isend_irecv(ops) if ops else []
            return _Pending(works, bufs)
        a, b = parts[self.rank]
        works = []
        if b > a and width > 0:
            src = (mine if mine.device == self.comm else mine.to(self.comm)).contiguous()
            works = dist.batch_isend_irecv([dist.P2POp(dist.isend, src, 0, self.group)])
            return _Pending(works, [src])  # the very tensor being sent stays referenced until wait()
        return _Pending(works, [])

    # ------------------------------------------------------------------ batch assembly helpers
    def _pad_batch(self, tensors, lens, dtype, min_len=0):
        """list of 1-D tensors -> zero-padded [B, max(len, min_len)] on self.device (the codec's own one-kernel gather
        when it has one)."""
        if not tensors:
            return torch.zeros((0, max(int(min_len), 1)), device=self.device, dtype=dtype)
        stack = getattr(self.codec, "_stack", None)
        if stack is not None and self.device.type == "cuda":
            with torch.cuda.device(self.device):
                # (a reshape makes a new tensor object per utterance: 1.3 us each, a third of rank 0's per-step host time
                # at 256 utterances; audio rows are 1-D already)
                return stack([t if t.dim() == 1 else t.reshape(-1) for t in tensors], lens, self.device, dtype, min_len)
        L = max(max(lens), 1, int(min_len))
        out = torch.zeros(len(tensors), L, device=self.device, dtype=dtype)
        for i, t in enumerate(tensors):
            if lens[i]:
                out[i, : lens[i]] = t.reshape(-1).to(self.device, dtype)
        return out

    def _encode_rows(self, mine, lens, a, b, overlap_seconds):
        """this rank's encode of its received rows.  Returns the codes as a padded (G, rows, Lc') tensor (codecs with the
        batch entry points encode_padded / decode_padded: no per-utterance tensors at all) or as a list of (G, T_i) tensors."""
        if b == a:
            return []
        if hasattr(self.codec, "encode_padded"):
            return self.codec.encode_padded(mine, lens[a:b], overlap_seconds=overlap_seconds)  # tensor or None
        local = [mine[i - a, : lens[i]] for i in range(a, b)]
        return self.codec.encode(local, overlap_seconds=overlap_seconds, device=self.device)["codes_list"]

    def _codes_rows(self, codes, clen, a, b, Lc):
        """local codes (padded tensor, None or list) -> padded int32 [(b-a)*G, Lc] rows (utterance-major, group-minor)."""
        G = self.codec.num_groups
        if codes is None or torch.is_tensor(codes):
            rows = torch.zeros(((b - a) * G, Lc), device=self.device, dtype=torch.int32)
            if codes is not None:
                wdt = min(Lc, codes.shape[2])
                rows.view(b - a, G, Lc)[:, :, :wdt] = codes[:, :, :wdt].permute(1, 0, 2)
            return rows
        rows = [c[g] if c.dtype == torch.int32 else c[g].to(torch.int32) for c in codes for g in range(G)]
        return self._pad_batch(rows, [clen[a + k] for k in range(b - a) for _ in range(G)], torch.int32, Lc)

    def _local_decode(self, codes, clen, a, b, t_max, overlap_seconds, Lw):
        """this rank's decode of `codes` (padded (G, rows, L) tensor, None or list) -> f32 rows [(b-a), Lw] for the gather."""
        if b == a:
            return torch.zeros((0, Lw), device=self.device, dtype=torch.float32)
        if hasattr(self.codec, "decode_padded") and (codes is None or torch.is_tensor(codes)):
            if codes is None:
                codes = torch.zeros((self.codec.num_groups, b - a, 1), device=self.device, dtype=torch.int32)
            # encode_padded returns whole windows (zero codes beyond each length): the batch the reference would decode
            # is padded to the longest utterance of the WHOLE batch and no further (its results depend on that length)
            codes = codes[:, :, : max(t_max, 1)]
            wav = self.codec.decode_padded(codes, clen[a:b], overlap_seconds=overlap_seconds, pad_to_length=t_max)
            if wav.shape[1] == Lw and wav.is_contiguous():
                return wav
            rows = torch.zeros((b - a, Lw), device=self.device, dtype=torch.float32)
            rows[:, : min(Lw, wav.shape[1])] = wav[:, :Lw]
            return rows
        wavs = self.codec.decode(list(codes), overlap_seconds=overlap_seconds, device=self.device,
                                 pad_to_length=t_max)["syn_wav_list"] if codes else []
        return self._pad_batch(list(wavs), [int(w.numel()) for w in wavs], torch.float32, Lw)

    # per-utterance views of the gathered buffers.  One unbind per shard instead of one Python slice per utterance (a view
    # costs microseconds of host time; at 8 x 32 utterances the per-row slices of both lists were ~2 ms of every step on
    # rank 0, with the GPUs idle); only rows shorter than the buffer are narrowed.
    def _split_codes(self, got, parts, clen):
        G = self.codec.num_groups
        out = []
        for (a, b), buf in zip(parts, got):
            if b == a:
                continue
            buf = buf if buf.device == self.device else buf.to(self.device)
            Lc = buf.shape[1]
            rows = buf.view(b - a, G, Lc).unbind(0)
            out.extend(r if clen[a + k] == Lc else r[:, : clen[a + k]] for k, r in enumerate(rows))
        return out

    def _split_wavs(self, got, parts, clen):
        up = self.codec.decoder_upsample_rate
        out = []
        for (a, b), buf in zip(parts, got):
            if b == a:
                continue
            buf = buf if buf.device == self.device else buf.to(self.device)
            Lw = buf.shape[1]
            rows = buf.unbind(0)
            out.extend(r if up * clen[a + k] == Lw else r[: up * clen[a + k]] for k, r in enumerate(rows))
        return out

    def _share_audio(self, wav_list):
        """(lengths on every rank, the padded [B, L] batch on rank 0).  Rank 0 assembles the batch BEFORE the lengths travel:
        if that fails (a tensor on the wrong device, out of memory) the broadcast carries the abort marker and every rank
        raises, instead of the peers waiting for rows that are never sent."""
        def prep():
            ln = [int(w.shape[-1]) for w in wav_list]
            return ln, (self._pad_batch(wav_list, ln, torch.float32) if ln else None)
        (got, err) = self._rank0(prep) if self.rank == 0 else (None, None)
        lens = self._share_ints(got[0] if got else None, error=err)
        return lens, (got[1] if got else None)

    # ------------------------------------------------------------------ public surface
    def encode(self, wav_list=None, overlap_seconds=10):
        """rank 0 passes the full list; returns {"codes_list": [...]} on rank 0, None elsewhere."""
        lens, batch = self._share_audio(wav_list)
        if not lens:
            return {"codes_list": []} if self.rank == 0 else None
        parts = partition(lens, self.world)
        L = max(lens)
        mine = self._scatter_rows(batch, parts, max(L, 1), torch.float32)
        a, b = parts[self.rank]
        codes = self._guarded(lambda: self._encode_rows(mine, lens, a, b, overlap_seconds))
        rate = self.codec.encoder_downsample_rate
        clen = [l // rate for l in lens]
        Lc = max(max(clen), 1)
        a, b = parts[self.rank]
        G = self.codec.num_groups
        cparts = [(pa * G, pb * G) for pa, pb in parts]
        got = self._gather_rows_async(self._codes_rows(codes, clen, a, b, Lc), cparts, Lc, torch.int32).wait()
        if self.rank != 0:
            return None
        return {"codes_list": self._split_codes(got, parts, clen)}

    def decode(self, codes_list=None, overlap_seconds=10):
        """rank 0 passes the full list; returns {"syn_wav_list": [...]} on rank 0, None elsewhere."""
        G = self.codec.num_groups
        def prep():  # rows = (utterance, group): one padded int32 matrix, shards are row slices of it
            cl = [int(c.shape[-1]) for c in codes_list]
            if not cl:
                return cl, None
            rows = [c[g].to(torch.int32) for c in codes_list for g in range(G)]
            return cl, self._pad_batch(rows, [n for n in cl for _ in range(G)], torch.int32, max(max(cl), 1))
        (got0, err) = self._rank0(prep) if self.rank == 0 else (None, None)
        clen = self._share_ints(got0[0] if got0 else None, error=err)
        batch = got0[1] if got0 else None
        if not clen:
            return {"syn_wav_list": []} if self.rank == 0 else None
        t_max = max(clen)
        parts = partition([max(c, 1) for c in clen], self.world)
        Lc = max(t_max, 1)
        cparts = [(pa * G, pb * G) for pa, pb in parts]
        mine = self._scatter_rows(batch, cparts, Lc, torch.int32)
        a, b = parts[self.rank]
        up = self.codec.decoder_upsample_rate
        Lw = max(up * t_max, 1)
        if hasattr(self.codec, "decode_padded"):
            local = mine.view(b - a, G, Lc).permute(1, 0, 2) if b > a else None  # (G, rows, Lc) view of the received rows
        else:
            local = [mine[(i - a) * G:(i - a + 1) * G, : clen[i]] for i in range(a, b)]
        rows = self._guarded(lambda: self._local_decode(local, clen, a, b, t_max, overlap_seconds, Lw))
        got = self._gather_rows_async(rows, parts, Lw, torch.float32).wait()
        if self.rank != 0:
            return None
        return {"syn_wav_list": self._split_wavs(got, parts, clen)}

    def encode_decode(self, wav_list=None, overlap_seconds=10):
        """Round trip of one batch (the configs[3] serving step): scatter audio -> encode -> (codes gathered while) decode ->
        gather waveforms.  The codes never leave their GPU between encode and decode; the one global integer (maximum
        code length, for the reference's T_max rule) follows from the broadcast lengths.  Rank 0 returns
        {"codes_list", "syn_wav_list"} equal to codec.decode(codec.encode(all)); other ranks return None."""
        lens, batch = self._share_audio(wav_list)
        if not lens:
            return {"codes_list": [], "syn_wav_list": []} if self.rank == 0 else None
        parts = partition(lens, self.world)
        rate, up, G = self.codec.encoder_downsample_rate, self.codec.decoder_upsample_rate, self.codec.num_groups
        clen = [l // rate for l in lens]
        t_max = max(clen)
        Lc, Lw = max(t_max, 1), max(up * t_max, 1)
        a, b = parts[self.rank]
        cparts = [(pa * G, pb * G) for pa, pb in parts]
        L = max(lens)
        mine = self._scatter_rows(batch, parts, max(L, 1), torch.float32)

        def local_round_trip():
            codes = self._encode_rows(mine, lens, a, b, overlap_seconds)
            return codes, self._local_decode(codes, clen, a, b, t_max, overlap_seconds, Lw)
        # the operand-range check of the split-f16 encoder is read back ONCE, after the decode has been enqueued (no stall
        # between encode and decode); if operands clipped the codec has switched to exact-f32 operands and the shard is redone
        defer = getattr(self.codec, "deferred_range_check", None)

        def guarded_round_trip():
            if defer is None:
                return local_round_trip()
            with defer() as chk:
                out = local_round_trip()
            return local_round_trip() if chk.clipped else out
        codes, rows = self._guarded(guarded_round_trip)
        pend_codes = self._gather_rows_async(self._codes_rows(codes, clen, a, b, Lc), cparts, Lc, torch.int32)
        pend_wavs = self._gather_rows_async(rows, parts, Lw, torch.float32)
        got_c = pend_codes.wait()
        got_w = pend_wavs.wait()
        if self.rank != 0:
            return None
        return {"codes_list": self._split_codes(got_c, parts, clen), "syn_wav_list": self._split_wavs(got_w, parts, clen)}
