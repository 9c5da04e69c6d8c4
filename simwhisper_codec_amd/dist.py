"""Utterance-level data parallelism over the GPUs of one node.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI; "gloo" on CPU
for tests).  The path has no exchange step: utterances are independent, so rank 0 fans the
audio out with point-to-point sends (its 7 xGMI links work concurrently, nothing is
reduced), every rank runs the unchanged single-GPU codec on its shard, and codes /
waveforms come back the same way.  The only global quantity is the decode padding length:
the reference's un-masked up-sampler / Vocos make a short utterance depend on the longest one
in its batch, so the global maximum code length is shared (one integer) and every shard pads
to it — sharded results are then identical to the single-GPU batch.
"""
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


class DataParallelCodec:
    """Wraps a codec object (AudioCodec surface: encode/decode returning the reference's dicts)."""

    def __init__(self, codec, device, group=None, comm_device=None):
        """comm_device: where the point-to-point buffers live (default: `device`, i.e. RCCL sends straight from HBM
        over xGMI; "cpu" for a gloo group driving HIP codecs, as the single-card test does)."""
        self.codec, self.device, self.group = codec, torch.device(device), group
        self.comm = torch.device(comm_device) if comm_device is not None else self.device
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    # ---- point-to-point fan-out / fan-in of flat buffers
    def _fan_out(self, flats, sizes, dtype):
        """rank 0: flats[r] is the 1-D tensor for rank r.  Returns this rank's tensor."""
        if self.rank == 0:
            ops = [dist.P2POp(dist.isend, flats[r].to(self.comm), r, self.group) for r in range(1, self.world) if sizes[r] > 0]
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            return flats[0]
        buf = torch.empty(sizes[self.rank], device=self.comm, dtype=dtype)
        if sizes[self.rank] > 0:
            for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, buf, 0, self.group)]):
                w.wait()
        return buf.to(self.device)

    def _fan_in(self, flat, sizes, dtype):
        """every rank contributes a 1-D tensor; rank 0 returns the list of all of them."""
        if self.rank == 0:
            bufs = [flat] + [torch.empty(sizes[r], device=self.comm, dtype=dtype) for r in range(1, self.world)]
            ops = [dist.P2POp(dist.irecv, bufs[r], r, self.group) for r in range(1, self.world) if sizes[r] > 0]
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            return [bufs[0]] + [b.to(self.device) for b in bufs[1:]]
        if sizes[self.rank] > 0:
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, flat.to(self.comm), 0, self.group)]):
                w.wait()
        return None

    def _share(self, obj):
        box = [obj]
        dist.broadcast_object_list(box, src=0, group=self.group)
        return box[0]

    def encode(self, wav_list=None, overlap_seconds=10):
        """rank 0 passes the full list; returns {"codes_list": [...]} on rank 0, None elsewhere."""
        lens = self._share([int(w.shape[-1]) for w in wav_list] if self.rank == 0 else None)
        parts = partition(lens, self.world)
        sizes = [sum(lens[a:b]) for a, b in parts]
        flats = None
        if self.rank == 0:
            flats = [torch.cat([w.reshape(-1).to(self.device, torch.float32) for w in wav_list[a:b]]) if b > a
                     else torch.empty(0, device=self.device) for a, b in parts]
        mine = self._fan_out(flats, sizes, torch.float32)
        a, b = parts[self.rank]
        local = list(torch.split(mine, lens[a:b])) if b > a else []
        codes = self.codec.encode(local, overlap_seconds=overlap_seconds, device=self.device)["codes_list"] if local else []
        G = self.codec.num_groups
        rate = self.codec.encoder_downsample_rate
        clen = [l // rate for l in lens]
        csz = [G * sum(clen[a:b]) for a, b in parts]
        flat = torch.cat([c.to(torch.int32).reshape(-1) for c in codes]) if codes else torch.empty(0, device=self.device, dtype=torch.int32)
        got = self._fan_in(flat, csz, torch.int32)
        if self.rank != 0:
            return None
        out = []
        for (a, b), buf in zip(parts, got):
            off = 0
            for i in range(a, b):
                out.append(buf[off:off + G * clen[i]].view(G, clen[i]))
                off += G * clen[i]
        return {"codes_list": out}

    def decode(self, codes_list=None, overlap_seconds=10):
        """rank 0 passes the full list; returns {"syn_wav_list": [...]} on rank 0, None elsewhere."""
        G = self.codec.num_groups
        clen = self._share([int(c.shape[-1]) for c in codes_list] if self.rank == 0 else None)
        t_max = max(clen) if clen else 0
        parts = partition([max(c, 1) for c in clen], self.world)
        sizes = [G * sum(clen[a:b]) for a, b in parts]
        flats = None
        if self.rank == 0:
            flats = [torch.cat([c.to(self.device, torch.int32).reshape(-1) for c in codes_list[a:b]]) if b > a
                     else torch.empty(0, device=self.device, dtype=torch.int32) for a, b in parts]
        mine = self._fan_out(flats, sizes, torch.int32)
        a, b = parts[self.rank]
        local, off = [], 0
        for i in range(a, b):
            local.append(mine[off:off + G * clen[i]].view(G, clen[i]).long())
            off += G * clen[i]
        wavs = self.codec.decode(local, overlap_seconds=overlap_seconds, device=self.device,
                                 pad_to_length=t_max)["syn_wav_list"] if local else []
        up = self.codec.decoder_upsample_rate
        wsz = [up * sum(clen[a:b]) for a, b in parts]
        flat = torch.cat([w.reshape(-1).to(torch.float32) for w in wavs]) if wavs else torch.empty(0, device=self.device)
        got = self._fan_in(flat, wsz, torch.float32)
        if self.rank != 0:
            return None
        out = []
        for (a, b), buf in zip(parts, got):
            off = 0
            for i in range(a, b):
                out.append(buf[off:off + up * clen[i]])
                off += up * clen[i]
        return {"syn_wav_list": out}
