"""world_size-2 (and 3) `gloo` tests of the data-parallel wrapper: sharded == un-sharded, incl. the
T_max rule of decode.  The codec here is a small CPU stand-in with the reference's T_max dependence
(the HIP codec needs a GPU); the real codec is covered by tests/test_parity_gpu.py."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from simwhisper_codec_amd.dist import DataParallelCodec, partition


class FakeCodec:
    """Deterministic per-utterance 'codec' whose decode depends on the padded batch length."""
    num_groups = 8
    encoder_downsample_rate = 1280
    decoder_upsample_rate = 1280

    def encode(self, wav_list, overlap_seconds=10, device=None):
        out = []
        for w in wav_list:
            T = w.shape[-1] // 1280
            fr = w[: T * 1280].view(T, 1280)
            base = (fr.abs().sum(1) * 1000).long() % 2016
            out.append(torch.stack([(base + g) % 2016 for g in range(8)]).to(torch.int32))
        return {"codes_list": out}

    def decode(self, codes_list, overlap_seconds=10, device=None, pad_to_length=None):
        L = max(max(c.shape[-1] for c in codes_list), int(pad_to_length or 0))
        out = []
        for c in codes_list:
            v = c.float().mean(0)  # (T,)
            out.append((v[:, None] * torch.linspace(0, 1, 1280)[None, :] + 0.001 * L).reshape(-1))
        return {"syn_wav_list": out}


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, lens, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        codec = FakeCodec()
        dp = DataParallelCodec(codec, "cpu")
        g = torch.Generator().manual_seed(0)
        wavs = [torch.randn(n, generator=g) for n in lens] if rank == 0 else None
        enc = dp.encode(wavs)
        dec = dp.decode(enc["codes_list"] if rank == 0 else None)
        rt = dp.encode_decode(wavs)  # the fused round trip: codes stay on their rank between encode and decode
        if rank == 0:
            want_c = codec.encode(wavs)["codes_list"]
            want_w = codec.decode(want_c)["syn_wav_list"]
            ok = all(torch.equal(a, b) for a, b in zip(enc["codes_list"], want_c)) and len(enc["codes_list"]) == len(lens)
            ok = ok and all(torch.equal(a, b) for a, b in zip(dec["syn_wav_list"], want_w))
            ok = ok and len(rt["codes_list"]) == len(lens) and len(rt["syn_wav_list"]) == len(lens)
            ok = ok and all(torch.equal(a, b) for a, b in zip(rt["codes_list"], want_c))
            ok = ok and all(torch.equal(a, b) for a, b in zip(rt["syn_wav_list"], want_w))
            ret.put(bool(ok))
        else:
            assert enc is None and dec is None and rt is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,lens", [(2, [5000, 12800, 3000, 40000, 1280]), (3, [2560, 2560]), (2, [1000]),
                                        # the 8-GPU node's layout (BASELINE.json configs[3]: 256 utterances, 32 per rank), short rows
                                        (8, [2560 + 1280 * (i % 3) for i in range(256)])])
def test_sharded_equals_unsharded(world, lens):
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, lens, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(timeout=5) is True


def test_partition_properties():
    for w, n in [([1] * 10, 3), ([5, 1, 1, 1, 9, 2], 4), ([3], 4), ([], 2), ([1, 1], 8)]:
        parts = partition(w, n)
        assert len(parts) == n and parts[0][0] == 0 and parts[-1][1] == len(w)
        assert all(parts[i][1] == parts[i + 1][0] for i in range(n - 1))
        assert all(a <= b for a, b in parts)
    parts = partition([1] * 256, 8)
    assert [b - a for a, b in parts] == [32] * 8


class _FailingCodec(FakeCodec):
    """raises on one rank only: the others must not be left inside a receive that is never matched"""

    def __init__(self, bad_rank):
        self.bad_rank = bad_rank

    def encode(self, wav_list, overlap_seconds=10, device=None):
        if dist.get_rank() == self.bad_rank:
            raise ValueError("this rank's shard failed")
        return super().encode(wav_list, overlap_seconds, device)


def _worker_fail(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        dp = DataParallelCodec(_FailingCodec(bad_rank=1), "cpu")
        g = torch.Generator().manual_seed(0)
        wavs = [torch.randn(n, generator=g) for n in [5000, 12800, 3000, 6000, 7000, 2560]] if rank == 0 else None
        seen = []
        for call in (dp.encode, dp.encode_decode):
            try:
                call(wavs)
                seen.append("no error")
            except ValueError as e:
                seen.append(f"own:{e}")
            except RuntimeError as e:
                seen.append("other" if "another rank failed" in str(e) else f"unexpected:{e}")
        # the group is still usable: every rank raised at the same point, no collective was left half done
        ok = dp.decode([torch.zeros(8, 3, dtype=torch.int32)] * 4 if rank == 0 else None)
        ret.put((rank, seen, (ok is not None) == (rank == 0)))
    finally:
        dist.destroy_process_group()


def test_a_failing_rank_fails_every_rank_together():
    """a rank that raises in its local encode must not strand the others in point-to-point waits: after the local
    work a status all-reduce makes every rank raise before the gathers are posted (dist.py `_agree`)."""
    world = 3
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_fail, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(ret.get(timeout=5) for _ in range(world))
    for rank, seen, usable in got:
        want = "own:this rank's shard failed" if rank == 1 else "other"
        assert seen == [want, want], (rank, seen)
        assert usable


def test_share_ints_is_one_broadcast(monkeypatch):
    """lengths travel in ONE fixed-size broadcast (count + values): one collective, one host read-back per step."""
    class _DP(DataParallelCodec):
        def __init__(self):
            self.world, self.rank, self.group, self.comm = 2, 0, None, torch.device("cpu")
    calls = []
    monkeypatch.setattr(dist, "broadcast", lambda t, src=0, group=None: calls.append(t.numel()))
    dp = _DP()
    assert dp._share_ints(list(range(7, 300))) == list(range(7, 300))
    assert calls == [DataParallelCodec._INTS_CAP]
    calls.clear()
    big = list(range(3000))
    assert dp._share_ints(big) == big and len(calls) == 2   # beyond the buffer: one more for the remainder
    calls.clear()
    assert dp._share_ints([]) == [] and len(calls) == 1


def _worker_rank0_fails(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    import datetime
    import time
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        dp = DataParallelCodec(FakeCodec(), "cpu")
        seen = []
        t0 = time.perf_counter()
        # (1) the file loop's case: rank 0 cannot load a file of the batch -> dp.abort(error) in place of the step
        try:
            if rank == 0:
                dp.abort(OSError("input_wavs/broken.flac: bad frame CRC"))
            else:
                dp.encode_decode(None)
            seen.append("no error")
        except OSError as e:
            seen.append(f"own:{e}")
        except RuntimeError as e:
            seen.append("told" if "rank 0 could not assemble" in str(e) else f"unexpected:{e}")
        # (2) rank 0's batch assembly itself fails (an entry that is not a tensor): inside encode(), before the broadcast
        for call in (dp.encode, dp.encode_decode):
            try:
                call([torch.zeros(2560), "not a tensor"] if rank == 0 else None)
                seen.append("no error")
            except (AttributeError, TypeError):
                seen.append("own")
            except RuntimeError as e:
                seen.append("told" if "rank 0 could not assemble" in str(e) else f"unexpected:{e}")
        # (3) and decode(): a code tensor of the wrong rank
        try:
            dp.decode([torch.zeros(8, 3, dtype=torch.int32), None] if rank == 0 else None)
            seen.append("no error")
        except (AttributeError, TypeError):
            seen.append("own")
        except RuntimeError as e:
            seen.append("told" if "rank 0 could not assemble" in str(e) else f"unexpected:{e}")
        took = time.perf_counter() - t0
        # the group is still usable: every rank raised at the same collective
        g = torch.Generator().manual_seed(0)
        ok = dp.encode_decode([torch.randn(n, generator=g) for n in (2560, 5120, 3000)] if rank == 0 else None)
        ret.put((rank, seen, (ok is not None) == (rank == 0), took))
    finally:
        dist.destroy_process_group()


def test_rank0_load_failure_reaches_every_rank_at_once():
    """a corrupt / unreadable file fails on rank 0 OUTSIDE every collective (inference.py: nxt.result(), stage_in;
    dist.py: _pad_batch).  The other ranks sat in the lengths broadcast until the 240 s process-group timeout; now the
    broadcast carries an abort marker (count -1) and every rank raises promptly."""
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_rank0_fails, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(ret.get(timeout=5) for _ in range(world))
    for rank, seen, usable, took in got:
        if rank == 0:
            assert seen == ["own:input_wavs/broken.flac: bad frame CRC", "own", "own", "own"], seen
        else:
            assert seen == ["told"] * 4, seen
        assert usable and took < 30.0
