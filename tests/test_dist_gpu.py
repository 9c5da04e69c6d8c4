"""Two ranks sharing the one GPU of the test box (gloo for the point-to-point traffic, HIP for the work): the
sharded DataParallelCodec must return exactly what the single-process batch returns — codes and waveforms bit for
bit, including the T_max rule of decode.  On a node the same wrapper runs one rank per GPU over RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, ret):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from common import PARAMS, state_dict
    from simwhisper_codec_amd import synth
    from simwhisper_codec_amd.codec import AudioCodec
    from simwhisper_codec_amd.dist import DataParallelCodec
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = AudioCodec(PARAMS["tiny"](), precision="mixed")
        m.load_state_dict(state_dict("tiny"), strict=True)
        m = m.to("cuda").eval()
        dp = DataParallelCodec(m, "cuda", comm_device="cpu")
        lens = [16000 * 3 + 17, 16000 * 7, 5000, 16000 * 2, 16000 * 5 + 999]
        wavs = [synth.synth_audio(n, index=300 + i, kind="speech" if i % 2 else "noise").cuda() for i, n in enumerate(lens)] \
            if rank == 0 else None
        enc = dp.encode(wavs)
        dec = dp.decode(enc["codes_list"] if rank == 0 else None)
        rt = dp.encode_decode(wavs)
        if rank == 0:
            want_c = m.encode(wavs)["codes_list"]
            want_w = m.decode(want_c)["syn_wav_list"]
            ok = len(enc["codes_list"]) == len(lens)
            ok = ok and all(torch.equal(a.long().cpu(), b.long().cpu()) for a, b in zip(enc["codes_list"], want_c))
            ok = ok and all(torch.equal(a.cpu(), b.cpu()) for a, b in zip(dec["syn_wav_list"], want_w))
            ok = ok and all(torch.equal(a.long().cpu(), b.long().cpu()) for a, b in zip(rt["codes_list"], want_c))
            ok = ok and all(torch.equal(a.cpu(), b.cpu()) for a, b in zip(rt["syn_wav_list"], want_w))
            ret.put(bool(ok))
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu_equal_single_process():
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert ret.get(timeout=5) is True


def _worker_nccl(port, ret):
    """world size 1 on the RCCL backend: init, tensor broadcasts, the scatter / gather code path with device buffers."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from common import PARAMS, state_dict
    from simwhisper_codec_amd import synth
    from simwhisper_codec_amd.codec import AudioCodec
    from simwhisper_codec_amd.dist import DataParallelCodec
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        m = AudioCodec(PARAMS["tiny"](), precision="mixed")
        m.load_state_dict(state_dict("tiny"), strict=True)
        m = m.to("cuda:0").eval()
        dp = DataParallelCodec(m, "cuda:0")
        lens = [16000 * 3 + 17, 16000 * 2, 5000, 0, 16000 * 4 + 999]
        wavs = [synth.synth_audio(n, index=310 + i, kind="speech" if i % 2 else "noise").cuda() for i, n in enumerate(lens)]
        enc = dp.encode(wavs)
        dec = dp.decode(enc["codes_list"])
        rt = dp.encode_decode(wavs)
        t = torch.ones(1, device="cuda:0")
        dist.all_reduce(t)  # the collective bench.py uses for the MAX over ranks
        want_c = m.encode(wavs)["codes_list"]
        want_w = m.decode(want_c)["syn_wav_list"]
        ok = all(torch.equal(a.long(), b.long()) for a, b in zip(enc["codes_list"], want_c))
        ok = ok and all(torch.equal(a, b) for a, b in zip(dec["syn_wav_list"], want_w))
        ok = ok and all(torch.equal(a.long(), b.long()) for a, b in zip(rt["codes_list"], want_c))
        ok = ok and all(torch.equal(a, b) for a, b in zip(rt["syn_wav_list"], want_w))
        ret.put(bool(ok))
    finally:
        dist.destroy_process_group()


def test_single_rank_rccl_equals_plain_codec():
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    p = ctx.Process(target=_worker_nccl, args=(_free_port(), ret))
    p.start()
    p.join(300)
    assert p.exitcode == 0
    assert ret.get(timeout=5) is True


def _worker_cli(rank, world, port, ind, outd, cfg):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    import inference
    inference.main(["--config_path", cfg, "--synthetic_checkpoint", "--device", "cuda", "--batch_size", "2", "--input_dir", ind,
                    "--output_dir", outd, "--precision", "mixed", "--dist_backend", "gloo"])


def test_cli_two_ranks_writes_the_single_gpu_files(tmp_path):
    """inference.py under two ranks (one GPU shared, gloo for the point-to-point traffic): rank 0 owns the files, both ranks
    compute; the WAV files equal those of the single-process CLI run byte for byte."""
    import sys, yaml
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from common import PARAMS
    import inference
    from simwhisper_codec_amd import synth
    from simwhisper_codec_amd.wavio import save_audio
    cfg = tmp_path / "tiny.yaml"
    cfg.write_text(yaml.safe_dump({"generator_params": PARAMS["tiny"]()}))
    ind, out1, out2 = tmp_path / "in", tmp_path / "out1", tmp_path / "out2"
    ind.mkdir()
    for i, n in enumerate([16000, 5000, 40000, 23456, 9999]):
        save_audio(str(ind / f"f{i}.wav"), synth.synth_audio(n, index=85 + i, kind="speech").reshape(1, -1), 16000)
    # single process, batches of 4 = the 2 x 2 files a two-rank step takes
    inference.main(["--config_path", str(cfg), "--synthetic_checkpoint", "--device", "cuda", "--batch_size", "4",
                    "--input_dir", str(ind), "--output_dir", str(out1), "--precision", "mixed"])
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker_cli, args=(r, 2, port, str(ind), str(out2), str(cfg))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    names = sorted(os.listdir(out1))
    assert names == sorted(os.listdir(out2)) and len(names) == 5
    for n in names:
        assert (out1 / n).read_bytes() == (out2 / n).read_bytes(), n


@pytest.mark.parametrize("n", [2, 3])
def test_bench_multi_rank_rehearsal(n):
    """bench.py --gpus N as the driver launches it (torch.distributed.run, one process per rank), rehearsed on one card:
    SWC_BENCH_BACKEND=gloo carries the traffic through host memory because RCCL refuses two ranks on one GPU.  Checks the
    multi-rank control flow of the bench: one JSON line, from rank 0, covering all ranks' utterances, answer check passed.
    (3 ranks: the box allows at most 6 processes on its GPU at once, the test runner and the launcher included; the 8-way layout runs in
    tests/test_dist_cpu.py.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SWC_BENCH_BACKEND="gloo")
    env.pop("SWC_LIB", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "2",
                        "--warmup", "1", "--batch", "4", "--seconds", "3", "--cpu-baseline", "off", "--no-inflight"],
                       cwd=root, env=dict(env, OMP_NUM_THREADS="3"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["steps"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert f"{4 * n} utterances in all" in d["config"]["workload"] and d["independent_shards"]["value"] > 0
    assert d["roofline"]["frac"] > 0
    assert d["parity"]["code_mismatches"] == 0 and "scatter_gather_error" not in d


def _worker_bookkeeping(port, ret):
    """rank 0 of an 8-rank layout, host side only: everything DataParallelCodec.encode_decode does on the host for 256
    utterances except the collectives themselves and the codec calls."""
    import sys, time
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from common import PARAMS, state_dict
    from simwhisper_codec_amd.codec import AudioCodec
    from simwhisper_codec_amd.dist import DataParallelCodec, partition
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        m = AudioCodec(PARAMS["tiny"](), precision="mixed")
        m.load_state_dict(state_dict("tiny"), strict=True)
        m = m.to("cuda").eval()
        dp = DataParallelCodec(m, "cuda")
        dp.world = 8                                   # the helpers below only index per-rank lists with it
        n, B, G = 160000, 256, m.num_groups
        wavs = [torch.zeros(n, device="cuda") for _ in range(B)]
        times, seg = [], {}
        got_c = [torch.zeros((32 * G, 125), dtype=torch.int32, device="cuda") for _ in range(8)]  # what the gathers deliver
        got_w = [torch.zeros((32, n), device="cuda") for _ in range(8)]
        batch = None
        for it in range(25):
            del batch                                          # (as in a serving loop: last step's batch has been dropped)
            torch.cuda.synchronize()
            t0 = t = time.perf_counter()

            def lap(name):
                nonlocal t
                now = time.perf_counter()
                seg.setdefault(name, []).append(now - t)
                t = now
            lens = [w.size(-1) for w in wavs]
            parts = partition(lens, dp.world)
            rate, up = m.encoder_downsample_rate, m.decoder_upsample_rate
            clen = [l // rate for l in lens]
            t_max = max(clen)
            cparts = [(pa * G, pb * G) for pa, pb in parts]
            lap("lengths+partition")
            batch = dp._pad_batch(wavs, lens, torch.float32)      # one gather kernel from an uploaded address list
            shards = [batch[a:b] for a, b in parts]               # what _scatter_rows sends
            lap("batch assembly")
            codes = dp._split_codes(got_c, parts, clen)
            lap("code views")
            out = {"codes_list": codes, "syn_wav_list": dp._split_wavs(got_w, parts, clen)}
            lap("waveform views")
            times.append(time.perf_counter() - t0)
            assert len(out["codes_list"]) == B and len(out["syn_wav_list"]) == B and len(shards) == 8 and t_max == 125
            assert [b - a for a, b in parts] == [32] * 8 and cparts[1] == (32 * G, 64 * G)
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/parity_report.txt", "a") as f:
            f.write("dist/rank0_bookkeeping_256 segments_us " + " ".join(
                f"{k.replace(' ', '_')}={1e6 * sorted(v)[len(v) // 2]:.0f}" for k, v in seg.items()) + "\n")
        times.sort()
        ret.put(times[len(times) // 2])
    finally:
        dist.destroy_process_group()


def test_rank0_bookkeeping_of_256_utterances_stays_small():
    """BASELINE.json configs[3] at the real shape (8 ranks x 32 utterances x 10 s): the host work rank 0 does per step
    around the collectives — lengths, partition, batch assembly (one gather kernel), per-utterance views of the gathered
    codes and waveforms — is measured (0.4 ms on the round-3 boxes; it was 1.9 ms with one copy per utterance) and written
    to the report; the assertion is a generous 5 ms, a wall-clock bound on a shared box that only a structural regression
    (a per-utterance copy loop, a stream sync per row) crosses — ADVICE r3: a 1 ms gate flakes under host load."""
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    p = ctx.Process(target=_worker_bookkeeping, args=(_free_port(), ret))
    p.start()
    p.join(300)
    assert p.exitcode == 0
    med = ret.get(timeout=5)
    out = os.path.join(os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_report.txt"), "a") as f:
        f.write(f"dist/rank0_bookkeeping_256 median_ms={1e3 * med:.3f}\n")
    assert med < 5e-3, f"rank-0 bookkeeping takes {1e3 * med:.2f} ms per step"

