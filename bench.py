#!/usr/bin/env python3
"""Headline benchmark: audio-seconds per second for encode()+decode() (RTF^-1).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one batch of synthetic 16 kHz audio already resident in HBM: 32 utterances x
10 s per GPU (BASELINE.json's metric shape; SURVEY.md 8d inputs: 0.1 * N(0,1), torch.Generator seed 1234, one draw per
utterance in order).  Weights: the closed-form synthetic checkpoint (291 M parameters, random-init statistics).

Two measurements, each W warm-up steps then EXACTLY K timed steps bracketed by barrier + device sync, MAX over ranks:
  scatter_gather  (`value`)  — BASELINE.json configs[3]: rank 0 owns all N x 32 utterances in its HBM; a step is
        DataParallelCodec.encode_decode on the RCCL backend: scatter of the audio over xGMI, encode -> decode on every
        GPU, gather of codes and waveforms back to rank 0, all inside the timed region.  For N = 1 the scatter / gather
        degenerates to the single-GPU step (a single-rank RCCL group is still initialised, so the code path is the same).
  independent_shards (second field) — every rank encodes + decodes its own 32 utterances; no data-path traffic at all.
  two_in_flight (third field) — the independent-shard steps with two batches in flight per GPU (pipeline.InFlight): the
        serving loop's gain from overlapping consecutive batches; an extra, never `value`.
Rank 0 prints ONE JSON line, with
  roofline     — the dominant kernel family (by accumulated device time), timed live with events on the launching stream
                 inside the timed steps (every 20th timed step, counting back from the last, carries the event pairs: ~0.8 ms each);
                 achieved = algorithmic FLOPs of the sampled launches / their summed duration; peak = dense MFMA peak of
                 its dtype; traffic = HBM bytes per launch from the committed rocprofv3 --pmc passes, tagged with the commit
                 they were taken at (a stale value is visible as a different commit).
  cpu_baseline — the CPU oracle (oracle/ref_cpu.py, the reference's algorithm incl. its 30 s padding) timed on this
                 box's host cores on a bounded sample of the same workload (N = 1 only): B = 8 x 10 s (BASELINE.json
                 configs[1]'s shape, the first 8 draws of the same generator), 1 warm-up + 3 timed passes, median (~60 s), on
                 every core the process may use (`cores` = threads used = min(`cores_visible` = len(sched_getaffinity),
                 `cpu_quota` = the cgroup's cpu.max)).
  parity       — the run checks its own answer: the codes of utterance 0 of the LAST timed step must equal the CPU oracle's
                 (at most 2 of 1000 may differ — the oracle's own thread-count sensitivity; the count is reported and has been 0
                 on every box) and its waveform must lie within the bf16-decode tolerance of the oracle's (N = 1, cpu baseline
                 on: the oracle's warm-up pass supplies them); otherwise utterance 0 encoded alone on the GPU (rows of a
                 batch are independent).  A wrong answer fails the bench.
  other_configs — after the metric measurements (N = 1): the other BASELINE.json configs, a few steps each, every one with
                 its own dominant-kernel roofline: B = 8 x 10 s `bf16` (configs[1]), 32 x 30 s `mixed` (configs[2]),
                 32 x 10 s `fp8` (configs[4]) and 32 x 10 s `fp32` (the reference's own arithmetic).  Never `value`.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import yaml  # noqa: E402

# MI355X dense MFMA peaks (MI355X_MICROARCH.md).  gemm_f16s executes 3 f16 MFMA passes per algorithmic multiply-add
# (hi*hi + hi*lo + lo*hi), so its ceiling in algorithmic FLOP/s is 2500 / 3.  gemm_fp8 runs on the block-scaled
# v_mfma_scale_f32_16x16x128_f8f6f4 (2x the bf16 rate).
PEAK_TFLOPS = {"gemm_bf16": 2500.0, "convnext_bf16": 2500.0, "mlp_bf16": 2500.0, "mlp_fp8fc1": 2500.0, "gemm_f32": 157.3,
               "gemm_f16s": 2500.0 / 3.0, "gemm_fp8": 5000.0}


class KernelTimer:
    """Per-launch event pairs around the MFMA kernels (ops.PROFILER hook), on the launching stream."""

    def __init__(self):
        self.rec = []
        self._cur = None

    def begin(self, kind, flops, chip_share=1.0):
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        self._cur = (kind, flops, e0, chip_share)

    def end(self):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        kind, flops, e0, share = self._cur
        self.rec.append((kind, flops, e0, e1, share))

    def summary(self):
        out = {}
        for kind, flops, e0, e1, share in self.rec:
            d = out.setdefault(kind, {"ms": 0.0, "flops": 0.0, "launches": 0})
            # a launch that shares the chip with a second chain on another stream (two half-batch chains of ConvNeXt
            # blocks, 125 workgroups each) is charged its share of the duration: the rate stays a whole-chip rate
            d["ms"] += e0.elapsed_time(e1) * share
            d["flops"] += flops
            d["launches"] += 1
        return out


def pmc_traffic(kind):
    """(HBM bytes per launch, provenance) of a kernel family from the committed rocprofv3 --pmc passes of this same command
    (tools/pmc_traffic.sh: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate passes).  PMC counters cannot be
    collected from inside the timed run, so this is the profiled value with the commit it was taken at, or (None, None)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")) as f:
            d = json.load(f)
        return round(d[kind]["hbm_bytes_per_launch"]), {"profile": d.get("_profile"), "commit": d.get("_commit")}
    except Exception:
        return None, None


KERNEL_NAMES = {"convnext_bf16": "swc_convnext_block (bf16)", "mlp_bf16": "swc_layer_tail / swc_mlp_block (16-bit operands: f16 inside the kernel, bf16 at its boundary)"}


def roofline_of(summ, step_ms, n_sampled, brief=False):
    """the `roofline` object from a KernelTimer summary: the dominant family (by accumulated device time) priced against the
    dense MFMA peak of its dtype, the others listed with their own fraction."""
    if not summ or not n_sampled:
        return None
    kind = max(summ, key=lambda k: summ[k]["ms"])
    d = summ[kind]
    ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
    out = {"bound": "mfma", "kernel": KERNEL_NAMES.get(kind, f"swc_gemm ({kind})"), "achieved": round(ach, 2),
           "peak": round(PEAK_TFLOPS[kind], 1), "unit": "TFLOP/s", "frac": round(ach / PEAK_TFLOPS[kind], 4),
           "launches_per_step": d["launches"] // n_sampled, "share_of_step": round(d["ms"] / n_sampled / step_ms, 3),
           "other": {k: {"TFLOP/s": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                         "frac": round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / PEAK_TFLOPS[k], 4),
                         "launches_per_step": v["launches"] // n_sampled,
                         "share_of_step": round(v["ms"] / n_sampled / step_ms, 3)}
                     for k, v in summ.items() if k != kind}}
    if brief:
        return out
    traffic, src = pmc_traffic(kind)
    tot_ms = sum(v["ms"] for v in summ.values())
    tot_fl = sum(v["flops"] for v in summ.values())
    out.update({"traffic": traffic, "traffic_source": src, "sampled_steps": n_sampled,
                "avg_launch_ms": round(d["ms"] / d["launches"], 4),
                "all_mfma_kernels": {"TFLOP/s": round(tot_fl / (tot_ms * 1e-3) / 1e12, 2),
                                     "share_of_step": round(tot_ms / n_sampled / step_ms, 3)},
                "commit": git_commit()})
    return out


def bench_inputs(n_utt, n_samples):
    """SURVEY.md 8d / BASELINE.md 2: 0.1 * N(0,1), torch.Generator().manual_seed(1234), one draw per utterance in order."""
    g = torch.Generator().manual_seed(1234)
    return [0.1 * torch.randn(n_samples, generator=g) for _ in range(n_utt)]


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_allotment():
    """(logical CPUs in this process's affinity mask, CPUs the cgroup quota grants or None).  The one-GPU boxes of the pool
    show all 256 logical CPUs of a 2 x 64-core EPYC in the mask and grant 16 through `cpu.max` (1600000 100000): threads
    beyond the quota only add throttling, so the quota is the core count a baseline can use there."""
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2
            q, per = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:  # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = float(f.read()), float(g.read())
                if q > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    return visible, quota


def cpu_baseline(gp, sd, shapes, seconds, threads, timed=3):
    """oracle/ref_cpu.py on the host cores: per shape 1 warm-up + `timed` timed encode+decode passes, median."""
    from oracle.ref_cpu import Oracle
    torch.set_num_threads(threads)
    ora = Oracle(gp, sd)
    res = []
    t_all = time.perf_counter()
    first = None
    for n_utt in shapes:
        wavs = bench_inputs(n_utt, int(seconds * 16000))
        times = []
        for it in range(1 + timed):
            print(f"bench: cpu baseline B={n_utt} pass {it}/{timed} on {threads} threads ...", file=sys.stderr, flush=True)
            t0 = time.perf_counter()
            codes = ora.encode(wavs)["codes_list"]
            wav = ora.decode(codes)["syn_wav_list"]
            if it:
                times.append(time.perf_counter() - t0)
            elif first is None:  # the oracle's answer for the first utterances of the workload: the bench's parity check
                n_code = int(seconds * 16000) // 1280
                first = {"codes": [c[:, :n_code].long() for c in codes], "wav": [w[: n_code * 1280].float() for w in wav]}
        res.append({"batch": n_utt, "audio-s/s": round(n_utt * seconds / statistics.median(times), 3),
                    "pass_s": [round(t, 2) for t in times]})
    head = res[-1]
    return {"value": head["audio-s/s"], "unit": "audio-s/s", "cores": threads, "kind": "port", "cpu": cpu_model(),
            "sample": f"{head['batch']} x {seconds:g} s utterances (same generator as the GPU run), 1 warm-up + {timed} timed "
                      f"encode+decode passes, median; fp32, oracle/ref_cpu.py (reference algorithm incl. 30 s padding); "
                      f"{time.perf_counter() - t_all:.0f} s of CPU work",
            "shapes": res}, first


LEVEL_FLOORS = {"bf16": (0.97, 0.9995), "fp8": (0.84, 0.995), "fp8_fc1": (0.94, 0.9995)}  # tests/test_parity_gpu.py: FSQ levels equal / within one


def other_parity(mdl, prec, seconds, wavs, out, oracle_key, dev):
    """The answer check of one `other_configs` entry (the timed steps' last output against an answer key), never a shape
    assert only.  Key: the CPU oracle's first utterances when this run holds them and the config is 10 s long (the same
    generator draws: utterance i of every 10 s config is utterance i of the metric workload), else utterance 0 encoded +
    decoded ALONE on the GPU by the same model (rows of a uniform batch are independent: bit-exact).
      fp32 / mixed / mixed_f32: codes of every keyed utterance equal (<= 2 of utterance 0's against the CPU oracle, its own
      thread-count sensitivity), waveform within the preset's tolerance;   bf16 / fp8: FSQ-level floors over the keyed utterances
      (codes agree statistically only), waveform given the key's codes within the bf16 tolerance."""
    codes = mdl.encode(wavs, overlap_seconds=10, device=dev)["codes_list"]
    if oracle_key is not None and seconds == 10.0:
        key_c, key_w, src = oracle_key["codes"], oracle_key["wav"], oracle_key["source"]
    else:
        c0 = mdl.encode(wavs[:1], overlap_seconds=10, device=dev)["codes_list"]
        w0 = mdl.decode(c0, overlap_seconds=10, device=dev)["syn_wav_list"]
        key_c, key_w, src = [c0[0].long().cpu()], [w0[0].float().cpu()], "utterance 0 alone on the GPU (batch independence)"
    n = min(len(key_c), len(codes))
    tol = {"fp32": 5e-5, "f16s": 1e-4}.get(prec, 5e-2)
    res = {"against": src, "utterances": n}
    if prec in LEVEL_FLOORS:  # (against the GPU's own batch of one too: it runs the unfused layers, a 16-bit encoder is not bit-stable across them)
        base, lev = torch.tensor([1, 8, 56, 336]), torch.tensor([8, 7, 6, 6])
        same = within1 = total = 0
        for i in range(n):
            a, b = codes[i].long().cpu(), key_c[i]
            d = (((a[..., None] // base) % lev) - ((b[..., None] // base) % lev)).abs()
            same += int((d == 0).sum()); within1 += int((d <= 1).sum()); total += d.numel()
        lo_eq, lo_w1 = LEVEL_FLOORS[prec]
        res.update({"levels_equal": round(same / total, 4), "levels_within_one": round(within1 / total, 5), "floors": [lo_eq, lo_w1]})
        assert same / total >= lo_eq and within1 / total >= lo_w1, f"FSQ levels {same / total:.4f} / {within1 / total:.5f} under {lo_eq} / {lo_w1}"
        # the decoder given the KEY's codes for the keyed rows
        mix = [key_c[i].to(dev).to(codes[i].dtype) if i < n else codes[i] for i in range(len(codes))]
        wl = mdl.decode(mix, overlap_seconds=10, device=dev)["syn_wav_list"]
    else:
        allowed = 2 if src.startswith("oracle") else 0
        mism = [int((codes[i].long().cpu() != key_c[i]).sum()) for i in range(n)]
        res.update({"code_mismatches": sum(mism), "codes_compared": sum(int(key_c[i].numel()) for i in range(n)),
                    "code_mismatches_allowed_utt0": allowed})
        assert mism[0] <= allowed, f"{mism[0]} codes of utterance 0 differ from {src}"
        wl = [w if m_ == 0 else None for w, m_ in zip(out["syn_wav_list"][:n], mism)]
    worst = 0.0
    for i in range(n):
        if wl[i] is None:
            continue  # a flipped borderline code changes the waveform legitimately
        e = float((wl[i].float().cpu() - key_w[i]).abs().max() / (key_w[i].abs().max() + 1e-12))
        worst = max(worst, e)
        assert i != 0 or e <= tol, f"waveform of utterance 0 is {e:.3e} from {src}, tolerance {tol}"
    res.update({"waveform_rel_err_max": worst, "waveform_tolerance": tol})
    return res


class _StdoutToStderr:
    """RCCL prints a version banner on stdout when its communicator is created; the bench's stdout carries one JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def git_commit():
    """HEAD of the checkout, or (on a GPU box: snapshot without .git) the commit stamped beside the library at build time."""
    try:
        c = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stdout=subprocess.PIPE,
                           stderr=subprocess.DEVNULL, text=True, timeout=5).stdout.strip()
        if c:
            return c
    except Exception:
        pass
    try:
        with open(os.path.join(ROOT, "simwhisper_codec_amd", "_build_commit.txt")) as f:
            return f.read().strip() or None
    except OSError:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--precision", default="mixed", choices=["fp32", "mixed", "mixed_f32", "bf16", "fp8", "fp8_fc1", "f16s"])
    ap.add_argument("--cpu-baseline", default="sample", choices=["sample", "small", "full", "off"],
                    help="sample: B=8 x 10 s, BASELINE.json configs[1]'s shape (~25 s of CPU work); small: 2 utterances "
                         "(~15 s); full: B=8 and B=32 as BASELINE.md 2 (minutes)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="host threads of the CPU baseline (default: every visible core)")
    ap.add_argument("--no-timer", action="store_true", help="no per-launch event pairs (no roofline object)")
    ap.add_argument("--no-dist", action="store_true", help="N=1 only: skip the process group and the scatter/gather measurement")
    ap.add_argument("--no-inflight", action="store_true",
                    help="skip the two-batches-in-flight extra (profiling passes: only serial steps in the kernel trace)")
    ap.add_argument("--other-configs", default="auto", choices=["auto", "on", "off"],
                    help="the other BASELINE.json configs after the metric (auto: N=1 and the default workload / preset)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus and world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    # SWC_BENCH_BACKEND=gloo: rehearsal of the multi-rank control flow on ONE card (every rank computes on a visible GPU,
    # LOCAL_RANK modulo their number; traffic goes through host memory).  RCCL refuses two ranks on one GPU, so this is the
    # only way to run `--gpus 2` on a one-GPU box; the numbers it prints are not the metric.
    backend = os.environ.get("SWC_BENCH_BACKEND", "nccl")
    local = local % max(torch.cuda.device_count(), 1) if backend != "nccl" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # one process per GPU over RCCL.  A single rank also gets its (one-member) group, so that N = 1 runs the very code
    # path the 8-GPU node runs: init, tensor broadcasts, barrier, MAX all-reduce, DataParallelCodec.
    use_dist, dist_note = False, None
    if world > 1 or not args.no_dist:
        import torch.distributed as dist
        try:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                s = socket.socket(); s.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(s.getsockname()[1]); s.close()
            import datetime
            # a failed peer must not leave the others waiting for the default 10 minutes
            with _StdoutToStderr():
                if backend == "nccl":
                    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev,
                                            timeout=datetime.timedelta(seconds=240))
                else:
                    dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=240))
                dist.barrier()  # creates the communicator now (and prints RCCL's banner to stderr)
                torch.cuda.synchronize(dev)
            use_dist = True
        except Exception as e:  # a single GPU still has its number; N > 1 cannot run without the group
            if world > 1:
                raise
            dist_note = f"single-rank RCCL group unavailable ({type(e).__name__}: {e}); plain single-GPU step timed"

    from simwhisper_codec_amd import ops, synth
    from simwhisper_codec_amd.codec import AudioCodec
    from simwhisper_codec_amd.dist import DataParallelCodec

    gp = yaml.safe_load(open(os.path.join(ROOT, "config", "SimWhisperCodec.yaml")))["generator_params"]
    sd = synth.synth_state_dict(gp)

    def build(precision):
        m = AudioCodec(gp, precision=precision)
        m.load_state_dict(sd, strict=True)
        return m.to(dev).eval()

    model = build(args.precision)
    n = int(args.seconds * 16000)
    all_wavs = bench_inputs(world * args.batch, n)  # the same draws on every rank; a rank keeps what it needs
    mine = [w.to(dev) for w in all_wavs[rank * args.batch:(rank + 1) * args.batch]]
    n_out = (n // 1280) * 1280

    # ---- the answer the timed steps must reproduce (outside every timed region).  N = 1 with the CPU baseline on: the
    # oracle's own codes / waveform of the first utterances (its warm-up pass; the baseline is timed here, before the GPU
    # measurements, for that reason).  Otherwise: utterance 0 of this rank encoded + decoded ALONE on the GPU — rows of a
    # uniform batch are independent, so the batched step must give the same bits.
    cpu_res, expect = None, None
    exact_codes = args.precision in ("fp32", "mixed", "mixed_f32", "f16s")  # bf16 / fp8 / fp8_fc1 encoders agree statistically only
    wav_tol = {"fp32": 5e-5, "f16s": 1e-4}.get(args.precision, 5e-2)   # tests/test_parity_gpu.py TOL_FP32 / TOL_F16S / TOL_BF16
    if world == 1 and rank == 0 and args.cpu_baseline != "off":
        # every core this process may USE (SURVEY.md 8d / BASELINE.md 2: "N = physical cores; report N"): the affinity mask
        # capped by the cgroup's CPU quota (the one-GPU box: 256 logical CPUs visible, 16 granted).  --cpu-threads overrides;
        # tools/cpu_threads.py sweeps the count (profiles/r04_cpu_threads.txt: more threads than the quota are slower)
        visible, quota = cpu_allotment()
        threads = args.cpu_threads or max(1, min(visible, int(quota) if quota else visible))
        shapes = {"sample": [min(8, args.batch)], "small": [min(2, args.batch)], "full": [8, 32]}[args.cpu_baseline]
        cpu_res, first = cpu_baseline(gp, sd, shapes, args.seconds, threads, timed=1 if args.cpu_baseline == "small" else 3)
        cpu_res["cores_visible"], cpu_res["cpu_quota"] = visible, quota
        expect = {"source": "oracle/ref_cpu.py (CPU, fp32)", "codes": first["codes"], "wav": first["wav"]}
    else:
        r0 = model.encode(mine[:1], overlap_seconds=10, device=dev)["codes_list"]
        w0 = model.decode(r0, overlap_seconds=10, device=dev)["syn_wav_list"]
        expect = {"source": "utterance 0 alone on the GPU (batch independence)", "codes": [r0[0].long().cpu()],
                  "wav": [w0[0].float().cpu()]}
        # (the waveform tolerance stays: a batch of one takes the two-GEMM form of the ConvNeXt blocks.  Presets with a bf16 /
        # fp8 encoder are checked on FSQ levels here as well: a batch of one runs the unfused transformer layers, the batch
        # swc_layer_tail, and a 16-bit encoder is not bit-stable across the two)
    # the CPU oracle's own codes move with its host thread count (2 of 94 544 between 1 and 16 threads,
    # profiles/r02_code_agreement.txt: a latent within 1e-6 of a rounding boundary): against IT up to 2 of utterance 0's 1 000
    # codes may differ (a wrong kernel flips hundreds; the count is on the line and has been 0 on every box); against the
    # GPU's own single-utterance run nothing may differ (batch independence is bit-exact)
    allowed = 2 if expect["source"].startswith("oracle") else 0
    parity = {}

    def check_answer(codes_list, wav_list):
        """codes of utterance 0 bit for bit (presets with an f32-class encoder), waveform of utterance 0 within the decode
        tolerance; agreement over all the rows the source covers is reported."""
        mism = total = 0
        worst = 0.0
        for i, want in enumerate(expect["codes"]):
            got = codes_list[i].long().cpu()
            assert got.shape == want.shape, (got.shape, want.shape)
            d = int((got != want).sum())
            if i == 0 and not exact_codes and args.precision in LEVEL_FLOORS:
                base, lev = torch.tensor([1, 8, 56, 336]), torch.tensor([8, 7, 6, 6])
                dl = (((got[..., None] // base) % lev) - ((want[..., None] // base) % lev)).abs()
                eq, w1 = float((dl == 0).float().mean()), float((dl <= 1).float().mean())
                lo_eq, lo_w1 = LEVEL_FLOORS[args.precision]
                parity.update({"levels_equal_utt0": round(eq, 4), "levels_within_one_utt0": round(w1, 5), "floors": [lo_eq, lo_w1]})
                assert eq >= lo_eq and w1 >= lo_w1, f"bench: FSQ levels of utterance 0 {eq:.4f} / {w1:.5f} under {lo_eq} / {lo_w1} ({expect['source']})"
            if i == 0 and exact_codes:
                # (the bit-exact bar against the oracle is held by tests/test_metric_shape_gpu.py and tests/test_parity_gpu.py)
                assert d <= allowed, f"bench: {d} of {want.numel()} codes of utterance 0 differ from {expect['source']}"
            mism += d; total += want.numel()
        if exact_codes:  # waveform given the SAME codes; presets with a statistical encoder differ in codes
            for i, want in enumerate(expect["wav"]):
                got = wav_list[i].float().cpu()
                assert got.shape == want.shape, (got.shape, want.shape)
                if int((codes_list[i].long().cpu() != expect["codes"][i]).sum()):
                    continue  # a flipped borderline code changes the waveform legitimately
                e = float((got - want).abs().max() / (want.abs().max() + 1e-12))
                if i == 0:
                    assert e <= wav_tol, f"bench: waveform of utterance 0 is {e:.3e} (relative to peak) from {expect['source']}, tolerance {wav_tol}"
                worst = max(worst, e)
        parity.update({"against": expect["source"], "utterances": len(expect["codes"]), "codes_compared": total,
                       "code_mismatches": mism, "codes_must_match": bool(exact_codes),
                       "code_mismatches_allowed_utt0": allowed if exact_codes else None, "waveform_rel_err_max": worst,
                       "waveform_tolerance": wav_tol})

    def fence():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def timed(step, check, timer):
        """W warm-ups, then exactly K steps between two fences; returns (seconds MAX over ranks, per-step ms list)."""
        for _ in range(args.warmup):
            step()
        sampled = set(range(args.steps - 1, -1, -20)) if timer is not None else set()  # event pairs around every launch cost ~0.8 ms per sampled step
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        fence()
        t0 = time.perf_counter()
        marks[0].record()
        out = None
        for i in range(args.steps):
            ops.PROFILER = timer if i in sampled else None
            out = step()
            marks[i + 1].record()
        ops.PROFILER = None
        fence()
        elapsed = time.perf_counter() - t0
        # the check is collective: a rank whose answer is wrong must not leave the others inside the next all-reduce
        # (they would pair up with a later collective of this rank): every rank learns the verdict, all raise together
        err = None
        try:
            check(out)
        except AssertionError as e:
            err = e
        if use_dist:
            t = torch.tensor([elapsed, 0.0 if err is None else 1.0], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed, bad = float(t[0].item()), float(t[1].item()) > 0
        else:
            bad = err is not None
        if err is not None:
            raise err
        if bad:
            raise AssertionError("bench: the answer check failed on another rank")
        per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
        return elapsed, per_step, len(sampled)

    # ---- measurement A: independent shards (no data-path traffic)
    def step_local():
        enc = model.encode(mine, overlap_seconds=10, device=dev)
        out = model.decode(enc["codes_list"], overlap_seconds=10, device=dev)
        out["codes_list"] = enc["codes_list"]
        return out

    def check_local(out):
        assert len(out["syn_wav_list"]) == args.batch and out["syn_wav_list"][0].shape[0] == n_out
        if rank == 0:
            check_answer(out["codes_list"], out["syn_wav_list"])

    timer = None if args.no_timer else KernelTimer()
    el_a, steps_a, n_sampled = timed(step_local, check_local, None if use_dist else timer)
    res_a = {"value": round(world * args.batch * args.seconds * args.steps / el_a, 2),
             "ms_per_step": round(1e3 * el_a / args.steps, 3), "ms_per_step_median": round(statistics.median(steps_a), 3)}

    # ---- measurement B: rank 0 scatters / gathers over RCCL (configs[3])
    res_b, b_err = None, None
    if use_dist:
        try:
            dp = DataParallelCodec(model, dev, comm_device=None if backend == "nccl" else "cpu")
            owned = [w.to(dev) for w in all_wavs] if rank == 0 else None

            def step_dp():
                return dp.encode_decode(owned, overlap_seconds=10)

            def check_dp(out):
                if rank == 0:
                    assert len(out["syn_wav_list"]) == world * args.batch and out["syn_wav_list"][-1].shape[0] == n_out
                    assert len(out["codes_list"]) == world * args.batch and out["codes_list"][-1].shape[-1] == n // 1280
                    check_answer(out["codes_list"], out["syn_wav_list"])

            el_b, steps_b, n_sampled = timed(step_dp, check_dp, timer)
            res_b = {"value": round(world * args.batch * args.seconds * args.steps / el_b, 2),
                     "ms_per_step": round(1e3 * el_b / args.steps, 3),
                     "ms_per_step_median": round(statistics.median(steps_b), 3)}
        except Exception as e:
            if world > 1:
                # every rank fails or none: the error is raised collectively by RCCL; report the shard number with the reason
                b_err = f"{type(e).__name__}: {e}"
            else:
                raise
    # ---- measurement C (after the two metric measurements, so that it cannot warm the chip for them): the same K steps with TWO batches in flight (pipeline.InFlight: two host threads, two streams, two
    # model objects over one set of operands): what the serving loop around the path gains when consecutive batches
    # overlap.  Reported beside `value`, never as `value`: a step there is one batch at a time.
    res_a2 = None
    try:
        if args.no_inflight:
            raise InterruptedError
        from simwhisper_codec_amd.pipeline import InFlight
        with InFlight(model, 2) as pipe:
            def step_pipe(mdl, w):
                enc = mdl.encode(w, overlap_seconds=10, device=dev)
                out = mdl.decode(enc["codes_list"], overlap_seconds=10, device=dev)
                out["codes_list"] = enc["codes_list"]
                return out
            pipe.map(step_pipe, [mine] * max(2, args.warmup))
            fence()
            t0 = time.perf_counter()
            outs = pipe.map(step_pipe, [mine] * args.steps)
            fence()
            el_a2 = time.perf_counter() - t0
            wrong = None
            try:
                check_local(outs[-1])
            except AssertionError as e:  # reported in the field; the all-reduce below still happens on every rank
                wrong = str(e)
            if use_dist:
                t = torch.tensor([el_a2], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el_a2 = float(t.item())
            res_a2 = {"value": round(world * args.batch * args.seconds * args.steps / el_a2, 2),
                      "ms_per_step": round(1e3 * el_a2 / args.steps, 3),
                      "note": "independent shards, two batches in flight per GPU (two streams); results bit-identical to one at a time"}
            if wrong:
                res_a2 = {"error": f"answer check failed: {wrong}"}
    except InterruptedError:
        res_a2 = {"skipped": "--no-inflight"}
    except Exception as e:  # an extra: never fails the bench line
        res_a2 = {"error": f"{type(e).__name__}: {e}"}

    main_res = res_b if res_b is not None else res_a
    el_main = el_b if res_b is not None else el_a

    # ---- the other BASELINE.json configs (N = 1, after everything the metric needs): a few steps each of encode()+decode()
    # on this same GPU, never `value`.  Each line carries its own dominant-kernel roofline (sampled on its last step).
    others = None
    default_run = args.batch == 32 and args.seconds == 10.0 and args.precision == "mixed"
    if world == 1 and (args.other_configs == "on" or (args.other_configs == "auto" and default_run)):
        others = {}
        del model
        torch.cuda.empty_cache()
        for tag, (ob, osec, oprec, osteps) in {
                "configs[1] batch=8x10s bf16": (8, 10.0, "bf16", 10),
                "configs[2] batch=32x30s mixed": (32, 30.0, "mixed", 4),
                "configs[4] batch=32x10s fp8 encoder linears": (32, 10.0, "fp8", 8),
                "configs[4'] batch=32x10s fp8 fc1 only (bf16's class of code agreement)": (32, 10.0, "fp8_fc1", 8),
                "batch=32x10s bf16 (both sides)": (32, 10.0, "bf16", 8),
                "batch=32x10s f16s (split-f16 on both sides: f32-class waveforms on the MFMA kernels)": (32, 10.0, "f16s", 4),
                "batch=32x10s fp32 (the reference's arithmetic)": (32, 10.0, "fp32", 3),
                "batch=32x10s mixed_f32 (exact-f32 encoder, the range guard's fallback)": (32, 10.0, "mixed_f32", 4)}.items():
            try:
                mdl = build(oprec)
                w = [x.to(dev) for x in bench_inputs(ob, int(osec * 16000))]

                def ostep():
                    return mdl.decode(mdl.encode(w, overlap_seconds=10, device=dev)["codes_list"], overlap_seconds=10, device=dev)
                for _ in range(2):
                    ostep()
                otimer = KernelTimer()
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for i in range(osteps):
                    ops.PROFILER = otimer if i == osteps - 1 else None
                    o = ostep()
                ops.PROFILER = None
                torch.cuda.synchronize(dev)
                el = time.perf_counter() - t0
                assert len(o["syn_wav_list"]) == ob and o["syn_wav_list"][0].shape[0] == (int(osec * 16000) // 1280) * 1280
                others[tag] = {"value": round(ob * osec * osteps / el, 1), "unit": "audio-s/s", "ms_per_step": round(1e3 * el / osteps, 3),
                               "steps": osteps, "roofline": roofline_of(otimer.summary(), 1e3 * el / osteps, 1, brief=True),
                               "parity": other_parity(mdl, oprec, osec, w, o, expect if cpu_res is not None else None, dev)}
                del mdl, w, o
                torch.cuda.empty_cache()
            except Exception as e:  # an extra: never fails the bench line
                ops.PROFILER = None
                others[tag] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        via = "RCCL point-to-point (xGMI)" if backend == "nccl" else f"{backend} through host memory (REHEARSAL on shared cards, not the metric)"
        par = (f"dp{world}: rank 0 scatters the audio / gathers codes + waveforms over {via}, "
               f"utterance shards of {args.batch} per GPU" if res_b is not None
               else f"dp{world} (independent utterance shards, no collective)")
        line = {
            "metric": f"audio-sec/sec encode+decode (RTF^-1), 16kHz batch={args.batch}x{args.seconds:g}s",
            "value": main_res["value"], "unit": "audio-s/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": main_res["ms_per_step"],
            "ms_per_step_median": main_res["ms_per_step_median"], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "mixed": "split-f16 x3 encode (f32-class, f32 accumulate) / bf16 decode (f32 accumulate)",
                      "mixed_f32": "f32 encode / bf16 decode (f32 accumulate)", "bf16": "bf16",
                      "fp8": "fp8 (e4m3, block-scaled MFMA) encoder-transformer linears / bf16 elsewhere (f32 accumulate)",
                      "fp8_fc1": "fp8 (e4m3, block-scaled MFMA) encoder fc1 / bf16 elsewhere (f32 accumulate)",
                      "f16s": "split-f16 x3 on both sides (f32-class, f32 accumulate)"}[args.precision],
            "data": "synthetic",
            "config": {"workload": f"batch={args.batch}x{args.seconds:g}s @16kHz per GPU ({world * args.batch} utterances in all), "
                                   f"encode()+decode(), 0.1*N(0,1) seed 1234, synthetic closed-form checkpoint (291M params)",
                       "precision": args.precision, "parallelism": par},
            "independent_shards": res_a,
            "two_in_flight": res_a2,
        }
        if b_err:
            line["scatter_gather_error"] = b_err
        if dist_note:
            line["note"] = dist_note
        if timer is not None:
            rf = roofline_of(timer.summary(), 1e3 * el_main / args.steps, n_sampled)
            if rf:
                line["roofline"] = rf
        if parity:
            line["parity"] = parity
        if others is not None:
            line["other_configs"] = others
        if cpu_res is not None:
            if args.cpu_baseline == "sample":
                cpu_res["full_shapes"] = ("B = 8 and B = 32 x 10 s (BASELINE.md 2) with `--cpu-baseline full`: "
                                          "profiles/r04_cpu_baseline_full.json (5.97 / 5.80 audio-s/s, 16 threads = the cgroup quota, EPYC 9575F)")
            line["cpu_baseline"] = cpu_res
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
