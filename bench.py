#!/usr/bin/env python3
"""Headline benchmark: audio-seconds per second for encode()+decode() (RTF^-1).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path (AudioCodec.encode then AudioCodec.decode) over one batch of
synthetic 16 kHz audio already resident in HBM: 32 utterances x 10 s per GPU (BASELINE.json's
metric shape; weak scaling: every rank processes its own 32 utterances, no data-path collective).
Weights: the closed-form synthetic checkpoint (291 M parameters, random-init statistics).
Rank 0 prints ONE JSON line, with
  roofline     — the dominant kernel family (swc_gemm, by accumulated device time), timed live with
                 events on the launching stream inside the timed steps (every 5th timed step, counting back
                 from the last, carries the event pairs); achieved = 2*M*N*K*taps summed over the sampled
                 launches / their summed duration; peak = dense MFMA peak of its dtype.
  cpu_baseline — the CPU oracle (oracle/ref_cpu.py, the reference's algorithm incl. its 30 s padding)
                 timed on this box's host cores on a bounded sample of the same workload (N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import yaml  # noqa: E402

# MI355X dense MFMA peaks (MI355X_MICROARCH.md).  gemm_f16s executes 3 f16 MFMA passes per algorithmic
# multiply-add (hi*hi + hi*lo + lo*hi), so its ceiling in algorithmic FLOP/s is 2500 / 3.
PEAK_TFLOPS = {"gemm_bf16": 2500.0, "gemm_f32": 157.3, "gemm_f16s": 2500.0 / 3.0, "gemm_fp8": 5000.0}


class GemmTimer:
    """Per-launch event pairs around swc_gemm (ops.PROFILER hook)."""

    def __init__(self):
        self.rec = []
        self._cur = None

    def begin(self, kind, flops):
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        self._cur = (kind, flops, e0)

    def end(self):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        kind, flops, e0 = self._cur
        self.rec.append((kind, flops, e0, e1))

    def summary(self):
        out = {}
        for kind, flops, e0, e1 in self.rec:
            d = out.setdefault(kind, {"ms": 0.0, "flops": 0.0, "launches": 0})
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += flops
            d["launches"] += 1
        return out


def pmc_traffic(kind):
    """HBM bytes per launch of the dominant kernel family from the committed rocprofv3 --pmc passes
    (profiles/r01_pmc_gemm_traffic.json: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate passes of this
    same command).  PMC cannot be collected from inside the timed run, so this is the profiled value, or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_gemm_traffic.json")) as f:
            return round(json.load(f)[kind]["hbm_bytes_per_launch"])
    except Exception:
        return None


def cpu_baseline(gp, sd, n_utt, seconds, threads):
    from oracle.ref_cpu import Oracle
    from simwhisper_codec_amd import synth
    torch.set_num_threads(threads)
    ora = Oracle(gp, sd)
    wavs = [synth.synth_audio(int(seconds * 16000), index=1000 + i) for i in range(n_utt)]
    t0 = time.perf_counter()
    codes = ora.encode(wavs)["codes_list"]
    ora.decode(codes)
    dt = time.perf_counter() - t0
    return {"value": round(n_utt * seconds / dt, 3), "unit": "audio-s/s", "cores": threads, "kind": "port",
            "sample": f"{n_utt} x {seconds:g} s utterances, encode+decode once, fp32, oracle/ref_cpu.py "
                      f"(reference algorithm incl. 30 s padding), {dt:.1f} s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--precision", default="mixed", choices=["fp32", "mixed", "mixed_f32", "bf16", "fp8"])
    ap.add_argument("--cpu-utts", type=int, default=8, help="utterances in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-gemm-timer", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # one process per GPU over RCCL; also initialised for a single rank launched by torch.distributed.run, so that the
    # N > 1 code path (init, barrier, MAX all-reduce) can be exercised on a one-GPU box
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from simwhisper_codec_amd import ops, synth
    from simwhisper_codec_amd.codec import AudioCodec

    gp = yaml.safe_load(open(os.path.join(ROOT, "config", "SimWhisperCodec.yaml")))["generator_params"]
    sd = synth.synth_state_dict(gp)
    model = AudioCodec(gp, precision=args.precision)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    n = int(args.seconds * 16000)
    wavs = [synth.synth_audio(n, index=rank * args.batch + i).to(dev) for i in range(args.batch)]

    def step():
        enc = model.encode(wavs, overlap_seconds=10, device=dev)
        return model.decode(enc["codes_list"], overlap_seconds=10, device=dev)

    def fence():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    timer = None if args.no_gemm_timer else GemmTimer()
    # the event pairs cost ~3 % of a step (218 event records), so they are placed on every 5th timed step only
    sampled = set(range(args.steps - 1, -1, -5))
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ops.PROFILER = timer if i in sampled else None
        out = step()
    ops.PROFILER = None
    fence()
    elapsed = time.perf_counter() - t0
    assert len(out["syn_wav_list"]) == args.batch and out["syn_wav_list"][0].shape[0] == (n // 1280) * 1280
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        audio_s = world * args.batch * args.seconds * args.steps
        line = {
            "metric": f"audio-sec/sec encode+decode (RTF^-1), 16kHz batch={args.batch}x{args.seconds:g}s",
            "value": round(audio_s / elapsed, 2), "unit": "audio-s/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "mixed": "split-f16 x3 encode (f32-class, f32 accumulate) / bf16 decode (f32 accumulate)",
                      "mixed_f32": "f32 encode / bf16 decode (f32 accumulate)", "bf16": "bf16",
                      "fp8": "fp8 (e4m3) encoder-transformer linears / bf16 elsewhere (f32 accumulate)"}[args.precision],
            "data": "synthetic",
            "config": {"workload": f"batch={args.batch}x{args.seconds:g}s @16kHz per GPU, encode()+decode(), "
                                   f"synthetic closed-form checkpoint (291M params)",
                       "precision": args.precision, "parallelism": f"dp{world} (utterance shards, no collective)"},
        }
        if timer is not None:
            summ = timer.summary()
            if summ:
                kind = max(summ, key=lambda k: summ[k]["ms"])
                d = summ[kind]
                ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
                line["roofline"] = {
                    "bound": "mfma", "kernel": f"swc_gemm ({kind})", "achieved": round(ach, 2),
                    "peak": round(PEAK_TFLOPS[kind], 1), "unit": "TFLOP/s", "frac": round(ach / PEAK_TFLOPS[kind], 4),
                    "traffic": pmc_traffic(kind),
                    # informational: the MFMA-only loop of this kernel (no LDS reads, DMA or barriers) sustains 1500 TFLOP/s
                    # on random 16-bit data at the clock the chip holds; not used for `frac`
                    "sustained_mfma_ceiling": {"gemm_bf16": 1500.0, "gemm_f16s": 500.0, "gemm_fp8": 1500.0}.get(kind),
                    "launches_per_step": d["launches"] // len(sampled), "sampled_steps": len(sampled),
                    "avg_launch_ms": round(d["ms"] / d["launches"], 4),
                    "share_of_step": round(d["ms"] / len(sampled) / (1e3 * elapsed / args.steps), 3),
                    "other": {k: {"TFLOP/s": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                                  "share_of_step": round(v["ms"] / len(sampled) / (1e3 * elapsed / args.steps), 3)}
                              for k, v in summ.items() if k != kind},
                }
        if world == 1 and args.cpu_utts > 0:
            threads = min(16, os.cpu_count() or 1)
            line["cpu_baseline"] = cpu_baseline(gp, sd, args.cpu_utts, args.seconds, threads)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
