#!/usr/bin/env python3
"""Batch codec round trip over a directory of audio files — the reference's CLI surface
(inference.py:12-21 there): same flags, same outputs (`<output_dir>/<basename>.wav`, PCM16).

    python inference.py --config_path ./config/SimWhisperCodec.yaml \\
        --checkpoint_path ./weights/SimWhisperCodec.pt --device cuda \\
        --batch_size 8 --input_dir input_wavs --output_dir output_wavs

Differences on purpose: `--device` is passed on to encode()/decode() (the reference forgets to),
file loading for batch i+1 and saving of batch i-1 overlap the GPU work of batch i, `--in_flight 2` (default) keeps two
batches on the GPU at a time (two streams over one set of weights: same files, more throughput), and
`--precision` / `--synthetic_checkpoint` exist because the trained weights cannot be fetched offline.

Several GPUs of one node (BASELINE.json configs[3]): launch the same command under torch.distributed.run

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 inference.py ... --batch_size 32

Rank 0 reads and writes the files; every step takes `--batch_size x world` files, scatters the audio over RCCL / xGMI
(simwhisper_codec_amd.dist.DataParallelCodec.encode_decode), every GPU encodes + decodes its share, and the waveforms
come back to rank 0.  Outputs are the files the single-GPU run writes (sharded decode pads to the global maximum).
"""
import argparse
import logging
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from audiocodec.model import AudioCodec  # noqa: E402
from simwhisper_codec_amd.wavio import find_audio_files, load_audio, save_audio  # noqa: E402


def set_logging(level="INFO"):
    rank = os.environ.get("RANK", 0)
    logging.basicConfig(level=getattr(logging, str(level).upper(), logging.INFO), stream=sys.stdout,
                        format=f"%(asctime)s [RANK {rank}] (%(module)s:%(lineno)d) %(levelname)s : %(message)s")


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--config_path", type=str, default="./config/SimWhisperCodec.yaml")
    p.add_argument("--checkpoint_path", type=str, default="./weights/SimWhisperCodec.pt")
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--batch_size", type=int, default=8)
    p.add_argument("--input_dir", type=str, default="input_wavs")
    p.add_argument("--output_dir", type=str, default="output_wavs")
    p.add_argument("--precision", type=str, default="mixed", choices=["fp32", "mixed", "mixed_f32", "bf16", "fp8"])
    p.add_argument("--synthetic_checkpoint", action="store_true",
                   help="ignore --checkpoint_path and use the closed-form synthetic weights (offline testing)")
    p.add_argument("--in_flight", type=int, default=2,
                   help="batches in flight on the GPU (simwhisper_codec_amd.pipeline.InFlight: consecutive batches overlap on "
                        "two streams; same output files, about 7 %% more throughput; 1 = one batch at a time)")
    p.add_argument("--dist_backend", type=str, default="nccl", help="torch.distributed backend under torch.distributed.run "
                   "(nccl = RCCL; gloo moves the audio through host memory: tests)")
    return p


def load_model(args, device):
    if args.synthetic_checkpoint:
        import yaml
        from simwhisper_codec_amd import synth
        gp = yaml.safe_load(open(args.config_path))["generator_params"]
        model = AudioCodec(gp)
        model.load_state_dict(synth.synth_state_dict(gp), strict=True)
    else:
        model = AudioCodec.load_from_checkpoint(config_path=args.config_path, ckpt_path=args.checkpoint_path)
    model.precision = args.precision
    return model.to(device).eval()


def main(argv=None):
    set_logging()
    args = build_parser().parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world > 1:
        return main_distributed(args, world)
    device = torch.device(args.device)
    generator = load_model(args, device)
    audio_paths = find_audio_files(input_dir=args.input_dir)
    os.makedirs(args.output_dir, exist_ok=True)
    logging.info(f"Processing {len(audio_paths)} audio files, output will be saved to {args.output_dir}")
    bs = args.batch_size
    batches = [audio_paths[i:i + bs] for i in range(0, len(audio_paths), bs)]

    def load(paths):
        return [load_audio(p, target_sample_rate=generator.input_sample_rate).reshape(-1).pin_memory()
                if device.type == "cuda" else load_audio(p, target_sample_rate=generator.input_sample_rate).reshape(-1)
                for p in paths]

    def save(paths, wavs):
        for path, wav in zip(paths, wavs):
            out = os.path.join(args.output_dir, os.path.splitext(os.path.basename(path))[0] + ".wav")
            save_audio(out, wav.reshape(1, -1), sample_rate=generator.output_sample_rate)

    def process(model, item):
        """one batch on `model` (the generator or its replica), on the calling thread's stream"""
        paths, cpu_wavs = item
        with torch.no_grad():
            wav_list = [w.to(device, non_blocking=True) for w in cpu_wavs]
            codes_list = model.encode(wav_list, overlap_seconds=10, device=device)["codes_list"]
            syn = model.decode(codes_list, overlap_seconds=10, device=device)["syn_wav_list"]
            return paths, [c.shape[-1] for c in codes_list], [w.cpu() for w in syn]

    pipe = None
    if device.type == "cuda" and args.in_flight > 1 and len(batches) > 1:
        from simwhisper_codec_amd.pipeline import InFlight
        pipe = InFlight(generator, args.in_flight)
    total_audio, t0 = 0.0, time.perf_counter()
    with ThreadPoolExecutor(max_workers=2) as pool:
        nxt = pool.submit(load, batches[0]) if batches else None
        pending_save, running = None, []

        def finish(fut):
            nonlocal total_audio, pending_save
            paths, clens, host = fut.result() if hasattr(fut, "result") else fut
            logging.info(f"Encoding completed, code lengths: {clens}")
            logging.info(f"Decoding completed, generated waveform lengths: {[len(w) for w in host]} samples")
            total_audio += sum(len(w) for w in host) / generator.output_sample_rate
            if pending_save is not None:
                pending_save.result()
            pending_save = pool.submit(save, paths, host)

        for bi, paths in enumerate(batches):
            logging.info(f"Processing batch {bi + 1}/{len(batches)}, files: {paths}")
            cpu_wavs = nxt.result()
            nxt = pool.submit(load, batches[bi + 1]) if bi + 1 < len(batches) else None
            logging.info(f"Successfully loaded {len(cpu_wavs)} audio files with lengths {[len(w) for w in cpu_wavs]} samples")
            if pipe is None:
                finish(process(generator, (paths, cpu_wavs)))
                continue
            running.append(pipe.submit(process, (paths, cpu_wavs)))
            if len(running) > args.in_flight:  # results are taken in submission order: output files as in the serial loop
                finish(running.pop(0))
        for fut in running:
            finish(fut)
        if pending_save is not None:
            pending_save.result()
    if pipe is not None:
        pipe.close()
    dt = time.perf_counter() - t0
    logging.info(f"All audio processing completed: {total_audio:.1f} s of audio in {dt:.2f} s "
                 f"({total_audio / max(dt, 1e-9):.1f} x real time incl. file IO)")


def main_distributed(args, world):
    """One process per GPU (torch.distributed.run): rank 0 owns the files, all ranks share the compute."""
    import torch.distributed as dist
    from simwhisper_codec_amd.dist import DataParallelCodec
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", 0))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    ngpu = torch.cuda.device_count()
    device = torch.device("cuda", local % max(ngpu, 1))
    torch.cuda.set_device(device)
    import datetime
    # a failed peer must not leave the others waiting for the default 10 minutes (failures of the local work are made
    # collective by DataParallelCodec itself; this bounds what is left: a rank that dies)
    tmo = datetime.timedelta(seconds=240)
    if args.dist_backend == "nccl":
        dist.init_process_group("nccl", device_id=device, timeout=tmo)
        comm = None
    else:
        dist.init_process_group(args.dist_backend, timeout=tmo)
        comm = "cpu"
    try:
        generator = load_model(args, device)
        dp = DataParallelCodec(generator, device, comm_device=comm)
        bs = args.batch_size * world
        if rank == 0:
            audio_paths = find_audio_files(input_dir=args.input_dir)
            os.makedirs(args.output_dir, exist_ok=True)
            logging.info(f"Processing {len(audio_paths)} audio files on {world} GPUs, output will be saved to {args.output_dir}")
            batches = [audio_paths[i:i + bs] for i in range(0, len(audio_paths), bs)]
        else:
            batches = None
        nb = dp._share_ints([len(batches)] if rank == 0 else None)[0]
        total_audio, t0 = 0.0, time.perf_counter()
        with ThreadPoolExecutor(max_workers=2) as pool, torch.no_grad():
            def load(paths):
                return [load_audio(p, target_sample_rate=generator.input_sample_rate).reshape(-1).pin_memory() for p in paths]

            def save(paths, wavs):
                for path, wav in zip(paths, wavs):
                    out = os.path.join(args.output_dir, os.path.splitext(os.path.basename(path))[0] + ".wav")
                    save_audio(out, wav.reshape(1, -1), sample_rate=generator.output_sample_rate)
            nxt = pool.submit(load, batches[0]) if rank == 0 and nb else None
            pending = None
            for bi in range(nb):
                wav_list = None
                if rank == 0:
                    logging.info(f"Processing batch {bi + 1}/{nb}, files: {batches[bi]}")
                    cpu_wavs = nxt.result()
                    nxt = pool.submit(load, batches[bi + 1]) if bi + 1 < nb else None
                    wav_list = [w.to(device, non_blocking=True) for w in cpu_wavs]
                out = dp.encode_decode(wav_list, overlap_seconds=10)
                if rank == 0:
                    logging.info(f"Decoding completed, generated waveform lengths: {[len(w) for w in out['syn_wav_list']]} samples")
                    host = [w.cpu() for w in out["syn_wav_list"]]
                    total_audio += sum(len(w) for w in host) / generator.output_sample_rate
                    if pending is not None:
                        pending.result()
                    pending = pool.submit(save, batches[bi], host)
            if pending is not None:
                pending.result()
        if rank == 0:
            dt = time.perf_counter() - t0
            logging.info(f"All audio processing completed: {total_audio:.1f} s of audio in {dt:.2f} s on {world} GPUs "
                         f"({total_audio / max(dt, 1e-9):.1f} x real time incl. file IO)")
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
