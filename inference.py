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
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from audiocodec.model import AudioCodec  # noqa: E402
from simwhisper_codec_amd.pipeline import HostStager  # noqa: E402
from simwhisper_codec_amd.wavio import find_audio_files, load_audio, read_pcm16, save_audio, save_pcm16  # noqa: E402


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
    p.add_argument("--precision", type=str, default="mixed", choices=["fp32", "mixed", "mixed_f32", "bf16", "fp8", "fp8_fc1", "f16s"])
    p.add_argument("--synthetic_checkpoint", action="store_true",
                   help="ignore --checkpoint_path and use the closed-form synthetic weights (offline testing)")
    p.add_argument("--in_flight", type=int, default=2,
                   help="batches in flight on the GPU (simwhisper_codec_amd.pipeline.InFlight: consecutive batches overlap on "
                        "two streams; same output files, about 7 %% more throughput; 1 = one batch at a time)")
    p.add_argument("--io_threads", type=int, default=8,
                   help="threads that read and write audio files (the files of a batch are independent; one thread writes "
                        "about 1 200 ten-second files per second, less than one GPU produces)")
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

    io = ThreadPoolExecutor(max_workers=max(1, args.io_threads))

    stager = HostStager()

    on_gpu = device.type == "cuda"

    def load_one(path):
        # a mono PCM16 file at the model's rate goes to the GPU as 16-bit samples (converted there: the same values);
        # everything else (other widths, channels, rates, FLAC) is decoded to f32 here
        pcm = read_pcm16(path, generator.input_sample_rate) if on_gpu else None
        return pcm if pcm is not None else load_audio(path, target_sample_rate=generator.input_sample_rate).reshape(-1)

    def save_one(item):
        path, wav = item
        out = os.path.join(args.output_dir, os.path.splitext(os.path.basename(path))[0] + ".wav")
        if wav.dtype == torch.int16:
            save_pcm16(out, wav, sample_rate=generator.output_sample_rate)
        else:
            save_audio(out, wav.reshape(1, -1), sample_rate=generator.output_sample_rate)

    def stage_in(cpu_wavs):
        if not on_gpu:
            return cpu_wavs
        if all(w.dtype == torch.int16 for w in cpu_wavs):
            return stager.to_device_pcm16(cpu_wavs, device)
        return stager.to_device([w if w.dtype == torch.float32 else w.to(torch.float32) * (1.0 / 32768.0) for w in cpu_wavs], device)

    # wall seconds per stage, summed over the threads that run them (they overlap).  The launch threads (process) do nothing but
    # launch: reading + staging runs in the loader thread, the device -> host copy + writing in the saver thread
    spent = {"load+h2d": 0.0, "encode+decode": 0.0, "d2h+save": 0.0}
    spent_lock = threading.Lock()

    def account(stage, seconds):
        with spent_lock:
            spent[stage] += seconds

    def load(paths):
        t = time.perf_counter()
        wavs = stage_in(list(io.map(load_one, paths)))
        if on_gpu:
            torch.cuda.current_stream(device).synchronize()   # the staging buffer is re-used by the next batch
        account("load+h2d", time.perf_counter() - t)
        return wavs

    def save(paths, wavs):
        t = time.perf_counter()
        host = stager.to_host(wavs) if on_gpu else wavs        # (the batch's stream was synchronised before it was handed back)
        list(io.map(save_one, zip(paths, host)))
        account("d2h+save", time.perf_counter() - t)

    def process(model, item):
        """one batch on `model` (the generator or its replica), on the calling thread's stream"""
        paths, wav_list = item
        with torch.no_grad():
            t1 = time.perf_counter()
            def round_trip():
                codes = model.encode(wav_list, overlap_seconds=10, device=device)["codes_list"]
                return codes, model.decode(codes, overlap_seconds=10, device=device)["syn_wav_list"]
            # the range check of the split-f16 encoder is read from a snapshot behind the encode kernels once the decode has been
            # enqueued (no stall between the two calls); a clipped batch is redone on exact-f32 operands
            defer = getattr(model, "deferred_range_check", None)
            if defer is None or not on_gpu:
                codes_list, syn = round_trip()
            else:
                with defer() as chk:
                    codes_list, syn = round_trip()
                if chk.clipped:
                    codes_list, syn = round_trip()
            out = stager.pcm16_on_device(syn) if on_gpu else [w.cpu() for w in syn]
            if on_gpu:
                torch.cuda.current_stream().synchronize()
            account("encode+decode", time.perf_counter() - t1)
            return paths, [c.shape[-1] for c in codes_list], out

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
    io.shutdown()
    dt = time.perf_counter() - t0
    logging.info(f"All audio processing completed: {total_audio:.1f} s of audio in {dt:.2f} s "
                 f"({total_audio / max(dt, 1e-9):.1f} x real time incl. file IO)")
    logging.info("stage wall seconds (overlapping threads): " + ", ".join(f"{k} {v:.2f}" for k, v in spent.items()))


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
        with ThreadPoolExecutor(max_workers=2) as pool, ThreadPoolExecutor(max_workers=max(1, args.io_threads)) as io, \
                torch.no_grad():
            stager = HostStager()

            def load_one(path):
                pcm = read_pcm16(path, generator.input_sample_rate)
                return pcm if pcm is not None else load_audio(path, target_sample_rate=generator.input_sample_rate).reshape(-1)

            def save_one(item):
                path, wav = item
                out = os.path.join(args.output_dir, os.path.splitext(os.path.basename(path))[0] + ".wav")
                save_pcm16(out, wav, sample_rate=generator.output_sample_rate)

            def stage_in(cpu_wavs):
                if all(w.dtype == torch.int16 for w in cpu_wavs):
                    return stager.to_device_pcm16(cpu_wavs, device)
                return stager.to_device([w if w.dtype == torch.float32 else w.to(torch.float32) * (1.0 / 32768.0)
                                         for w in cpu_wavs], device)

            def load(paths):
                return list(io.map(load_one, paths))

            def save(paths, wavs):
                list(io.map(save_one, zip(paths, wavs)))
            nxt = pool.submit(load, batches[0]) if rank == 0 and nb else None
            pending = None
            for bi in range(nb):
                wav_list = None
                if rank == 0:
                    logging.info(f"Processing batch {bi + 1}/{nb}, files: {batches[bi]}")
                    try:
                        cpu_wavs = nxt.result()
                        nxt = pool.submit(load, batches[bi + 1]) if bi + 1 < nb else None
                        wav_list = stage_in(cpu_wavs)
                    except Exception as e:
                        # an unreadable / corrupt file fails on rank 0 alone, outside every collective: tell the other ranks
                        # (they sit in the lengths broadcast of this step) so that every rank stops now, as the single-GPU
                        # loop and the reference do, instead of at the process-group timeout
                        logging.error(f"batch {bi + 1}/{nb}: cannot load {batches[bi]}: {type(e).__name__}: {e}")
                        dp.abort(e)
                        raise
                out = dp.encode_decode(wav_list, overlap_seconds=10)
                if rank == 0:
                    logging.info(f"Decoding completed, generated waveform lengths: {[len(w) for w in out['syn_wav_list']]} samples")
                    host = stager.to_host(stager.pcm16_on_device(out["syn_wav_list"]))
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
