#!/usr/bin/env python3
"""End-to-end rate of the reference's CLI surface (inference.py: files in -> codec round trip -> files out)
on synthetic WAV files, beside the rate of the same batches with the audio already on the GPU (what bench.py's `value` is).

usage: python tools/cli_throughput.py [n_files=512] [seconds=10] [batch_size=32] [in_flight=2]
Writes the inputs (PCM16, 16 kHz, seed-1234 signals of bench.py) and the outputs under a temporary directory
(SWC_CLI_TMP or /tmp) and removes them.
"""
import logging
import os
import re
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import inference  # noqa: E402
from bench import bench_inputs  # noqa: E402
from simwhisper_codec_amd.wavio import save_audio  # noqa: E402

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 512
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
bs = int(sys.argv[3]) if len(sys.argv) > 3 else 32
in_flight = int(sys.argv[4]) if len(sys.argv) > 4 else 2
extra = sys.argv[5:]                      # further inference.py flags, e.g. --io_threads 4
if os.environ.get("SWC_SWITCH_INTERVAL"):
    sys.setswitchinterval(float(os.environ["SWC_SWITCH_INTERVAL"]))

tmp = tempfile.mkdtemp(prefix="swc_cli_", dir=os.environ.get("SWC_CLI_TMP", "/tmp"))
try:
    src, dst = os.path.join(tmp, "in"), os.path.join(tmp, "out")
    os.makedirs(src)
    wavs = bench_inputs(32, int(secs * 16000))
    t0 = time.perf_counter()
    for i in range(n_files):
        save_audio(os.path.join(src, f"utt_{i:05d}.wav"), wavs[i % 32].reshape(1, -1), sample_rate=16000)
    print(f"wrote {n_files} x {secs:g} s PCM16 files in {time.perf_counter() - t0:.2f} s", flush=True)

    class Grab(logging.Handler):
        last = None

        def emit(self, rec):
            m = re.search(r"([0-9.]+) s of audio in ([0-9.]+) s", rec.getMessage())
            if m:
                Grab.last = (float(m.group(1)), float(m.group(2)))
            if "stage wall seconds" in rec.getMessage():
                print("   ", rec.getMessage(), flush=True)
    logging.getLogger().addHandler(Grab())
    logging.getLogger().setLevel(logging.WARNING)
    argv = ["--config_path", os.path.join(ROOT, "config", "SimWhisperCodec.yaml"), "--synthetic_checkpoint", "--device", "cuda",
            "--batch_size", str(bs), "--input_dir", src, "--output_dir", dst, "--in_flight", str(in_flight)] + extra
    inference.set_logging = lambda *a, **k: None   # keep the per-batch INFO lines out of the report
    for rep in range(2):                           # first pass: model build, library load, page cache
        shutil.rmtree(dst, ignore_errors=True)
        logging.getLogger().setLevel(logging.INFO)
        logging.getLogger().handlers = [h for h in logging.getLogger().handlers if isinstance(h, Grab)]
        t0 = time.perf_counter()
        inference.main(argv)
        wall = time.perf_counter() - t0
        audio, loop = Grab.last
        print(f"pass {rep}: {audio:.0f} s of audio, file loop {loop:.2f} s = {audio / loop:8.1f} audio-s/s incl. file IO "
              f"(whole call incl. model load {wall:.2f} s)", flush=True)
    assert len(os.listdir(dst)) == n_files
finally:
    shutil.rmtree(tmp, ignore_errors=True)
