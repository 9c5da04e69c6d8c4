#!/usr/bin/env python3
"""Where the file loop of inference.py loses throughput: the same 64 batches (32 x 10 s) through InFlight(2)
  A  audio already on the GPU (bench.py's two_in_flight)           B  + host -> device staging of every batch (HostStager)
  C  B + device -> host of every result                             D  C + conversion to PCM16 and file writes on 8 threads
  E  D + file reads on 8 threads (the whole loop)"""
import os, sys, time, tempfile, shutil
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, yaml
from simwhisper_codec_amd import synth
from simwhisper_codec_amd.codec import AudioCodec
from simwhisper_codec_amd.pipeline import InFlight, HostStager
from simwhisper_codec_amd.wavio import load_audio, save_audio
from bench import bench_inputs

gp = yaml.safe_load(open(os.path.join(ROOT, "config", "SimWhisperCodec.yaml")))["generator_params"]
m = AudioCodec(gp, precision="mixed"); m.load_state_dict(synth.synth_state_dict(gp)); m = m.to("cuda:0").eval()
dev = torch.device("cuda", 0)
cpu = bench_inputs(32, 160000)
gpu = [w.to(dev) for w in cpu]
NB = 64
tmp = tempfile.mkdtemp(prefix="swc_stage_")
for i, w in enumerate(cpu):
    save_audio(os.path.join(tmp, f"in_{i}.wav"), w.reshape(1, -1), 16000)
stager = HostStager()
io = ThreadPoolExecutor(8)


def run(mode):
    def process(model, item):
        with torch.no_grad():
            wavs = gpu if mode == "A" else stager.to_device(item, dev)
            codes = model.encode(wavs, overlap_seconds=10, device=dev)["codes_list"]
            syn = model.decode(codes, overlap_seconds=10, device=dev)["syn_wav_list"]
            if mode in "AB":
                torch.cuda.current_stream().synchronize()
                return None
            return stager.to_host(syn)

    def load():
        return list(io.map(lambda i: load_audio(os.path.join(tmp, f"in_{i}.wav"), target_sample_rate=16000).reshape(-1), range(32)))

    def save(host):
        list(io.map(lambda iw: save_audio(os.path.join(tmp, f"out_{iw[0]}.wav"), iw[1].reshape(1, -1), 16000), enumerate(host)))

    with InFlight(m, 2) as pipe, ThreadPoolExecutor(2) as ctl:
        for _ in range(4):
            pipe.submit(process, cpu).result()
        t0 = time.perf_counter()
        running, saves = [], []
        nxt = ctl.submit(load) if mode == "E" else None
        for b in range(NB):
            item = cpu
            if mode == "E":
                item = nxt.result()
                nxt = ctl.submit(load)
            running.append(pipe.submit(process, item))
            if len(running) > 2:
                host = running.pop(0).result()
                if mode in "DE":
                    saves.append(ctl.submit(save, host))
        for f in running:
            host = f.result()
            if mode in "DE":
                saves.append(ctl.submit(save, host))
        for s in saves:
            s.result()
        dt = time.perf_counter() - t0
    print(f"{mode}: {NB * 320 / dt:9.1f} audio-s/s  ({dt / NB * 1e3:.2f} ms per batch)", flush=True)


for rep in range(2):
    for mode in "ABCDE":
        run(mode)
shutil.rmtree(tmp, ignore_errors=True)
