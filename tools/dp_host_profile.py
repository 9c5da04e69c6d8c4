#!/usr/bin/env python3
"""Host-side cost of rank 0's batch assembly at the 8-GPU shape (256 utterances x 10 s): cProfile of DataParallelCodec._pad_batch
(AudioCodec._stack: one gather kernel from an uploaded address list)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from common import PARAMS, state_dict
from simwhisper_codec_amd.codec import AudioCodec
m = AudioCodec(PARAMS["tiny"](), precision="mixed"); m.load_state_dict(state_dict("tiny"), strict=True); m = m.to("cuda").eval()
wavs = [torch.zeros(160000, device="cuda") for _ in range(256)]
lens = [160000] * 256
dev = torch.device("cuda", 0)
with torch.cuda.device(dev):
    for _ in range(5):
        b = m._stack(wavs, lens, dev, torch.float32, 0)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        torch.cuda.synchronize(); t0 = time.perf_counter(); b = m._stack(wavs, lens, dev, torch.float32, 0); ts.append(time.perf_counter() - t0)
    print("median _stack host time: %.0f us" % (1e6 * sorted(ts)[10]))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(20):
        b = m._stack(wavs, lens, dev, torch.float32, 0)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
