#!/usr/bin/env python3
"""Host-side copy rates that bound the file loop of inference.py: pageable -> pinned memcpy, pin_memory(), H2D / D2H from pageable
and pinned memory, for one 10 s utterance (640 KB) and one batch of 32 (20 MB)."""
import time
import torch

def t(f, n=20):
    f(); torch.cuda.synchronize()
    s = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - s) / n * 1e3

for n, what in ((160000, "one 10 s utterance (640 KB)"), (32 * 160000, "32 utterances (20.5 MB)")):
    src = torch.rand(n)
    pin = torch.empty(n).pin_memory()
    page = torch.empty(n)
    dev = torch.empty(n, device="cuda")
    mb = n * 4 / 1e6
    rows = [
        ("pageable -> pageable copy_", lambda: page.copy_(src)),
        ("pageable -> pinned copy_", lambda: pin.copy_(src)),
        ("pinned -> pageable copy_", lambda: page.copy_(pin)),
        ("pin_memory() (allocate + copy)", lambda: src.pin_memory()),
        ("H2D from pageable (.to)", lambda: src.to("cuda")),
        ("H2D from pinned, non_blocking + sync", lambda: dev.copy_(pin, non_blocking=True)),
        ("D2H to pageable (.cpu())", lambda: dev.cpu()),
        ("D2H to pinned, non_blocking + sync", lambda: pin.copy_(dev, non_blocking=True)),
    ]
    print(what)
    for name, f in rows:
        ms = t(f)
        print(f"  {name:40s} {ms:8.3f} ms  {mb / ms:8.2f} GB/s")
