#!/usr/bin/env python3
"""Build libswc_<tag>.so from the current sources with extra hipcc flags for chosen files (same-box A/B of compiler flags):
    tools/build_full_variant.py <tag> [file.hip=flag,flag ...] ...
e.g. tools/build_full_variant.py ilpall swc_gemm.hip=-mllvm,-amdgpu-sched-strategy=max-ilp swc_attention16.hip=-mllvm,-amdgpu-sched-strategy=max-ilp
Files without an entry reuse the product build's object (simwhisper_codec_amd/build/*.o).  The product library is never touched."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simwhisper_codec_amd import build as b
tag = sys.argv[1]
extra = dict(a.split("=", 1) for a in sys.argv[2:])
objs, procs = [], []
for src in b.SOURCES:
    obj = os.path.join(b.HERE, "build", src.replace(".hip", ".o"))
    if src in extra:
        obj = f"/tmp/{tag}_{src.replace('.hip', '.o')}"
        cmd = [b._hipcc(), f"--offload-arch={b.ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-DSWC_TUNING", "-I",
               os.path.join(ROOT, "include"), "-I", b.CSRC] + b.EXTRA_FLAGS.get(src, []) + [f for f in extra[src].split(",") if f] + \
              ["-c", os.path.join(b.CSRC, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    objs.append(obj)
for src, p in procs:
    out, _ = p.communicate()
    if p.returncode:
        sys.exit(f"{src}: {out}")
lib = os.path.join(b.HERE, f"libswc_{tag}.so")
subprocess.check_call([b._hipcc(), f"--offload-arch={b.ARCH}", "-shared", "-fPIC", "-o", lib] + objs)
print("built", lib)
