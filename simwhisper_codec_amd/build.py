"""Build libswc_hip.so (gfx950) in-tree with hipcc.

`python -m simwhisper_codec_amd.build` or `build_library()`; the .so lands next to
this file so that it travels with the source tree (git-ignored, not gpurun-ignored).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(HERE, "libswc_hip.so")
SOURCES = ["swc_api.hip", "swc_gemm.hip", "swc_attention.hip", "swc_attention16.hip", "swc_pointwise.hip", "swc_convnext.hip", "swc_mlp.hip", "swc_convnext64.hip", "swc_projln.hip"]
ARCH = "gfx950"
# per-file flags.  -fno-slp-vectorize: hipcc otherwise packs adjacent f32 mul/add/fma into v_pk_*_f32, which issue at
# half rate on gfx950 and cost extra v_mov shuffles — slower beside MFMAs (softmax, epilogues)
# -mllvm -amdgpu-sched-strategy=max-ilp: LLVM's max-ILP machine scheduler instead of the default (occupancy-driven) one for
# the two MFMA-loop files: same instructions, another order (results bit-identical); same-box A/B of the whole step
# 22.19 -> 21.90 ms (swc_gemm.hip) -> 21.80 ms (+ swc_convnext.hip), split-f16 GEMMs -2.5 ... -3.4 %, ConvNeXt block 254.8 ->
# 249.9 us; swc_attention16.hip gets slower with it (99.9 -> 102.9 us) and keeps the default (profiles/r03_sched_strategy_ab.txt)
_MAX_ILP = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
EXTRA_FLAGS = {"swc_attention16.hip": ["-fno-slp-vectorize"], "swc_convnext.hip": ["-fno-slp-vectorize"] + _MAX_ILP,
               "swc_mlp.hip": ["-fno-slp-vectorize"] + _MAX_ILP, "swc_convnext64.hip": ["-fno-slp-vectorize"] + _MAX_ILP,
               "swc_projln.hip": ["-fno-slp-vectorize"] + _MAX_ILP,
               "swc_gemm.hip": _MAX_ILP}  # (-fno-slp-vectorize measured slower for swc_gemm.hip: -10 %)


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))] + [os.path.join(ROOT, "include", "swc.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link libswc_hip.so. Returns its path."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    common = [
        hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC",
        "-ffp-contract=on",  # contraction only inside one expression; parity code uses __f*_rn
        "-I", os.path.join(ROOT, "include"), "-I", CSRC,
    ]
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = common + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose:
            print(out)
    tmp = LIB_PATH + ".tmp"
    link = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", tmp] + objs
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    os.replace(tmp, LIB_PATH)
    stamp_commit()
    return LIB_PATH


IO_LIB_PATH = os.path.join(HERE, "libswc_io.so")
IO_SOURCES = ["swc_flac.c"]


def build_io_library(force=False):
    """Host-side helper library (plain C, gcc): the FLAC decoder behind wavio.load_audio.  No GPU code, no dependency."""
    srcs = [os.path.join(CSRC, f) for f in IO_SOURCES]
    if not force and os.path.exists(IO_LIB_PATH) and all(os.path.getmtime(f) <= os.path.getmtime(IO_LIB_PATH) for f in srcs):
        return IO_LIB_PATH
    cc = next((c for c in (os.environ.get("CC"), shutil.which("gcc"), shutil.which("cc"), shutil.which("clang")) if c), None)
    if cc is None:
        raise RuntimeError("no C compiler found for libswc_io.so (set CC)")
    tmp = IO_LIB_PATH + ".tmp"
    r = subprocess.run([cc, "-O2", "-std=c99", "-Wall", "-shared", "-fPIC", "-o", tmp] + srcs,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"building libswc_io.so failed:\n{r.stdout}")
    os.replace(tmp, IO_LIB_PATH)
    return IO_LIB_PATH


def stamp_commit():
    """The GPU box receives a snapshot without .git: leave the commit the library was built from next to it, so that
    bench.py / profile summaries taken there can name it (the file is git-ignored and travels with the .so)."""
    try:
        r = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                           text=True, timeout=5)
        dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--untracked-files=no"], stdout=subprocess.PIPE,
                               stderr=subprocess.DEVNULL, text=True, timeout=5).stdout.strip()
        if r.returncode == 0 and r.stdout.strip():
            with open(os.path.join(HERE, "_build_commit.txt"), "w") as f:
                f.write(r.stdout.strip() + ("+dirty" if dirty else "") + "\n")
    except Exception:
        pass


if __name__ == "__main__":
    path = build_library(force="--force" in sys.argv, verbose="-v" in sys.argv)
    print(path)
    print(build_io_library(force="--force" in sys.argv))
