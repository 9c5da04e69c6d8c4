#!/usr/bin/env python3
"""Structural guard on the ISA of the hot swc_gemm kernels.

hipcc's schedule of the K loop depends on code far away from it: a never-taken branch in the staging lambda is worth 6.5 %
of the whole step (profiles/r02_ab_sched_region.txt), a multiply by 1.0 in the epilogue 11 % of the split-f16 fc1 launch
(profiles/r03_gemm_epilogue_ab.txt).  A toolchain bump or an innocent edit can therefore change the kernels silently.
This script extracts the gfx950 code objects from the built libswc_hip.so (clang offload bundles in .hip_fatbin),
disassembles the kernels the metric runs and asserts the shape of their K loop:

  * no scratch (spill) instruction anywhere in the kernel;
  * the inner loop holds exactly the expected MFMAs, fragment reads (ds_read_b128) and LDS-DMA instructions of one K slice;
  * one s_barrier and one `s_waitcnt vmcnt(0)` per slice;
  * the LDS-DMA issue block of slice t+1 is one block in front of slice t's body, not interleaved with its MFMAs;
  * no fragment read sits between the slice's barrier and its last MFMA (only register-operand MFMAs follow the barrier);
  * at most half of the slice's MFMAs sit behind its barrier (the 11 % slower schedule had 59 of 96 there, and a second wait).

usage: python tools/check_isa.py [path/to/libswc_hip.so]      (exit code 1 on a violated expectation)
A PERFORMANCE lint, not a correctness check: __graft_entry__.build() prints its findings and goes on (SWC_STRICT_ISA=1 makes
them fatal there); tests/test_host_cpu.py::test_hot_kernel_isa_shape holds the shipped build to it.
Needs /opt/rocm/lib/llvm/bin/llvm-objdump; skipped with a note if that is missing.
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"

# kernel (demangled template arguments) -> expectations of ONE K slice of the inner loop
#   MODE 1 bf16 (BK 64: MT*4 tiles x 2 k-groups), MODE 2 split-f16 (BK 32: MT*4 tiles x 3 products)
EXPECT = {
    # name fragment                                  mfma  ds_read  dma   what it is
    # (split-f16: the number of MFMAs behind the slice fence is set in the source — 3 row blocks of 12 at 256 rows, 1 at 192)
    "gemm_kernelILi2E6f16s_tLi8ELi2ELi4ELb1ELi128ELi1E": (96, 24, 8, "split-f16 fc1 + GELU (256 x 256)", 36),
    "gemm_kernelILi2E6f16s_tLi6ELi2ELi4ELb1ELi128ELi0E": (72, 20, 7, "split-f16 qkv (192 x 256)", 12),
    "gemm_kernelILi2EfLi6ELi2ELi4ELb1ELi128ELi2E": (72, 20, 7, "split-f16 out-proj / fc2 (192 x 256, f32 out)", 12),
    "gemm_kernelILi1EtLi8ELi2ELi4ELb1ELi128ELi1E": (64, 24, 8, "bf16 fc1 / pwconv1 + GELU (256 x 256)"),
    "gemm_kernelILi1EtLi6ELi2ELi4ELb1ELi128ELi0E": (48, 20, 7, "bf16 qkv (192 x 256)"),
    "gemm_kernelILi1EfLi6ELi2ELi4ELb1ELi128ELi2E": (48, 20, 7, "bf16 out-proj / fc2 / heads (192 x 256, f32 out)"),
}


def code_objects(path):
    """gfx950 code objects of every clang offload bundle in the file"""
    data = open(path, "rb").read()
    out = []
    for m in re.finditer(re.escape(MAGIC), data):
        base = m.start()
        n, = struct.unpack_from("<Q", data, base + len(MAGIC))
        pos = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, pos)
            triple = data[pos + 24:pos + 24 + tlen].decode()
            pos += 24 + tlen
            if "gfx950" in triple and size:
                out.append(data[base + off:base + off + size])
    return out


def disassemble(blob):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(blob)
        f.flush()
        r = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    return r.stdout


def functions(asm):
    cur, body = None, []
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            if cur:
                yield cur, body
            cur, body = m.group(1), []
        elif cur and line.strip():
            body.append(line.strip())
    if cur:
        yield cur, body


def check(path):
    if not os.path.exists(OBJDUMP):
        print(f"check_isa: {OBJDUMP} not found, skipped")
        return 0
    found, bad = {}, []
    for blob in code_objects(path):
        asm = disassemble(blob)
        for name, body in functions(asm):
            for frag, exp in EXPECT.items():
                if frag in name:
                    found[frag] = (name, body, exp)
    for frag, exp in EXPECT.items():
        if frag not in found:
            # a renamed template parameter or a toolchain that mangles differently: nothing can be said about the kernel
            # (a note, not a failure; the numerics tests do not depend on this lint)
            print(f"check_isa: note: {exp[3]}: kernel {frag} not found in {path} (not checked)")
            continue
        name, body, exp = found[frag]
        n_mfma, n_read, n_dma, what = exp[:4]
        n_behind = exp[4] if len(exp) > 4 else None
        body = [ln for ln in body if re.search(r"//\s*[0-9A-Fa-f]+:", ln)]  # instructions only (no labels / padding notes)
        text = [re.sub(r"\s*//.*$", "", ln) for ln in body]
        if any(t.startswith("scratch_") for t in text):
            bad.append(f"{what}: scratch (spill) instructions in the kernel")
        # instruction addresses and branch targets (llvm-objdump: `// <address>: <encoding> <function+0xoffset>`)
        addr = [int(re.search(r"//\s*([0-9A-Fa-f]+):", ln).group(1), 16) for ln in body]
        start = addr[0]
        loops = []
        for i, ln in enumerate(body):
            if text[i].startswith(("s_cbranch", "s_branch")):
                m = re.search(r"\+0x([0-9A-Fa-f]+)>", ln)
                tgt = start + (int(m.group(1), 16) if m else 0)
                if tgt <= addr[i]:  # backward edge: [target, branch] is a loop
                    j = next(k for k, a_ in enumerate(addr) if a_ >= tgt)
                    loops.append((i - j, j, i))
        # the K loop: the innermost loop (smallest range) that holds one slice's MFMAs and a workgroup barrier
        loop = None
        for _, j, i in sorted(loops):
            seg = text[j:i + 1]
            if sum(t.startswith("v_mfma") for t in seg) >= n_mfma and any(t.startswith("s_barrier") for t in seg):
                # the loop has two back edges to the same body: from its header (no further slice to stage) and from the
                # DMA block behind the header; take the widest range with this target
                i = max(i2 for _, j2, i2 in loops if j2 == j)
                loop = text[j:i + 1]
                break
        if loop is None:
            bad.append(f"{what}: no K loop found (a backward branch around barrier + MFMAs)")
            continue
        nm = sum(t.startswith("v_mfma") for t in loop)
        nr = sum(t.startswith("ds_read_b128") for t in loop)
        nd = sum(t.startswith("global_load_lds") for t in loop)
        mf = [i for i, t in enumerate(loop) if t.startswith("v_mfma")]
        dm = [i for i, t in enumerate(loop) if t.startswith("global_load_lds")]
        # the loop is laid out body first, header + DMA block last: in execution order the DMA block of slice t+1 runs
        # right before the body of slice t.  Either way no DMA may sit between two MFMAs
        dma_split = bool(dm) and not (min(dm) > max(mf) or max(dm) < min(mf))
        bar = next(i for i, t in enumerate(loop) if t.startswith("s_barrier"))
        reads_after_bar = sum(t.startswith("ds_read") for t in loop[bar:max(mf) + 1])
        vm0 = sum(bool(re.match(r"s_waitcnt vmcnt\(0\)", t)) for t in loop)
        msgs = []
        if nm != n_mfma:
            msgs.append(f"{nm} MFMAs per slice, expected {n_mfma}")
        if nr != n_read:
            msgs.append(f"{nr} ds_read_b128 per slice, expected {n_read}")
        if nd != n_dma:
            msgs.append(f"{nd} LDS-DMA per slice, expected {n_dma}")
        if dma_split:
            msgs.append("LDS-DMA instructions of the next slice are interleaved with this slice's MFMAs")
        if reads_after_bar:
            msgs.append(f"{reads_after_bar} fragment reads behind the slice barrier")
        if vm0 != 1:
            msgs.append(f"{vm0} `s_waitcnt vmcnt(0)` per slice, expected 1")
        behind = sum(t.startswith("v_mfma") for t in loop[bar:])
        if n_behind is not None and behind != n_behind:
            msgs.append(f"{behind} MFMAs behind the slice fence, the source asks for {n_behind}")
        if 2 * behind > nm:   # the slow schedule of r03 (fc1 226 us instead of 204) had 59 of 96 behind the barrier and two waits
            msgs.append(f"{behind} of {nm} MFMAs sit behind the slice barrier (more than half: the schedule that measured 11 % slower)")
        for m_ in msgs:
            bad.append(f"{what}: {m_}")
        print(f"check_isa: {what:52s} slice = {nd} DMA -> {nr} reads / {nm} MFMAs ({sum(t.startswith('v_mfma') for t in loop[bar:])} behind the barrier)"
              + ("" if not msgs else "   <-- " + "; ".join(msgs)))
    for b in bad:
        print("check_isa: FAIL", b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(check(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "simwhisper_codec_amd", "libswc_hip.so")))
