"""Test infrastructure: a small FLAC ENCODER (RFC 9639 subset, numpy + Python ints) that writes the streams the native
decoder (simwhisper_codec_amd/csrc/swc_flac.c) is tested on.  No FLAC tool or library exists in the build environment, so
the streams are made here: every subframe type (CONSTANT, VERBATIM, FIXED 0-4, LPC), Rice and Rice2 residual coding with
partition orders and escape partitions, wasted bits, the four channel assignments, explicit block sizes and a short last
block, CRC-8 / CRC-16, and the MD5 signature (hashlib) in STREAMINFO."""
import hashlib

import numpy as np


class Bits:
    def __init__(self):
        self.v, self.n = 0, 0

    def put(self, value, nbits):
        if nbits:
            self.v = (self.v << nbits) | (int(value) & ((1 << nbits) - 1))
            self.n += nbits

    def unary(self, q):
        self.put(1, q + 1)  # q zeros then a one

    def align(self):
        if self.n % 8:
            self.put(0, 8 - self.n % 8)

    def bytes(self):
        assert self.n % 8 == 0
        return self.v.to_bytes(self.n // 8, "big") if self.n else b""


def crc8(data):
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    return c


def crc16(data):
    c = 0
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
    return c


def utf8_number(v):
    """the frame header's number in the UTF-8-like coding (1 to 7 bytes, up to 36 bits)"""
    if v < 0x80:
        return bytes([v])
    nbytes = next(k for k, lim in ((2, 1 << 11), (3, 1 << 16), (4, 1 << 21), (5, 1 << 26), (6, 1 << 31), (7, 1 << 36)) if v < lim)
    cont = [0x80 | ((v >> (6 * i)) & 0x3F) for i in range(nbytes - 1)][::-1]
    lead = ((0xFF << (8 - nbytes)) & 0xFF) | (v >> (6 * (nbytes - 1)))
    return bytes([lead] + cont)


def zigzag(r):
    return (r << 1) if r >= 0 else ((-r) << 1) - 1


def put_residual(bw, res, bs, order, porder, rice2=False, escape_part=None):
    pbits, esc = (5, 31) if rice2 else (4, 15)
    bw.put(1 if rice2 else 0, 2)
    bw.put(porder, 4)
    parts = 1 << porder
    idx = 0
    for p in range(parts):
        cnt = (bs >> porder) - (order if p == 0 else 0)
        seg = res[idx:idx + cnt]
        idx += cnt
        raw = max([int(abs(int(x))).bit_length() + 1 for x in seg] + [1])
        if escape_part == p and raw <= 31:
            bw.put(esc, pbits)
            bw.put(raw, 5)
            for x in seg:
                bw.put(int(x), raw)
            continue
        us = [zigzag(int(x)) for x in seg]
        mean = (sum(us) / max(len(us), 1)) if us else 0
        k = max(0, min(esc - 1, int(np.log2(mean + 1))))
        bw.put(k, pbits)
        for u in us:
            bw.unary(u >> k)
            bw.put(u & ((1 << k) - 1), k)


FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def put_subframe(bw, s, bps, kind, porder=0, rice2=False, escape_part=None, wasted=0, lpc=None):
    """s: list of Python ints.  kind: 'constant' | 'verbatim' | ('fixed', order) | ('lpc', order)."""
    bs = len(s)
    if wasted:
        assert all(x % (1 << wasted) == 0 for x in s)
        s = [x >> wasted for x in s]
        bps -= wasted
    bw.put(0, 1)
    if kind == "constant":
        bw.put(0, 6)
    elif kind == "verbatim":
        bw.put(1, 6)
    elif kind[0] == "fixed":
        bw.put(8 + kind[1], 6)
    else:
        bw.put(32 + kind[1] - 1, 6)
    if wasted:
        bw.put(1, 1)
        bw.unary(wasted - 1)
    else:
        bw.put(0, 1)
    if kind == "constant":
        bw.put(s[0], bps)
    elif kind == "verbatim":
        for x in s:
            bw.put(x, bps)
    elif kind[0] == "fixed":
        order = kind[1]
        for x in s[:order]:
            bw.put(x, bps)
        c = FIXED[order]
        res = [s[i] - sum(c[j] * s[i - 1 - j] for j in range(order)) for i in range(order, bs)]
        put_residual(bw, res, bs, order, porder, rice2, escape_part)
    else:
        order = kind[1]
        coef, prec, shift = lpc
        for x in s[:order]:
            bw.put(x, bps)
        bw.put(prec - 1, 4)
        bw.put(shift, 5)
        for cq in coef:
            bw.put(cq, prec)
        res = [s[i] - (sum(coef[j] * s[i - 1 - j] for j in range(order)) >> shift) for i in range(order, bs)]
        put_residual(bw, res, bs, order, porder, rice2, escape_part)


def lpc_coefficients(x, order, prec=12):
    """least-squares predictor of the block, quantised to `prec`-bit signed coefficients and a shift >= 0"""
    x = np.asarray(x, dtype=np.float64)
    rows = np.stack([x[order - 1 - j:len(x) - 1 - j] for j in range(order)], axis=1)
    c, *_ = np.linalg.lstsq(rows, x[order:], rcond=None)
    m = max(float(np.abs(c).max()), 1e-9)
    shift = max(0, min(15, prec - 2 - int(np.ceil(np.log2(m + 1e-12)))))
    q = np.clip(np.round(c * (1 << shift)), -(1 << (prec - 1)), (1 << (prec - 1)) - 1).astype(np.int64)
    return [int(v) for v in q], prec, shift


BS_CODES = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13, 16384: 14, 32768: 15}
SR_CODES = {88200: 1, 176400: 2, 192000: 3, 8000: 4, 16000: 5, 22050: 6, 24000: 7, 32000: 8, 44100: 9, 48000: 10, 96000: 11}
BPS_CODES = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6, 32: 7}


def encode(samples, sr, bps, blocksize=1024, plan=None, md5=True, id3=False):
    """samples: int array (n, channels).  plan(frame_index, channel) -> dict(kind=..., porder=..., rice2=..., escape_part=...,
    wasted=...) and plan(frame_index, None) -> channel assignment 0 (independent) / 8 / 9 / 10 for stereo."""
    x = np.asarray(samples, dtype=np.int64)
    n, ch = x.shape
    pcm = bytearray()
    nb = (bps + 7) // 8
    for row in x:
        for v in row:
            pcm += int(v).to_bytes(nb, "little", signed=True)
    sig = hashlib.md5(bytes(pcm)).digest() if md5 else bytes(16)
    si = Bits()
    si.put(blocksize, 16); si.put(blocksize, 16); si.put(0, 24); si.put(0, 24)
    si.put(sr, 20); si.put(ch - 1, 3); si.put(bps - 1, 5); si.put(n, 36)
    out = bytearray()
    if id3:
        out += b"ID3\x04\x00\x00" + bytes([0, 0, 0, 10]) + bytes(10)
    out += b"fLaC" + bytes([0x80 | 0]) + (34).to_bytes(3, "big") + si.bytes() + sig
    fi = 0
    for start in range(0, n, blocksize):
        blk = x[start:start + blocksize]
        bs = len(blk)
        ca = plan(fi, None) if (plan and ch == 2) else (ch - 1)
        ca_code = ca if ca >= 8 else ch - 1
        hdr = Bits()
        hdr.put(0xFFF8, 16)
        if bs in BS_CODES:
            bcode = BS_CODES[bs]
        else:
            bcode = 6 if bs <= 256 else 7
        scode = SR_CODES.get(sr, 0)
        hdr.put(bcode, 4); hdr.put(scode, 4)
        hdr.put(ca_code, 4); hdr.put(BPS_CODES.get(bps, 0), 3); hdr.put(0, 1)
        hb = hdr.bytes() + utf8_number(fi)
        if bcode == 6:
            hb += bytes([bs - 1])
        elif bcode == 7:
            hb += (bs - 1).to_bytes(2, "big")
        hb += bytes([crc8(hb)])
        chans = [blk[:, c].tolist() for c in range(ch)]
        widths = [bps] * ch
        if ca_code == 8:
            chans = [chans[0], [a - b for a, b in zip(chans[0], chans[1])]]; widths = [bps, bps + 1]
        elif ca_code == 9:
            chans = [[a - b for a, b in zip(chans[0], chans[1])], chans[1]]; widths = [bps + 1, bps]
        elif ca_code == 10:
            chans = [[(a + b) >> 1 for a, b in zip(chans[0], chans[1])], [a - b for a, b in zip(chans[0], chans[1])]]
            widths = [bps, bps + 1]
        body = Bits()
        for c in range(ch):
            spec = dict(kind=("fixed", 2), porder=0, rice2=False, escape_part=None, wasted=0)
            if plan:
                spec.update(plan(fi, c) or {})
            s = chans[c]
            kind = spec["kind"]
            if kind == "constant" and len(set(s)) != 1:
                kind = "verbatim"
            order = 0 if kind in ("constant", "verbatim") else kind[1]
            if order > bs or (bs >> spec["porder"]) < order or bs % (1 << spec["porder"]):
                kind, order = "verbatim", 0
            lpc = None
            if kind not in ("constant", "verbatim") and kind[0] == "lpc":
                sw = [v >> spec["wasted"] for v in s] if spec["wasted"] else s
                lpc = lpc_coefficients(sw, order) if len(sw) > 2 * order + 2 else ([0] * order, 12, 0)
            put_subframe(body, s, widths[c], kind, spec["porder"], spec["rice2"], spec["escape_part"], spec["wasted"], lpc)
        body.align()
        frame = hb + body.bytes()
        out += frame + crc16(frame).to_bytes(2, "big")
        fi += 1
    return bytes(out)
