"""CPU ORACLE (test infrastructure) for the SWC1 code bitstream: a numpy restatement of the format in
simwhisper_codec_amd/bitstream.py.  The reference has no packed format (codes stay in memory, model.py:302),
so this format is pinned by its own specification, the round-trip property and this independent implementation
— "parity unpinned" with respect to the reference by construction."""
import numpy as np


def pack(codes):
    """codes (8, T) ints < 2048 -> bytes (11 * T,) uint8."""
    codes = np.asarray(codes, dtype=np.uint64)
    G, T = codes.shape
    assert G == 8
    word = np.zeros(T, dtype=object)
    for g in range(8):
        word = word + (codes[g].astype(object) << (11 * g))
    out = np.zeros((T, 11), dtype=np.uint8)
    for i in range(11):
        out[:, i] = np.array([(int(w) >> (8 * i)) & 0xFF for w in word], dtype=np.uint8) if T else 0
    return out.reshape(-1)


def unpack(payload, T):
    b = np.asarray(payload, dtype=np.uint8).reshape(T, 11)
    codes = np.zeros((8, T), dtype=np.int32)
    for t in range(T):
        w = int.from_bytes(bytes(b[t].tolist()), "little")
        for g in range(8):
            codes[g, t] = (w >> (11 * g)) & 0x7FF
    return codes
