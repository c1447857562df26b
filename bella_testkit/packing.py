"""2-bit read packing exactly as csrc/assemble.hpp:k_pack_reads lays it out (host mirror for tests/tools)."""
import numpy as np


def pack_codes(codes: np.ndarray) -> np.ndarray:
    """base t of the concatenated array -> bits 2(t%16).. of word t/16; 4 zero words of padding."""
    n = codes.shape[0]
    nw = (n + 15) // 16
    buf = np.zeros(nw * 16, dtype=np.uint32)
    buf[:n] = codes
    buf = buf.reshape(nw, 16)
    sh = (2 * np.arange(16, dtype=np.uint32))[None, :]
    words = np.bitwise_or.reduce(buf << sh, axis=1).astype(np.uint32)
    return np.concatenate([words, np.zeros(4, dtype=np.uint32)])
