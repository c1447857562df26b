"""TEST INFRASTRUCTURE: ctypes binding of oracle/bella_oracle.c (the CPU restatement) and, when present,
of oracle/_ref/libbella_ref.so (the reference's own code).  Only tests/, smoke() and bench.py's
cpu_baseline leg import this module; nothing under bella_amd/ does."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ODIR, "liboracle.so")
REFLIB = os.path.join(ODIR, "_ref", "libbella_ref.so")

PAIR_DT = np.dtype([("rid", "<u4"), ("cid", "<u4"), ("count", "<u2"), ("seedH", "<u2"), ("seedV", "<u2"),
                    ("nbins", "<u2"), ("support", "<u2"), ("binov", "<u2"), ("overlap", "<i4")])
ALN_DT = np.dtype([("score", "<i4"), ("begH", "<i4"), ("endH", "<i4"), ("begV", "<i4"), ("endV", "<i4"),
                   ("strand", "<i4"), ("flagged", "<i4"), ("steps", "<i4")])

u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
u16p = np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")


def build_oracle(force: bool = False) -> str:
    src = os.path.join(ODIR, "bella_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-std=c99", "-shared", "-fPIC", "-o", LIB, src, "-lm"])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build_oracle())
        L = _lib
        L.oracle_build_B.restype = C.c_int64
        L.oracle_build_B.argtypes = [C.c_uint32, C.c_uint64, u32p, u32p, u16p, u32p, u32p, u16p]
        L.oracle_transpose.argtypes = [C.c_uint32, C.c_uint32, u32p, u32p, u16p, u32p, u32p, u16p]
        L.oracle_symbolic.argtypes = [C.c_uint32, u32p, u32p, u32p, u32p, u32p, u32p]
        L.oracle_symbolic_range.argtypes = [C.c_uint32, C.c_uint32, u32p, u32p, u32p, u32p, u32p, u32p]
        L.oracle_numeric_cols.argtypes = [u32p, C.c_uint32, u32p, np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS"), u32p, u32p, u16p, u32p, u32p,
                                          u16p, C.POINTER(C.c_char_p), u32p, C.c_int, C.c_int, C.c_void_p]
        L.oracle_numeric.argtypes = [C.c_uint32, u32p, u32p, u16p, u32p, u32p, u16p, C.POINTER(C.c_char_p), u32p,
                                     C.c_int, C.c_int, u32p, C.c_void_p]
        L.oracle_xavier_align.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_void_p]
        L.oracle_logan_align.argtypes = L.oracle_xavier_align.argtypes
        L.oracle_xavier_xdrop.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_void_p]
        L.oracle_post_align.restype = C.c_int
        L.oracle_post_align.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32,
                                        C.c_double, C.c_double, C.POINTER(C.c_uint16)]
        L.oracle_slope.restype = C.c_double
        L.oracle_count_kmers.argtypes = [C.c_uint32, C.POINTER(C.c_char_p), u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_uint64)]
        L.oracle_count_kmers.restype = C.c_int64
        L.oracle_count_syncmers.argtypes = L.oracle_count_kmers.argtypes
        L.oracle_count_syncmers.restype = C.c_int64
        L.oracle_count_minimizers.argtypes = [C.c_uint32, C.POINTER(C.c_char_p), u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64),
                                              C.POINTER(C.c_uint64)]
        L.oracle_count_minimizers.restype = C.c_int64
        L.oracle_kmer_hash.argtypes = [C.c_uint64]
        L.oracle_kmer_hash.restype = C.c_uint64
        L.oracle_slope.argtypes = [C.c_double]
        L.oracle_choose_bin.restype = C.c_uint32
        L.oracle_choose_bin.argtypes = [u32p, C.c_uint32]
        L.oracle_overlapop.restype = C.c_int
        L.oracle_overlapop.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_uint16, C.c_uint16, C.c_uint16]
    return _lib


def count_kmers(seqs, k=17, lower=2, upper=8, syncmer=False, window=0):
    """returns (dict_codes u64[nk] ascending, dict_counts u16[nk], t_kmer, t_read, t_pos, ndistinct)"""
    nreads = len(seqs)
    arr = (C.c_char_p * max(nreads, 1))(*[bytes(s) for s in seqs])
    lens = np.asarray([len(s) for s in seqs] or [0], np.uint32)
    nt, nd = C.c_uint64(0), C.c_uint64(0)
    fn = lib().oracle_count_syncmers if syncmer else lib().oracle_count_kmers
    if window and not syncmer:                       # main.cpp:165-171: -s switches the minimizer mode off
        base = lib().oracle_count_minimizers
        fn = lambda nr, a, l, kk, lo, up, *rest: base(nr, a, l, kk, window, lo, up, *rest)
    nk = fn(nreads, arr, lens, k, lower, upper, None, None, None, None, None, C.byref(nt), C.byref(nd))
    codes = np.zeros(max(nk, 1), np.uint64)
    counts = np.zeros(max(nk, 1), np.uint16)
    tk = np.zeros(max(nt.value, 1), np.uint32)
    tr = np.zeros(max(nt.value, 1), np.uint32)
    tp = np.zeros(max(nt.value, 1), np.uint16)
    fn(nreads, arr, lens, k, lower, upper, codes.ctypes.data, counts.ctypes.data, tk.ctypes.data, tr.ctypes.data, tp.ctypes.data,
       C.byref(nt), C.byref(nd))
    n = nt.value
    return codes[:nk].copy(), counts[:nk].copy(), tk[:n].copy(), tr[:n].copy(), tp[:n].copy(), nd.value


def build_B(nreads, tk, tr, tp):
    n = len(tk)
    colptr = np.zeros(nreads + 1, np.uint32)
    rowids = np.zeros(max(n, 1), np.uint32)
    vals = np.zeros(max(n, 1), np.uint16)
    nnz = lib().oracle_build_B(nreads, n, np.ascontiguousarray(tk, np.uint32), np.ascontiguousarray(tr, np.uint32),
                               np.ascontiguousarray(tp, np.uint16), colptr, rowids, vals)
    return colptr, rowids[:nnz].copy(), vals[:nnz].copy()


def transpose(nreads, nkmers, Bc, Br, Bv):
    nnz = len(Br)
    Ac = np.zeros(nkmers + 1, np.uint32)
    Ar = np.zeros(max(nnz, 1), np.uint32)
    Av = np.zeros(max(nnz, 1), np.uint16)
    lib().oracle_transpose(nreads, nkmers, Bc, _pad(Br, np.uint32), _pad(Bv, np.uint16), Ac, Ar, Av)
    return Ac, Ar[:nnz].copy(), Av[:nnz].copy()


def _pad(a, dt):
    a = np.ascontiguousarray(a, dt)
    return a if a.size else np.zeros(1, dt)


def spgemm(seqs, nkmers, Bc, Br, Bv, k=17, bin_size=500):
    """returns (colflop, colptrC, pairs[PAIR_DT]) in the reference's 1-thread output order"""
    nreads = len(seqs)
    Ac, Ar, Av = transpose(nreads, nkmers, Bc, Br, Bv)
    flop = np.zeros(nreads, np.uint32)
    nnzc = np.zeros(nreads, np.uint32)
    lib().oracle_symbolic(nreads, Bc, _pad(Br, np.uint32), Ac, _pad(Ar, np.uint32), flop, nnzc)
    colptrC = np.zeros(nreads + 1, np.uint32)
    np.cumsum(nnzc, out=colptrC[1:])
    pairs = np.zeros(int(colptrC[-1]), PAIR_DT)
    arr = (C.c_char_p * nreads)(*[bytes(s) for s in seqs])
    lens = np.asarray([len(s) for s in seqs], np.uint32)
    out = pairs if pairs.size else np.zeros(1, PAIR_DT)
    lib().oracle_numeric(nreads, Bc, _pad(Br, np.uint32), _pad(Bv, np.uint16), Ac, _pad(Ar, np.uint32),
                         _pad(Av, np.uint16), arr, lens, k, bin_size, colptrC, out.ctypes.data)
    return flop, colptrC, pairs


# ---- full-size helpers: the columns are independent, so ranges / samples of them can be spread over the host cores ----
_PAR = {}


def _sym_job(rng):
    lo, hi = rng
    P = _PAR
    flop = np.zeros(P["nreads"], np.uint32)
    nnzc = np.zeros(P["nreads"], np.uint32)
    lib().oracle_symbolic_range(lo, hi, P["Bc"], P["Br"], P["Ac"], P["Ar"], flop, nnzc)
    return lo, hi, flop[lo:hi].copy(), nnzc[lo:hi].copy()


def _num_job(cols):
    P = _PAR
    cols = np.ascontiguousarray(cols, np.uint32)
    nz = np.ascontiguousarray(P["nnzc"][cols], np.uint32)
    off = np.zeros(len(cols), np.uint64)
    np.cumsum(nz[:-1], out=off[1:])
    out = np.zeros(max(int(nz.sum()), 1), PAIR_DT)
    lib().oracle_numeric_cols(cols, len(cols), nz, off, P["Bc"], P["Br"], P["Bv"], P["Ac"], P["Ar"], P["Av"], P["arr"], P["lens"], P["k"],
                              P["bin"], out.ctypes.data)
    return out[:int(nz.sum())]


def seq_pointers(asc: np.ndarray, offsets: np.ndarray):
    """char** into ONE ASCII buffer (reads back to back, no terminators: the restatement takes lengths): what a million-read set
    hands the restatement instead of a million bytes objects.  Returns (pointer array, lengths); `asc` must stay alive."""
    offs = np.asarray(offsets, np.int64)
    addr = (np.uint64(asc.ctypes.data) + offs[:-1].astype(np.uint64)).astype(np.uint64)
    arr = (C.c_char_p * max(len(addr), 1)).from_buffer_copy(addr.tobytes() if len(addr) else b"\0" * 8)
    return arr, np.diff(offs).astype(np.uint32)


def spgemm_parallel(seqs, nkmers, Bc, Br, Bv, sample_cols, k=17, bin_size=500, procs=None):
    """symbolic phase on ALL columns and numeric phase on `sample_cols` only, over a fork pool.
    returns (colflop, colnnz, {col: pairs[PAIR_DT] in slot order}).  `seqs`: a list of bytes, or (ascii buffer, offsets)."""
    import multiprocessing as mp
    packed = isinstance(seqs, tuple)
    nreads = len(seqs[1]) - 1 if packed else len(seqs)
    Ac, Ar, Av = transpose(nreads, nkmers, Bc, Br, Bv)
    procs = procs or min(64, os.cpu_count() or 1)
    _PAR.update(nreads=nreads, Bc=Bc, Br=_pad(Br, np.uint32), Bv=_pad(Bv, np.uint16), Ac=Ac, Ar=_pad(Ar, np.uint32), Av=_pad(Av, np.uint16), k=k,
                bin=bin_size)
    step = max(1, nreads // (procs * 8))
    ranges = [(lo, min(nreads, lo + step)) for lo in range(0, nreads, step)]
    flop = np.zeros(nreads, np.uint32)
    nnzc = np.zeros(nreads, np.uint32)
    with mp.get_context("fork").Pool(procs) as pool:
        for lo, hi, f, z in pool.imap_unordered(_sym_job, ranges):
            flop[lo:hi] = f
            nnzc[lo:hi] = z
    sample_cols = np.asarray(sample_cols, np.uint32)
    if packed:
        arr, lens = seq_pointers(seqs[0], seqs[1])
        _PAR.update(nnzc=nnzc, arr=arr, lens=lens, keep=seqs[0])
    else:
        _PAR.update(nnzc=nnzc, arr=(C.c_char_p * nreads)(*[bytes(s) for s in seqs]), lens=np.asarray([len(s) for s in seqs], np.uint32))
    chunks = [sample_cols[x:x + 16] for x in range(0, len(sample_cols), 16)]
    per_col = {}
    with mp.get_context("fork").Pool(procs) as pool:
        for cols, rec in zip(chunks, pool.map(_num_job, chunks)):
            o = 0
            for c in cols:
                n = int(nnzc[c])
                per_col[int(c)] = rec[o:o + n]
                o += n
    _PAR.clear()
    return flop, nnzc, per_col


def xavier_align(row: bytes, col: bytes, i: int, j: int, x: int = 7, k: int = 17):
    out = np.zeros(1, ALN_DT)
    lib().oracle_xavier_align(row, len(row), col, len(col), i, j, x, k, out.ctypes.data)
    return out[0]


def xavier_xdrop(target: bytes, query: bytes, begH: int, begV: int, k: int, x: int):
    out = np.zeros(1, ALN_DT)
    lib().oracle_xavier_xdrop(target, len(target), query, len(query), begH, begV, k, x, out.ctypes.data)
    return out[0]


def logan_align(row: bytes, col: bytes, i: int, j: int, x: int = 7, k: int = 17):
    """the exact gapped X-drop of the reference's CUDA build (loganGPU/functions.cuh), pair prepared as overlap.hpp:918-944 does"""
    out = np.zeros(1, ALN_DT)
    lib().oracle_logan_align(row, len(row), col, len(col), i, j, x, k, out.ctypes.data)
    return out[0]


def post_align_gpu(score, begV, endV, begH, endH, lenH, lenV, ratiophi, delta=0.1):
    """PostAlignDecisionGPU (overlap.hpp:797-871): the threshold test in double"""
    ov = C.c_uint16(0)
    L = lib()
    L.oracle_post_align_gpu.argtypes = L.oracle_post_align.argtypes
    ok = L.oracle_post_align_gpu(int(score), int(begV), int(endV), int(begH), int(endH), int(lenH), int(lenV), float(ratiophi), float(delta),
                                 C.byref(ov))
    return bool(ok), int(ov.value)


def post_align(score, begV, endV, begH, endH, lenH, lenV, ratiophi, delta=0.1):
    ov = C.c_uint16(0)
    ok = lib().oracle_post_align(int(score), int(begV), int(endV), int(begH), int(endH), int(lenH), int(lenV),
                                 float(ratiophi), float(delta), C.byref(ov))
    return bool(ok), int(ov.value)


def slope(e):
    return lib().oracle_slope(float(e))


# ---- output formatting (overlap.hpp:472-489, 580-585), independent of the product's writer ----------
def skip_lines(names, lens, pairs) -> bytes:
    out = []
    for p in pairs:
        out.append("%s\t%s\t%d\t%d\t%d\t%d\n" % (names[p["cid"]], names[p["rid"]], p["count"], p["overlap"],
                                                 lens[p["cid"]] & 0xFFFF, lens[p["rid"]] & 0xFFFF))
    return "".join(out).encode()


def align_lines(names, seqs, pairs, x=7, k=17, err=0.15, delta=0.1, paf=False):
    """returns (bytes, alignments[ALN_DT], passed mask)"""
    phi = slope(err)
    out = []
    alns = np.zeros(len(pairs), ALN_DT)
    passed = np.zeros(len(pairs), bool)
    for n, p in enumerate(pairs):
        rid, cid = int(p["rid"]), int(p["cid"])
        a = xavier_align(seqs[rid], seqs[cid], int(p["seedH"]), int(p["seedV"]), x, k)
        alns[n] = a
        ok, ov = post_align(a["score"], a["begV"], a["endV"], a["begH"], a["endH"], len(seqs[rid]), len(seqs[cid]), phi, delta)
        passed[n] = ok
        if not ok:
            continue
        r1, r2 = len(seqs[rid]) & 0xFFFF, len(seqs[cid]) & 0xFFFF
        st = "c" if a["strand"] else "n"
        if not paf:
            out.append("%s\t%s\t%d\t%d\t%d\t%s\t%d\t%d\t%d\t%d\t%d\t%d\n" % (
                names[cid], names[rid], p["count"], a["score"], ov, st, a["begV"], a["endV"], r2, a["begH"], a["endH"], r1))
        else:
            bH, eH = int(a["begH"]), int(a["endH"])
            if st == "c":
                bH, eH = r1 - eH, r1 - int(np.uint32(bH))   # toOriginalCoordinates, overlap.hpp:149-154
            out.append("%s\t%d\t%d\t%d\t%s\t%s\t%d\t%d\t%d\t%d\t%d\t%d\n" % (
                names[cid], r2, a["begV"], a["endV"], "-" if st == "c" else "+", names[rid], r1, bH, eH, a["score"], ov, 255))
    return "".join(out).encode(), alns, passed


# ---- the reference itself, in-process (only where oracle/_ref was built) ---------------------------
_ref = None


def have_ref() -> bool:
    return os.path.exists(REFLIB)


def ref():
    global _ref
    if _ref is None:
        _ref = C.CDLL(REFLIB)
        R = _ref
        R.bella_ref_build_B.restype = C.c_int64
        R.bella_ref_build_B.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64, u32p, u32p, u16p, u32p, u32p, u16p]
        R.bella_ref_transpose_B.restype = C.c_int64
        R.bella_ref_transpose_B.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64, u32p, u32p, u16p, u32p, u32p, u16p]
        R.bella_ref_hashspgemm.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64, u32p, u32p, u16p, C.POINTER(C.c_char_p),
                                           C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_double, C.c_double, C.c_double, C.c_char_p, C.c_char_p, C.c_size_t,
                                           C.c_char_p, C.c_size_t]
        R.bella_ref_xavier_align.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.POINTER(C.c_int), C.c_char_p]
        R.bella_ref_seqan_align.argtypes = R.bella_ref_xavier_align.argtypes
        R.bella_ref_slope.restype = C.c_double
        R.bella_ref_slope.argtypes = [C.c_double]
    return _ref


def ref_build_B(nreads, nkmers, tk, tr, tp):
    n = len(tk)
    colptr = np.zeros(nreads + 1, np.uint32)
    rowids = np.zeros(max(n, 1), np.uint32)
    vals = np.zeros(max(n, 1), np.uint16)
    nnz = ref().bella_ref_build_B(nreads, nkmers, n, _pad(tk, np.uint32), _pad(tr, np.uint32), _pad(tp, np.uint16),
                                  colptr, rowids, vals)
    return colptr, rowids[:nnz].copy(), vals[:nnz].copy()


def ref_hashspgemm(seqs, names, nkmers, tk, tr, tp, outfile, k=17, bin_size=500, xdrop=7, skip=True, paf=False,
                   err=0.15, delta=0.1, mem_mb=64000.0):
    """runs the reference's HashSpGEMM on the tuples; returns (file bytes, stdout, stderr)"""
    nreads = len(seqs)
    sarr = (C.c_char_p * nreads)(*[bytes(s) for s in seqs])
    narr = (C.c_char_p * nreads)(*[n.encode() for n in names])
    so = C.create_string_buffer(1 << 16)
    se = C.create_string_buffer(1 << 16)
    ref().bella_ref_hashspgemm(nreads, nkmers, len(tk), _pad(tk, np.uint32), _pad(tr, np.uint32), _pad(tp, np.uint16),
                               sarr, narr, k, bin_size, xdrop, int(skip), int(paf), err, delta, mem_mb,
                               outfile.encode(), so, len(so), se, len(se))
    data = open(outfile, "rb").read() if os.path.exists(outfile) else b""
    return data, so.value.decode(), se.value.decode()


def ref_seqan_align(row: bytes, col: bytes, i, j, x=7, k=17):
    """alignSeqAn (align.hpp:93) = SeqAn's gapped X-drop extendSeed from the reference tree"""
    o = (C.c_int * 5)()
    st = C.create_string_buffer(2)
    ref().bella_ref_seqan_align(row, col, len(row), i, j, x, k, o, st)
    return list(o), st.value.decode()[:1]


def ref_xavier_align(row: bytes, col: bytes, i, j, x=7, k=17):
    o = (C.c_int * 5)()
    st = C.create_string_buffer(2)
    ref().bella_ref_xavier_align(row, col, len(row), i, j, x, k, o, st)
    return list(o), st.value.decode()[:1]
