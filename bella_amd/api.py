"""Host-side mirror of the reference's call surface for the overlap hot path, over the C ABI.

`Engine` wraps one `bella_ctx` (one GPU).  `hash_spgemm(...)` is shaped like the reference's
`HashSpGEMM(A, B, multop, addop, reads, getvaluetype, filename, bpars, ratiophi)`
(include/overlap.hpp:650-652): operands + reads + BELLApars in, results delivered as the output file in
BELLA / PAF format and the stdout protocol numbers."""
from __future__ import annotations

import ctypes as C
import os
import dataclasses
import sys

import numpy as np

from . import _lib
from ._lib import ALN_DT, EXT_DT, PAIR_DT, SEED_DT, Memory, Params, Timings, WriteStats


class BellaHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("bella_hip error %d: %s" % (code, msg))
        self.code = code


@dataclasses.dataclass
class BellaPars:
    """BELLApars (include/common/common.h:46-74): the fields of the hot path, reference defaults."""
    kmerSize: int = 17
    binSize: int = 500
    xDrop: int = 7
    skipAlignment: bool = False
    outputPaf: bool = False
    errorRate: float = 0.15
    deltaChernoff: float = 0.10

    def c(self) -> Params:
        return Params(self.kmerSize, self.binSize, self.xDrop, int(self.skipAlignment), self.errorRate, self.deltaChernoff)


def _p(a):
    return a.ctypes.data if a is not None and a.size else None


_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


class Engine:
    def __init__(self, device: int = 0):
        self.lib = _lib.load()
        h = C.c_void_p()
        rc = self.lib.bella_hip_init(device, C.byref(h))
        if rc:
            raise BellaHipError(rc, self.lib.bella_hip_strerror(rc).decode() + " (bella_hip_init; no CPU fallback exists)")
        self.h = h
        self.nreads = 0
        self.lengths = None
        self.names = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.bella_hip_destroy(self.h)
            self.h = None

    __del__ = close

    def _chk(self, rc):
        if rc:
            raise BellaHipError(rc, self.lib.bella_hip_last_error(self.h).decode() or self.lib.bella_hip_strerror(rc).decode())

    # ---- reads (readVector_) ----
    def set_reads(self, rs):
        asc = np.ascontiguousarray(_ACGT[rs.codes])       # the C ABI takes ASCII bases, as the reference's readVector_ holds them
        offs = np.ascontiguousarray(rs.offsets, dtype=np.uint64)
        self._chk(self.lib.bella_hip_set_reads(self.h, _p(asc), offs.ctypes.data, rs.nreads))
        self.nreads = rs.nreads
        self.lengths = rs.lengths
        self.names = rs.names

    def load_fastq(self, path):
        """ParallelFASTQ + get_fq_name (kmercode/fq_reader.c) + 2-bit packing: the file goes straight into the library.
        `path` may be a list of files (the reference's -f list, kmercount.hpp:82-105): read ids continue from file to file."""
        n, nb = C.c_uint32(0), C.c_uint64(0)
        if isinstance(path, (list, tuple)):
            arr = (C.c_char_p * max(len(path), 1))(*[os.fsencode(x) for x in path])
            self._chk(self.lib.bella_hip_load_fastq_list(self.h, arr, len(path), C.byref(n), C.byref(nb)))
        else:
            self._chk(self.lib.bella_hip_load_fastq(self.h, os.fsencode(path), C.byref(n), C.byref(nb)))
        self.nreads = n.value
        need = C.c_uint64(0)
        self._chk(self.lib.bella_hip_get_read_names(self.h, None, 0, None, C.byref(need)))
        buf = C.create_string_buffer(max(need.value, 1))
        offs = np.zeros(self.nreads + 1, np.uint64)
        self._chk(self.lib.bella_hip_get_read_names(self.h, buf, need.value, offs.ctypes.data, None))
        raw = buf.raw
        self.names = [raw[int(offs[i]):int(offs[i + 1]) - 1].decode() for i in range(self.nreads)]
        lens = np.zeros(max(self.nreads, 1), np.uint32)
        self._chk(self.lib.bella_hip_get_read_lengths(self.h, lens.ctypes.data))
        self.lengths = lens[:self.nreads]
        return self.nreads, nb.value

    def ingest_stats(self):
        """what the last load_fastq did: file bytes, bases, reads, host threads, index_ms, upload_ms"""
        st = _lib.IngestStats()
        self._chk(self.lib.bella_hip_get_ingest_stats(self.h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in st._fields_}

    def set_reads_raw(self, ascii_bases: np.ndarray, offsets: np.ndarray, names=None):
        offs = np.ascontiguousarray(offsets, dtype=np.uint64)
        asc = np.ascontiguousarray(ascii_bases, dtype=np.uint8)
        self._chk(self.lib.bella_hip_set_reads(self.h, _p(asc), offs.ctypes.data, len(offs) - 1))
        self.nreads = len(offs) - 1
        self.lengths = np.diff(offs).astype(np.uint32)
        self.names = names

    # ---- SplitCount + tuple generation (kmercount.hpp:467-677, main.cpp:393-416) ----
    def count_kmers(self, k=17, lower=2, upper=8, syncmer=False, window=0):
        """reliable dictionary and tuples of the reads on the device (syncmer=True: the reference's -s mode);
        returns (nkmers, ntuples, ndistinct)"""
        nk, nt, nd = C.c_uint32(0), C.c_uint64(0), C.c_uint64(0)
        if window and not syncmer:                      # main.cpp:165-171: -w selects minimizers unless -s is given
            self._chk(self.lib.bella_hip_count_minimizers(self.h, k, window, lower, upper, C.byref(nk), C.byref(nt), C.byref(nd)))
        else:
            fn = self.lib.bella_hip_count_syncmers if syncmer else self.lib.bella_hip_count_kmers
            self._chk(fn(self.h, k, lower, upper, C.byref(nk), C.byref(nt), C.byref(nd)))
        self.nkmers_counted, self.ntuples_counted = nk.value, nt.value
        return nk.value, nt.value, nd.value

    def get_dictionary(self):
        codes = np.zeros(max(self.nkmers_counted, 1), np.uint64)
        counts = np.zeros(max(self.nkmers_counted, 1), np.uint16)
        self._chk(self.lib.bella_hip_get_dictionary(self.h, codes.ctypes.data, counts.ctypes.data))
        return codes[:self.nkmers_counted], counts[:self.nkmers_counted]

    def get_tuples(self):
        n = self.ntuples_counted
        tk, tr, tp = np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.uint16)
        self._chk(self.lib.bella_hip_get_tuples(self.h, tk.ctypes.data, tr.ctypes.data, tp.ctypes.data))
        return tk[:n], tr[:n], tp[:n]

    def assemble_counted(self):
        self._chk(self.lib.bella_hip_assemble_counted(self.h))

    def assemble_counted_panel(self, first_read, nreads_panel):
        self._chk(self.lib.bella_hip_assemble_counted_panel(self.h, first_read, nreads_panel))

    # ---- operands ----
    def assemble_tuples(self, k, nkmers, tk, tr, tp):
        tk = np.ascontiguousarray(tk, np.uint32); tr = np.ascontiguousarray(tr, np.uint32); tp = np.ascontiguousarray(tp, np.uint16)
        self._chk(self.lib.bella_hip_assemble_tuples(self.h, k, nkmers, len(tk), _p(tk), _p(tr), _p(tp)))

    def set_B(self, k, nkmers, colptr, rowids, values):
        colptr = np.ascontiguousarray(colptr, np.uint32); rowids = np.ascontiguousarray(rowids, np.uint32)
        values = np.ascontiguousarray(values, np.uint16)
        self._chk(self.lib.bella_hip_set_B(self.h, k, nkmers, colptr.ctypes.data, _p(rowids), _p(values)))

    # ---- multi-GPU assembly: row-block panels ----
    def set_B_panel(self, k, nkmers, first_read, nreads_panel, colptr, rowids, values):
        """bella_hip_set_B_panel: this context's row block of the reference's CSC of B (whole arrays given, the slice is uploaded)"""
        colptr = np.ascontiguousarray(colptr, np.uint32); rowids = np.ascontiguousarray(rowids, np.uint32); values = np.ascontiguousarray(values, np.uint16)
        self._chk(self.lib.bella_hip_set_B_panel(self.h, k, nkmers, first_read, nreads_panel, _p(colptr), _p(rowids), _p(values)))

    def assemble_panel(self, k, nkmers, first_read, nreads_panel, tk, tr, tp):
        tk = np.ascontiguousarray(tk, np.uint32); tr = np.ascontiguousarray(tr, np.uint32); tp = np.ascontiguousarray(tp, np.uint16)
        self._chk(self.lib.bella_hip_assemble_panel(self.h, k, nkmers, first_read, nreads_panel, len(tk), _p(tk), _p(tr), _p(tp)))

    def panel_tensors(self, device_index=0):
        """(rowcnt int32[rows], rowids int32[nnz], values int16[nnz]) torch tensors ALIASING the library's device buffers
        (valid until the next assembly call) -- what goes into the all-gather."""
        import torch
        first, rows, nnz = C.c_uint32(0), C.c_uint32(0), C.c_uint64(0)
        pc, pr, pv = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._chk(self.lib.bella_hip_panel_device_ptrs(self.h, C.byref(first), C.byref(rows), C.byref(nnz), C.byref(pc), C.byref(pr),
                                                       C.byref(pv)))

        class _Alias:
            def __init__(self, ptr, n, typestr):
                self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}

        dev = torch.device("cuda", device_index)

        def t(ptr, n, typestr, dt):
            return torch.as_tensor(_Alias(ptr, n, typestr), device=dev) if n else torch.zeros(0, dtype=dt, device=dev)
        return (t(pc.value, rows.value, "<i4", torch.int32), t(pr.value, nnz.value, "<i4", torch.int32),
                t(pv.value, nnz.value, "<i2", torch.int16))

    # ---- multi-GPU: the library's own RCCL communicator (include/bella_hip.h) ----
    def comm_available(self) -> bool:
        """librccl can be loaded in this process (check on every rank BEFORE the collective comm_init)"""
        return bool(self.lib.bella_hip_comm_available())

    def comm_id(self, local: bool = False) -> bytes:
        """the 128-byte id rank 0 hands to every rank; local=True: the in-process transport (contexts of one process)"""
        buf = C.create_string_buffer(128)
        rc = (self.lib.bella_hip_comm_id_local if local else self.lib.bella_hip_comm_id)(buf)
        if rc:
            raise BellaHipError(rc, "librccl could not be loaded")
        return buf.raw

    def comm_init(self, nranks: int, rank: int, comm_id: bytes, local: bool = False):
        buf = C.create_string_buffer(bytes(comm_id), 128)
        self._chk((self.lib.bella_hip_comm_init_local if local else self.lib.bella_hip_comm_init)(self.h, nranks, rank, buf))

    def comm_destroy(self):
        self._chk(self.lib.bella_hip_comm_destroy(self.h))

    def count_kmers_dist(self, first_read, nreads_block, k=17, lower=2, upper=8, syncmer=False, window=0):
        """collective (needs comm_init): the dictionary is counted across the ranks, tuples only for this rank's read block"""
        nk, nt, nd = C.c_uint32(0), C.c_uint64(0), C.c_uint64(0)
        sel = 1 if syncmer else (2 if window else 0)
        self._chk(self.lib.bella_hip_count_kmers_dist(self.h, k, lower, upper, sel, window, first_read, nreads_block, C.byref(nk), C.byref(nt),
                                                      C.byref(nd)))
        self.nkmers_counted, self.ntuples_counted = nk.value, nt.value
        return nk.value, nt.value, nd.value

    def allgather_panels(self):
        """collective: every rank's row block of B -> the whole matrix on every rank, device layout built"""
        self._chk(self.lib.bella_hip_allgather_panels(self.h))

    def set_B_device(self, k, nkmers, colptr_t, rowids_t, values_t):
        """full B from torch device tensors (int32 colptr[nreads+1], int32 rowids, int16 values)"""
        nnz = int(rowids_t.numel())
        self._chk(self.lib.bella_hip_set_B_device(self.h, k, nkmers, colptr_t.data_ptr(), rowids_t.data_ptr() if nnz else None,
                                                  values_t.data_ptr() if nnz else None, nnz))

    def get_B(self):
        nnz = C.c_uint64(0)
        self._chk(self.lib.bella_hip_get_B(self.h, C.byref(nnz), None, None, None))
        colptr = np.zeros(self.nreads + 1, np.uint32)
        rowids = np.zeros(max(nnz.value, 1), np.uint32)
        values = np.zeros(max(nnz.value, 1), np.uint16)
        self._chk(self.lib.bella_hip_get_B(self.h, C.byref(nnz), colptr.ctypes.data, rowids.ctypes.data, values.ctypes.data))
        return colptr, rowids[:nnz.value], values[:nnz.value]

    def set_partition(self, first, stride):
        self._chk(self.lib.bella_hip_set_partition(self.h, first, stride))

    def set_column_range(self, first, count):
        self._chk(self.lib.bella_hip_set_column_range(self.h, first, count))

    def set_debug(self, flags):
        self._chk(self.lib.bella_hip_set_debug(self.h, flags))

    # ---- HashSpGEMM ----
    TUNE = {"lds_tiers": 0, "kcount_budget": 1, "wide_budget": 2, "xdrop_variant": 3, "row_lists": 4, "xdrop_class_min": 5,
            "layout_order": 6, "inline_entries": 7, "row_path": 8, "cache_bytes": 9, "dist_layout": 10, "compact_b": 11}

    def reserve(self, nbytes: int) -> float:
        """bella_hip_reserve: one slab of device memory up front (the stages then allocate nothing from the driver); returns the ms it took"""
        ms = C.c_double(0.0)
        self._chk(self.lib.bella_hip_reserve(self.h, int(nbytes), C.byref(ms)))
        return ms.value

    def trim(self):
        """bella_hip_trim: released buffers the context keeps for reuse go back to the driver"""
        self._chk(self.lib.bella_hip_trim(self.h))

    def set_tuning(self, what: str, *values):
        """bella_hip_set_tuning: per-context tuning parameters (tests, A/B measurements); no values = the default"""
        v = np.asarray(values, dtype=np.uint64)
        self._chk(self.lib.bella_hip_set_tuning(self.h, self.TUNE[what], v.ctypes.data if len(v) else None, len(v)))

    def overlap(self, pars: BellaPars):
        n, f = C.c_uint64(0), C.c_uint64(0)
        cp = pars.c()
        self._chk(self.lib.bella_hip_overlap(self.h, C.byref(cp), C.byref(n), C.byref(f)))
        self.npairs, self.flops = n.value, f.value
        return n.value, f.value

    def count_pairs(self, pars: BellaPars):
        """bella_hip_count_pairs: the symbolic phase alone (estimateFLOP + estimateNNZ_Hash + prefixsum) -> colptrC, nnz(C), products"""
        n, f = C.c_uint64(0), C.c_uint64(0)
        cp = pars.c()
        colptrC = np.zeros(self.nreads + 1, np.uint64)
        self._chk(self.lib.bella_hip_count_pairs(self.h, C.byref(cp), colptrC.ctypes.data, C.byref(n), C.byref(f)))
        return colptrC, n.value, f.value

    def count_flops(self, pars: BellaPars):
        """estimateFLOP alone (bella_hip_count_pairs without the pair outputs): the products of the context's columns"""
        f = C.c_uint64(0)
        cp = pars.c()
        self._chk(self.lib.bella_hip_count_pairs(self.h, C.byref(cp), None, None, C.byref(f)))
        return f.value

    def get_pairs(self, ext=True):
        pairs = np.zeros(self.npairs, PAIR_DT)
        ex = np.zeros(self.npairs, EXT_DT) if ext else None
        colptrC = np.zeros(self.nreads + 1, np.uint64)
        self._chk(self.lib.bella_hip_get_pairs(self.h, _p(pairs), _p(ex) if ext else None, colptrC.ctypes.data))
        return pairs, ex, colptrC

    # ---- RunPairWiseAlignments ----
    def align_pairs(self, pars: BellaPars, exact: bool = False):
        """exact=True: the growing-band gapped X-drop of the reference's CUDA build (LOGAN / SeqAn extendSeed) instead of Xavier"""
        n = C.c_uint64(0)
        cp = pars.c()
        fn = self.lib.bella_hip_align_pairs_exact if exact else self.lib.bella_hip_align_pairs
        self._chk(fn(self.h, C.byref(cp), C.byref(n)))
        return n.value

    def get_alignments(self):
        out = np.zeros(self.npairs, ALN_DT)
        if self.npairs:
            self._chk(self.lib.bella_hip_get_alignments(self.h, out.ctypes.data))
        return out

    def xdrop_batch(self, seeds: np.ndarray, pars: BellaPars, exact: bool = False):
        seeds = np.ascontiguousarray(seeds, SEED_DT)
        out = np.zeros(len(seeds), ALN_DT)
        cp = pars.c()
        fn = self.lib.bella_hip_xdrop_batch_exact if exact else self.lib.bella_hip_xdrop_batch
        self._chk(fn(self.h, _p(seeds), len(seeds), C.byref(cp), _p(out)))
        return out

    def memory(self) -> Memory:
        m = Memory()
        self._chk(self.lib.bella_hip_get_memory(self.h, C.byref(m)))
        return m

    def timings(self) -> Timings:
        t = Timings()
        self._chk(self.lib.bella_hip_get_timings(self.h, C.byref(t)))
        return t


# ---------------------------------------------------------------------------------------------------
# output writers: overlap.hpp:472-473 (BELLA), :476-489 (PAF), :580-585 (--skip-alignment)
# ---------------------------------------------------------------------------------------------------
def overlap_of_seed(pairs, lengths, k):
    """chain.hpp:47-71 overlapop on the chosen seed (int, NOT truncated to u16) from the pair flags."""
    lenH = lengths[pairs["rid"]].astype(np.int64)
    lenV = lengths[pairs["cid"]].astype(np.int64)
    posH = pairs["seedH"].astype(np.int64)
    posV = pairs["seedV"].astype(np.int64)
    oriented = (pairs["flags"] & 1).astype(bool)
    begH = np.where(oriented, posH, (lenH - posH - k) & 0xFFFF)
    endH = (begH + k) & 0xFFFF
    endV = (posV + k) & 0xFFFF
    return np.minimum(begH, posV) + np.minimum(lenH - endH, lenV - endV) + k


def format_skip(names, lengths, pairs, k) -> bytes:
    ov = overlap_of_seed(pairs, lengths, k)
    l16 = lengths & 0xFFFF
    out = ["%s\t%s\t%d\t%d\t%d\t%d\n" % (names[c], names[r], cnt, o, l16[c], l16[r])
           for r, c, cnt, o in zip(pairs["rid"].tolist(), pairs["cid"].tolist(), pairs["count"].tolist(), ov.tolist())]
    return "".join(out).encode()


def format_aligned(names, lengths, pairs, alns, paf=False) -> bytes:
    l16 = (lengths & 0xFFFF).tolist()
    out = []
    for p, a in zip(pairs.tolist(), alns.tolist()):
        rid, cid, count = p[0], p[1], p[2]
        score, bH, eH, bV, eV, ov, strand, passed = a[:8]
        if not passed:
            continue
        r1, r2 = l16[rid], l16[cid]
        if not paf:
            out.append("%s\t%s\t%d\t%d\t%d\t%s\t%d\t%d\t%d\t%d\t%d\t%d\n" % (
                names[cid], names[rid], count, score, ov, "c" if strand else "n", bV, eV, r2, bH, eH, r1))
        else:
            if strand:
                bH, eH = r1 - eH, r1 - bH          # toOriginalCoordinates, overlap.hpp:149-154
            out.append("%s\t%d\t%d\t%d\t%s\t%s\t%d\t%d\t%d\t%d\t%d\t%d\n" % (
                names[cid], r2, bV, eV, "-" if strand else "+", names[rid], r1, bH, eH, score, ov, 255))
    return "".join(out).encode()


def write_output(filename: str, pars: BellaPars, names, lengths, pairs, alns=None, nthreads: int = 0) -> WriteStats:
    """bella_hip_write_output: the library's multi-threaded writer (overlap.hpp:603-642); APPENDS to `filename`."""
    lib = _lib.load()
    enc = [n.encode() if isinstance(n, str) else bytes(n) for n in names]
    arr = (C.c_char_p * len(enc))(*enc)
    lens = np.ascontiguousarray(lengths, np.uint32)
    pairs = np.ascontiguousarray(pairs, PAIR_DT)
    st = WriteStats()
    cp = pars.c()
    if alns is not None:
        alns = np.ascontiguousarray(alns, ALN_DT)
    rc = lib.bella_hip_write_output(filename.encode(), C.byref(cp), 1 if pars.outputPaf else 0, len(enc), C.cast(arr, C.c_void_p),
                                    lens.ctypes.data, pairs.ctypes.data if len(pairs) else None,
                                    alns.ctypes.data if alns is not None and len(alns) else None, len(pairs), nthreads, C.byref(st))
    if rc:
        raise BellaHipError(rc, "bella_hip_write_output failed")
    return st


def hash_spgemm(engine: Engine, pars: BellaPars, filename: str, stdout=sys.stdout, stages: int = 1):
    """HashSpGEMM-shaped driver (include/overlap.hpp:650-789): the operands and reads are already in `engine`.
    Writes `filename` and the stdout protocol lines nnz(C) (:686) and, when aligning, outputted per stage (:771).
    stages > 1: the output is formed in stages of consecutive columns like the reference does under a memory budget
    (:682-789); the file is the same, every pass only holds its own columns."""
    nreads = engine.nreads
    bounds = [(nreads * b) // stages for b in range(stages + 1)]
    outputted, total = [], 0
    open(filename, "wb").close()
    try:
        for b in range(stages):
            if stages > 1:
                engine.set_column_range(bounds[b], bounds[b + 1] - bounds[b])
            npairs, _ = engine.overlap(pars)
            total += npairs
            pairs, _, _ = engine.get_pairs(ext=False)
            if pars.skipAlignment:
                write_output(filename, pars, engine.names, engine.lengths, pairs)
            else:
                outputted.append(engine.align_pairs(pars))
                write_output(filename, pars, engine.names, engine.lengths, pairs, engine.get_alignments())
    finally:
        if stages > 1:
            engine.set_column_range(0, 0xFFFFFFFF)
    print(total, file=stdout)
    for o in outputted:
        print(o, file=stdout)
    return total
