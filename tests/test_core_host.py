"""CPU tests of the per-lane functions the kernels inline (bella_amd/csrc/core.hpp, xdrop.hpp), compiled as
host C++ by tests/harness/core_harness.cpp and compared with the oracle on the golden sets.  This is a
unit-test harness: the product library has no CPU path."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import _oracle as O
from bella_testkit import packing, synth
from conftest import ROOT, load_golden

HSRC = os.path.join(ROOT, "tests", "harness", "core_harness.cpp")
HLIB = os.path.join(ROOT, "tests", "harness", "libcore_harness.so")

ALN_DT = np.dtype([("score", "<i4"), ("begH", "<i4"), ("endH", "<i4"), ("begV", "<i4"), ("endV", "<i4"), ("ov", "<u2"),
                   ("strand", "u1"), ("passed", "u1"), ("steps", "<u4"), ("flagged", "<u4")])


@pytest.fixture(scope="module")
def H():
    deps = [HSRC] + [os.path.join(ROOT, "bella_amd", "csrc", f) for f in ("core.hpp", "xdrop.hpp", "fastq.hpp")]
    if not os.path.exists(HLIB) or any(os.path.getmtime(d) > os.path.getmtime(HLIB) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", HLIB, HSRC])
    h = C.CDLL(HLIB)
    h.h_choose.restype = C.c_uint32
    h.h_choose.argtypes = [C.c_void_p, C.c_uint32]
    h.h_overlap_estimate.restype = C.c_int
    h.h_xdrop_pair.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                               C.c_int, C.c_double, C.c_double, C.c_void_p]
    h.h_kmer_words.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]
    h.h_fold_pair.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_void_p]
    h.h_parse_fastq.restype = C.c_long
    h.h_parse_fastq.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64)]
    h.h_fastq_name.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64]
    h.h_parse_fastq2.restype = C.c_long
    h.h_parse_fastq2.argtypes = [C.c_char_p, C.c_int, C.c_uint, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_char_p,
                                 C.c_uint64, C.POINTER(C.c_uint64), C.c_uint64, C.c_uint64]
    return h


def products_by_pair(g, Bc, Br, Bv):
    """per output pair (cid, rid): products in the reference's order as (posH, posV, ov)"""
    nreads = g.rs.nreads
    Ac, Ar, Av = O.transpose(nreads, g.nkmers, Bc, Br, Bv)
    lens = g.rs.lengths
    out = {}
    for i in range(nreads):
        for j in range(Bc[i], Bc[i + 1]):
            km, posV = Br[j], int(Bv[j])
            for x in range(Ac[km], Ac[km + 1]):
                key = int(Ar[x])
                if key <= i:
                    continue
                posH = int(Av[x])
                ov = O.lib().oracle_overlapop(g.seqs[key], int(lens[key]), g.seqs[i], int(lens[i]), posH, posV, g.k) & 0xFFFF
                out.setdefault((i, key), []).append((posH, posV, ov))
    return out


@pytest.mark.parametrize("name", ["sanity3", "toy120", "toyhifi50", "toyrep90"])
def test_fold_pair_matches_oracle(H, name):
    g = load_golden(name)
    Bc, Br, Bv = O.build_B(g.rs.nreads, g.tk, g.tr, g.tp)
    _, _, pairs = O.spgemm(g.seqs, g.nkmers, Bc, Br, Bv, g.k)
    prods = products_by_pair(g, Bc, Br, Bv)
    assert len(prods) == len(pairs)
    out = np.zeros(7, np.uint32)
    multi = 0
    for p in pairs:
        lst = prods[(int(p["cid"]), int(p["rid"]))]
        hv = np.asarray([a | (b << 16) for a, b, _ in lst], np.uint32)
        ov = np.asarray([c for _, _, c in lst], np.uint16)
        for shuffle in (0, 12345):
            H.h_fold_pair(hv.ctypes.data, ov.ctypes.data, len(lst), g.k, 500, 1, shuffle, out.ctypes.data)
            assert (int(out[0]), int(out[4]), int(out[5])) == (int(p["count"]), int(p["seedH"]), int(p["seedV"])), (p, shuffle)
            assert (int(out[1]), int(out[2]), int(out[3])) == (int(p["nbins"]), int(p["support"]), int(p["binov"]))
        multi += int(p["nbins"]) > 1
    if name == "toyrep90":
        assert multi > 0          # the repeat genome must exercise multi-bin states


def test_fold_many_bins_random(H):
    """adversarial lists: overlaps far apart -> many bins (> 16: std::sort path), then merges"""
    rng = np.random.default_rng(5)
    out = np.zeros(7, np.uint32)
    import ctypes
    for trial in range(300):
        m = int(rng.integers(2, 120))
        ovs = (rng.integers(0, 40, size=m) * int(rng.integers(100, 900))).astype(np.uint16)
        ph = rng.integers(0, 3000, size=m).astype(np.uint32)
        pv = rng.integers(0, 3000, size=m).astype(np.uint32)
        hv = (ph | (pv << 16)).astype(np.uint32)
        # python model of A.3 with libstdc++-equal choose via the oracle
        count, bins = 1, [[int(ovs[0]), [(int(ph[0]), int(pv[0]))]]]
        for t in range(1, m):
            q, oq = (int(ph[t]), int(pv[t])), int(ovs[t])
            ins, orphans = [], []
            for b in bins:
                if abs(b[0] - oq) < 500:
                    ins += [x for x in b[1] if abs(x[0] - q[0]) > 17 and abs(x[1] - q[1]) > 17]
                else:
                    orphans.append(b)
            count = (count + 1 + len(ins)) & 0xFFFF
            bins = [[oq, [q] + ins]] + orphans
        sup = np.asarray([len(b[1]) for b in bins], np.uint32)
        w = O.lib().oracle_choose_bin(sup, len(sup))
        H.h_fold_pair(hv.ctypes.data, ovs.ctypes.data, m, 17, 500, 1, trial, out.ctypes.data)
        assert int(out[0]) == count and int(out[1]) == len(bins)
        assert (int(out[2]), int(out[3]), int(out[4]), int(out[5])) == (len(bins[w][1]), bins[w][0], bins[w][1][0][0], bins[w][1][0][1]), trial


def test_id_sorter_matches_oracle_restatement(H):
    rng = np.random.default_rng(1)
    for trial in range(500):
        n = int(rng.integers(1, 600))
        s = rng.integers(1, int(rng.integers(2, 7)), size=n).astype(np.uint32)
        assert H.h_choose(s.ctypes.data, n) == O.lib().oracle_choose_bin(s, n)


def test_kmer_words(H):
    rs = synth.make_reads(5, read_len=300, seed=9)
    pk = packing.pack_codes(rs.codes)
    for k in (15, 17, 31, 32):
        fw, rc, rid, pos, valid = synth.kmer_words(rs, k)
        out = np.zeros(3, np.uint64)
        for gidx in list(range(0, 40)) + [100, 333, 777, len(fw) - 1]:
            H.h_kmer_words(pk.ctypes.data, gidx, k, out.ctypes.data)
            assert int(out[1]) == int(fw[gidx]) and int(out[2]) == int(rc[gidx]), (k, gidx)


def test_overlap_estimate_matches_oracle(H):
    rng = np.random.default_rng(2)
    a = bytes(synth.BASES[rng.integers(0, 4, 900, dtype=np.uint8)])
    for trial in range(2000):
        l1, l2 = int(rng.integers(40, 900)), int(rng.integers(40, 900))
        p1, p2 = int(rng.integers(0, l1 - 17)), int(rng.integers(0, l2 - 17))
        r1 = a[:l1]
        # plant the same / a different k-mer to drive checkstrand both ways
        same = bool(rng.integers(0, 2))
        r2 = bytearray(a[100:100 + l2] if l2 <= 800 else a[:l2])
        if same:
            r2[p2:p2 + 17] = r1[p1:p1 + 17]
        r2 = bytes(r2)
        oriented = r1[p1:p1 + 17] == r2[p2:p2 + 17]
        exp = O.lib().oracle_overlapop(r1, l1, r2, len(r2), p1, p2, 17)
        assert H.h_overlap_estimate(p1, p2, l1, len(r2), int(oriented), 17) == exp


def _xdrop_pairs(H, g, pairs, limit=None):
    pk = packing.pack_codes(g.rs.codes)
    offs = g.rs.offsets
    phi = O.slope(g.err)
    out = np.zeros(1, ALN_DT)
    bad = []
    nflag = 0
    for n, p in enumerate(pairs[:limit]):
        rid, cid = int(p["rid"]), int(p["cid"])
        H.h_xdrop_pair(pk.ctypes.data, int(offs[rid]), len(g.seqs[rid]), int(offs[cid]), len(g.seqs[cid]), int(p["seedH"]),
                       int(p["seedV"]), g.k, g.xdrop, phi, 0.1, out.ctypes.data)
        a = O.xavier_align(g.seqs[rid], g.seqs[cid], int(p["seedH"]), int(p["seedV"]), g.xdrop, g.k)
        ok, ov = O.post_align(a["score"], a["begV"], a["endV"], a["begH"], a["endH"], len(g.seqs[rid]), len(g.seqs[cid]), phi)
        got = out[0]
        exp = (int(a["score"]), int(a["begH"]), int(a["endH"]), int(a["begV"]), int(a["endV"]), ov, int(a["strand"]), int(ok),
               int(a["flagged"]), int(a["steps"]))
        have = (int(got["score"]), int(got["begH"]), int(got["endH"]), int(got["begV"]), int(got["endV"]), int(got["ov"]),
                int(got["strand"]), int(got["passed"]), int(got["flagged"]), int(got["steps"]))
        nflag += int(a["flagged"])
        if exp != have:
            bad.append((n, exp, have))
    return bad, nflag


@pytest.mark.parametrize("name", ["sanity3", "toy120", "toylen80", "toyhifi50"])
def test_xdrop_lane_function_matches_oracle(H, name):
    g = load_golden(name)
    Bc, Br, Bv = O.build_B(g.rs.nreads, g.tk, g.tr, g.tp)
    _, _, pairs = O.spgemm(g.seqs, g.nkmers, Bc, Br, Bv, g.k)
    bad, nflag = _xdrop_pairs(H, g, pairs, limit=1200)
    assert not bad, bad[:3]


def test_ordered_insertion_fixed_point_equals_sequential():
    """The atomicMin displacement scheme (spgemm.hpp phase O / assemble.hpp) under random interleavings."""
    rng = np.random.default_rng(3)
    EMPTY = 0xFFFFFFFF
    for trial in range(300):
        d = int(rng.integers(1, 200))
        ht = 16
        while ht < d:
            ht <<= 1
        keys = rng.choice(1 << 20, size=d, replace=False).astype(np.uint64)
        prio = rng.permutation(d)                     # first-occurrence order
        # sequential reference: insert in prio order, linear probing
        seq = [EMPTY] * ht
        for s in np.argsort(prio):
            h = int(keys[s] * 107) & (ht - 1)
            while seq[h] != EMPTY:
                h = (h + 1) & (ht - 1)
            seq[h] = int(s)
        # concurrent: every key is an agent (item, h); a random agent takes one atomicMin step at a time
        T2 = [EMPTY] * ht
        agents = [[(int(prio[s]) << 16) | s, int(keys[s] * 107) & (ht - 1)] for s in range(d)]
        live = list(range(d))
        while live:
            ai = int(rng.integers(0, len(live)))
            ag = agents[live[ai]]
            old = T2[ag[1]]
            T2[ag[1]] = min(old, ag[0])
            if old == EMPTY:
                live.pop(ai)
                continue
            if old > ag[0]:
                ag[0] = old
            ag[1] = (ag[1] + 1) & (ht - 1)
        got = [EMPTY if x == EMPTY else (x & 0xFFFF) for x in T2]
        assert got == seq, trial


# ---- FASTQ ingest (bella_amd/csrc/fastq.hpp, host code of the library) -----------------------------------------------------

def _parse(H, path):
    size = os.path.getsize(path)
    bases = np.zeros(size + 1, np.uint8)
    offs = np.zeros(size // 4 + 4, np.uint64)
    names = C.create_string_buffer(size + 2)
    nb = C.c_uint64(0)
    n = H.h_parse_fastq(os.fsencode(path), bases.ctypes.data, bases.size, offs.ctypes.data, offs.size, names, size + 2, C.byref(nb))
    assert n >= 0, n
    return n, bases[:nb.value].tobytes(), offs[:n + 1].astype(np.int64), names.value.decode().split("\n")[:-1]


def test_fastq_parse_matches_golden_reads_and_reference_names(H, golden, tmp_path):
    """the names the reference printed in its output files (fq_reader.c get_fq_name, main.cpp:357) and the bases it aligned"""
    import gzip
    g = golden
    p = tmp_path / "r.fastq"
    p.write_bytes(gzip.open(os.path.join(ROOT, "tests", "golden", g.name, "reads.fastq.gz"), "rb").read())
    n, bases, offs, names = _parse(H, str(p))
    assert n == g.rs.nreads and names == g.names
    assert bases == synth.BASES[g.rs.codes].tobytes() and np.array_equal(offs, g.rs.offsets)
    used = {ln.split(b"\t")[0].decode() for ln in g.out["skip"].split(b"\n") if ln} | \
           {ln.split(b"\t")[1].decode() for ln in g.out["skip"].split(b"\n") if ln}
    assert used <= set(names)


def test_fastq_name_rules(H):
    def nm(h):
        out = C.create_string_buffer(256)
        H.h_fastq_name(h.encode(), out, 256)
        return out.value.decode()
    assert nm("@read1") == "read1"
    assert nm("@read1   ") == "read1"
    assert nm("@m54/12/0_100 RQ=0.8") == "m54/12/0_100"
    assert nm("@r\tcomment here") == "r"
    assert nm("@pair 1:N:0:ATCACG") == "pair/1"            # new Illumina comment -> /1 (fq_reader.c:121-124)
    assert nm("@pair 2:Y:0:ATCACG") == "pair/2"
    assert nm("@pair 2:Y:18:ATCACG") == "pair"             # two-digit control field: "unknown pairing format" there too
    assert nm("@pair 3:N:0:ATCACG") == "pair"
    assert nm("@pair/1") == "pair/1"                       # already in /x form: untouched
    assert nm("@pair/1 extra") == "pair/1 extra"[:6]
    assert nm("@name x") == "name"


def test_fastq_edge_cases(H, tmp_path):
    p = tmp_path / "e.fastq"
    p.write_bytes(b"@a c\nACGT\n+\n!!!!\n@b\nGGCC\r\n+b\n!!!!")           # CRLF bases, no final newline
    n, bases, offs, names = _parse(H, str(p))
    assert (n, bases, offs.tolist(), names) == (2, b"ACGTGGCC", [0, 4, 8], ["a", "b"])
    p.write_bytes(b"@a\nACGT\n+\n!!!!\n\n@b\nGG\n+\n!!\n")                 # an empty line ends the parse (fq_reader.c:567-574)
    assert _parse(H, str(p))[0] == 1
    p.write_bytes(b"@a\nACGT\n+\n")                                            # short record is dropped
    assert _parse(H, str(p))[0] == 0
    p.write_bytes(b"")
    assert _parse(H, str(p))[0] == 0


def _parse2(H, path, mode, threads=1, slice_bytes=0, gather=(0, 0)):
    size = os.path.getsize(path)
    bases = np.zeros(size + 1, np.uint8)
    offs = np.zeros(size // 4 + 4, np.uint64)
    names = C.create_string_buffer(size + 256)
    nb = C.c_uint64(0)
    n = H.h_parse_fastq2(os.fsencode(path), mode, threads, slice_bytes, bases.ctypes.data, bases.size, offs.ctypes.data, offs.size, names,
                         size + 256, C.byref(nb), gather[0], gather[1])
    if n == -1:
        return -1, names.value.decode()
    assert n >= 0, n
    if gather[1]:
        return n, bases[:nb.value].tobytes()
    return n, bases[:nb.value].tobytes(), offs[:n + 1].tolist(), names.value.decode().split("\n")[:-1]


def _random_fastq(rng, nrec):
    """records with the things that could fool a parallel splitter: '@' and '+' at the start of quality lines, CRLF, one-base reads"""
    out = []
    for i in range(nrec):
        ln = int(rng.integers(1, 40))
        seq = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), ln))
        q0 = [b"@", b"+", b"!", b"I"][int(rng.integers(0, 4))]
        qual = q0 + bytes(rng.integers(33, 74, ln - 1).astype(np.uint8))
        cr = b"\r" if rng.integers(0, 5) == 0 else b""
        hdr = b"@r%d" % i + [b"", b" 1:N:0:ACG", b"\tcomment", b"/2", b"   "][int(rng.integers(0, 5))]
        out.append(hdr + b"\n" + seq + cr + b"\n+" + (b"r%d" % i if rng.integers(0, 2) else b"") + b"\n" + qual + b"\n")
    return out


def test_fastq_index_equals_the_serial_parse_for_any_slicing(H, tmp_path):
    """fastq.hpp: the threaded index (what bella_hip_load_fastq uses) against the line-by-line restatement, slices down to 1 byte so
    that every slice boundary falls inside headers, bases, '+' lines and quality strings that begin with '@'"""
    rng = np.random.default_rng(5)
    p = tmp_path / "x.fastq"
    for trial in range(12):
        recs = _random_fastq(rng, int(rng.integers(1, 60)))
        data = b"".join(recs)
        kind = trial % 6
        if kind == 1: data = data[:-1]                                           # no final newline
        if kind == 2: data = data[:len(data) - len(recs[-1]) // 2]               # cut inside the last record
        if kind == 3: data = b"".join(recs[:len(recs) // 2]) + b"\n" + b"".join(recs[len(recs) // 2:])   # an empty line in the middle
        if kind == 4: data = data + b"\n\n"
        if kind == 5 and len(recs) > 2:                                          # a header without '@' after some good records
            data = b"".join(recs[:2]) + b"oops\nAC\n+\n!!\n" + b"".join(recs[2:])
        p.write_bytes(data)
        want = _parse2(H, str(p), 0)
        for threads, sl in ((1, 1), (3, 1), (4, 7), (2, 64), (8, 1 << 20), (0, 0)):
            got = _parse2(H, str(p), 1, threads, sl)
            assert got == want, (trial, threads, sl)
        if want[0] > 0 and len(want[1]) > 3:
            nb = len(want[1])
            for _ in range(5):
                o = int(rng.integers(0, nb - 1)); n = int(rng.integers(1, nb - o + 1))
                assert _parse2(H, str(p), 1, 3, 5, gather=(o, n))[1] == want[1][o:o + n]
    p.write_bytes(b"\n@a\nAC\n+\n!!\n")                                          # empty first line: nothing is read
    assert _parse2(H, str(p), 0)[0] == 0 and _parse2(H, str(p), 1, 2, 1)[0] == 0
    p.write_bytes(b"@a\nAC\n+\n!!\nbad")                                          # header of a short last record is still checked
    assert _parse2(H, str(p), 0) == _parse2(H, str(p), 1, 2, 3) and _parse2(H, str(p), 0)[0] == -1


def test_torch_read_generator_follows_the_numpy_generator_statistically():
    """bella_testkit.synth.make_reads_torch (the GPU box's fast generator of the SURVEY 8(d) sets) against make_reads: same read
    length distribution, same yield of reliable k-mers and tuples (the parameters are the contract, the PRNG is not); truth in
    the names; reproducible per seed"""
    from bella_testkit import synth
    a = synth.make_reads_torch(240, read_len=3000, err=0.15, seed=4, device="cpu", chunk_bases=100000)   # several chunks
    b = synth.make_reads(240, read_len=3000, err=0.15, seed=4)
    assert a.nreads == b.nreads == 240 and int(a.codes.max()) <= 3
    assert abs(float(a.lengths.mean()) - float(b.lengths.mean())) < 10          # 3000 * (1 - 0.045 + 0.09) = 3135
    ta, tb = synth.count_and_tuples(a, 17, 2, 8), synth.count_and_tuples(b, 17, 2, 8)
    assert abs(ta.nkmers - tb.nkmers) < 0.08 * tb.nkmers and abs(len(ta.kmer) - len(tb.kmer)) < 0.08 * len(tb.kmer)
    for nm, ln in zip(a.names[:5], a.lengths[:5]):
        idx, start, L, strand = (int(x) for x in nm[1:].split("_"))
        assert L == 3000 and strand in (0, 1) and 0 <= start
    again = synth.make_reads_torch(240, read_len=3000, err=0.15, seed=4, device="cpu", chunk_bases=100000)
    assert np.array_equal(again.codes, a.codes) and again.names == a.names
