"""GPU parity tests (run with `-m gpu` on an MI355X): the HIP path, called through the C ABI, against the oracle
(tests/_oracle.py -> oracle/bella_oracle.c) on the same inputs and against the reference's golden outputs."""
import json
import os
import sys

import numpy as np
import pytest

import _oracle as O
from bella_amd import BellaPars, Engine, api
from bella_testkit import synth
from bella_amd.api import BellaHipError
from conftest import GOLD, ROOT, load_golden, set_mode

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = Engine(0)
    yield e
    e.close()


def oracle_pairs(rs, seqs, nkmers, tk, tr, tp, k=17):
    Bc, Br, Bv = O.build_B(rs.nreads, tk, tr, tp)
    flop, colptrC, pairs = O.spgemm(seqs, nkmers, Bc, Br, Bv, k)
    return (Bc, Br, Bv), flop, colptrC, pairs


def check_pairs(got, ext, exp, lengths, k):
    assert len(got) == len(exp)
    for f in ("rid", "cid", "count", "seedH", "seedV"):
        assert np.array_equal(got[f], exp[f]), f
    if ext is not None:
        for f in ("nbins", "support", "binov"):
            assert np.array_equal(ext[f], exp[f]), f
    assert np.array_equal(api.overlap_of_seed(got, lengths, k), exp["overlap"].astype(np.int64))


def test_assembly_matches_reference_layout(eng, golden):
    g = golden
    eng.set_reads(g.rs)
    eng.assemble_tuples(g.k, g.nkmers, g.tk, g.tr, g.tp)
    got = eng.get_B()
    exp = O.build_B(g.rs.nreads, g.tk, g.tr, g.tp)
    for a, b in zip(got, exp):
        assert np.array_equal(a, b)           # colptr, k-mer ids in MergeDuplicates slot order, positions


@pytest.mark.parametrize("debug,rowlists", [(0, 0), (1, 0), (1024, 0), (1025, 0), (0, 1), (1, 1), (2048, 1), (65536, 0), (65537, 0), (65536, 1)])
def test_spgemm_pairs_bit_exact(eng, golden, debug, rowlists):
    g = golden
    set_mode(eng, debug)                       # 1 = force the global-workspace row path; 1024 = the lists of A' in order of first appearance
                                               # (default: k-mer order); 2048 = as if the row lists asked for did not fit in memory;
                                               # 65536 = B' entries with one later read carry it instead of pointing at it (default from
                                               # A' > 192 MB on: the 100k-read tests)
    eng.set_tuning("row_lists", rowlists)      # 1 = the products ready-made at assembly time (callers with repeated passes); default: every
                                               # pass expands B' x A' itself
    try:
        eng.set_reads(g.rs)
        eng.assemble_tuples(g.k, g.nkmers, g.tk, g.tr, g.tp)
        if rowlists == 1 and not debug & 2048 and len(g.tk) > 0:
            assert eng.timings().expand_ms > 0 and eng.memory().rowlist_bytes > 0
        if debug & (1024 | 2048) or len(g.tk) == 0:          # (without either, a long-list input gets the lists on its own: DESIGN 3)
            assert eng.timings().expand_ms == 0 and eng.memory().rowlist_bytes == 0
        n, flops = eng.overlap(BellaPars(skipAlignment=True, kmerSize=g.k))
        pairs, ext, colptrC = eng.get_pairs()
        _, flop, ecol, exp = oracle_pairs(g.rs, g.seqs, g.nkmers, g.tk, g.tr, g.tp, g.k)
        assert flops == int(flop.sum()) and n == len(exp)
        assert np.array_equal(colptrC, ecol.astype(np.uint64))
        check_pairs(pairs, ext, exp, g.rs.lengths, g.k)
    finally:
        set_mode(eng, 0)
        eng.set_tuning("row_lists")


@pytest.mark.parametrize("debug,rowlists", [(0, 0), (8192, 0), (0, 1), (65536, 0)])
def test_symbolic_phase_alone_matches_oracle(eng, golden, debug, rowlists):
    """bella_hip_count_pairs = estimateFLOP + estimateNNZ_Hash + prefixsum (overlap.hpp:157-276,110-146): colptrC, nnz(C) and the products
    without a numeric pass -- whole, per stage (column range) and per partition; then the numeric phase on the same context"""
    g = golden
    set_mode(eng, debug)                       # 8192: the bitmaps of the symbolic phase in global memory
    eng.set_tuning("row_lists", rowlists)
    try:
        eng.set_reads(g.rs)
        eng.assemble_tuples(g.k, g.nkmers, g.tk, g.tr, g.tp)
        pars = BellaPars(skipAlignment=True, kmerSize=g.k)
        _, flop, ecol, exp = oracle_pairs(g.rs, g.seqs, g.nkmers, g.tk, g.tr, g.tp, g.k)
        ecnt = np.diff(ecol.astype(np.int64))
        t0 = eng.timings()
        colptrC, n, flops = eng.count_pairs(pars)
        assert n == len(exp) and flops == int(flop.sum()) and np.array_equal(colptrC, ecol.astype(np.uint64))
        assert eng.count_flops(pars) == int(flop.sum())
        t1 = eng.timings()
        assert t1.symbolic_passes == t0.symbolic_passes + 1 and t1.numeric_passes == t0.numeric_passes and t1.numeric_columns == t0.numeric_columns
        nr = g.rs.nreads
        lo, hi = nr // 3, nr // 3 + max(1, nr // 2)
        eng.set_column_range(lo, hi - lo)
        eng.set_partition(1, 2)
        colptrC, n, flops = eng.count_pairs(pars)
        own = np.zeros(nr, bool); own[lo:min(hi, nr)] = True; own &= (np.arange(nr) % 2 == 1)
        assert np.array_equal(np.diff(colptrC.astype(np.int64)), np.where(own, ecnt, 0)) and n == int(ecnt[own].sum()) and flops == int(flop[own].sum())
        eng.set_column_range(0, 0xFFFFFFFF)
        eng.set_partition(0, 1)
        n, flops = eng.overlap(pars)           # the numeric phase still starts from clean arrays
        pairs, ext, colptrC = eng.get_pairs()
        assert n == len(exp) and np.array_equal(colptrC, ecol.astype(np.uint64))
        check_pairs(pairs, ext, exp, g.rs.lengths, g.k)
        assert eng.timings().numeric_columns == t1.numeric_columns + nr
    finally:
        set_mode(eng, 0)
        eng.set_tuning("row_lists")
        eng.set_column_range(0, 0xFFFFFFFF)
        eng.set_partition(0, 1)


def test_set_B_boundary_equals_tuple_assembly(eng, golden):
    """HashSpGEMM boundary: the reference's own B arrays in, same result."""
    g = golden
    Bc, Br, Bv = O.build_B(g.rs.nreads, g.tk, g.tr, g.tp)
    eng.set_reads(g.rs)
    eng.set_B(g.k, g.nkmers, Bc, Br, Bv)
    eng.overlap(BellaPars(skipAlignment=True, kmerSize=g.k))
    pairs, ext, _ = eng.get_pairs()
    _, _, _, exp = oracle_pairs(g.rs, g.seqs, g.nkmers, g.tk, g.tr, g.tp, g.k)
    check_pairs(pairs, ext, exp, g.rs.lengths, g.k)


def test_golden_files_byte_identical(eng, golden, tmp_path):
    g = golden
    eng.set_reads(g.rs)
    eng.assemble_tuples(g.k, g.nkmers, g.tk, g.tr, g.tp)
    import io
    so = io.StringIO()
    f = str(tmp_path / "o.out")
    api.hash_spgemm(eng, BellaPars(skipAlignment=True, errorRate=g.err, kmerSize=g.k), f, stdout=so)
    assert open(f, "rb").read() == g.out["skip"]
    assert so.getvalue().split()[0] == g.stdout["skip"][2]           # nnz(C), overlap.hpp:686
    for key, paf in (("align", False), ("paf", True)):
        so = io.StringIO()
        api.hash_spgemm(eng, BellaPars(errorRate=g.err, outputPaf=paf, kmerSize=g.k), f, stdout=so)
        # no tolerance: alignments that hit the reference's uninitialised maxpos (`flagged`) never pass the
        # threshold, so the files are the reference's byte for byte (toyjunk220: 12,714 of 21,915 pairs are flagged)
        assert open(f, "rb").read() == g.out[key]
        assert so.getvalue().split()[1] == g.stdout["align"][3]       # outputted, overlap.hpp:771
    if "flagged_pairs" in g.meta:
        alns = eng.get_alignments()
        assert int(alns["flagged"].sum()) == g.meta["flagged_pairs"] and len(alns) == g.meta["candidate_pairs"]
        assert not (alns["flagged"].astype(bool) & alns["passed"].astype(bool)).any()


def test_alignments_match_oracle_fieldwise(eng, golden):
    g = golden
    eng.set_reads(g.rs)
    eng.assemble_tuples(g.k, g.nkmers, g.tk, g.tr, g.tp)
    pars = BellaPars(errorRate=g.err, kmerSize=g.k)
    eng.overlap(pars)
    npass = eng.align_pairs(pars)
    pairs, _, _ = eng.get_pairs(ext=False)
    alns = eng.get_alignments()
    assert npass == int(alns["passed"].sum())
    phi = O.slope(g.err)
    for p, a in list(zip(pairs, alns))[:1500]:
        rid, cid = int(p["rid"]), int(p["cid"])
        e = O.xavier_align(g.seqs[rid], g.seqs[cid], int(p["seedH"]), int(p["seedV"]), g.xdrop, g.k)
        ok, ov = O.post_align(e["score"], e["begV"], e["endV"], e["begH"], e["endH"], len(g.seqs[rid]), len(g.seqs[cid]), phi)
        exp = (int(e["score"]), int(e["begH"]), int(e["endH"]), int(e["begV"]), int(e["endV"]), ov, int(e["strand"]), int(ok),
               int(e["steps"]), int(e["flagged"]))
        got = (int(a["score"]), int(a["begH"]), int(a["endH"]), int(a["begV"]), int(a["endV"]), int(a["ov"]), int(a["strand"]),
               int(a["passed"]), int(a["steps"]), int(a["flagged"]))
        assert got == exp, (rid, cid)
    # the other statements of the same kernel (one launch in length-sorted order; slices with compaction between launches = the default; pair order; the scalar
    # statement of xavier.h) give the same records, every field, every pair
    for variant in (0, 1, 2, 3):
        eng.set_tuning("xdrop_variant", variant)
        try:
            assert eng.align_pairs(pars) == npass
            other = eng.get_alignments()
        finally:
            eng.set_tuning("xdrop_variant")
        assert np.array_equal(other, alns), variant
    # the slices with the batch cut into four classes by step estimate (the long extensions on streams of their own, at a higher wave
    # priority): the default for batches of a million extensions, forced here
    eng.set_tuning("xdrop_class_min", 1024)
    try:
        assert eng.align_pairs(pars) == npass
        other = eng.get_alignments()
    finally:
        eng.set_tuning("xdrop_class_min")
    assert np.array_equal(other, alns)


def test_xavier_known_answers_on_gpu(eng):
    kats = json.load(open(os.path.join(GOLD, "xavier_kat.json")))
    seqs, seeds, exps = [], [], []
    for kat in kats:
        if kat["kind"] == "xdrop":
            row, col, i, j = kat["target"], kat["query"], kat["begH"], kat["begV"]
        else:
            row, col, i, j = kat["row"], kat["col"], kat["i"], kat["j"]
        seqs += [row.encode(), col.encode()]
        seeds.append((len(seqs) - 2, len(seqs) - 1, i, j, kat["k"], kat["x"]))
        exps.append(kat)
    rs = synth.readset_from_seqs(seqs)
    eng.set_reads(rs)
    for (rid, cid, i, j, k, x), kat in zip(seeds, exps):
        sd = np.zeros(1, api.SEED_DT)
        sd[0] = (rid, cid, i, j)
        a = eng.xdrop_batch(sd, BellaPars(kmerSize=k, xDrop=x))[0]
        got = [int(a["score"]), int(a["begH"]), int(a["endH"]), int(a["begV"]), int(a["endV"])]
        # every answer, flagged ones included, against the oracle (which defines the first maxpos as 0 where the reference reads an
        # uninitialised variable, xavier.h:165); the reference's own answer for the unflagged ones
        o = O.xavier_align(seqs[rid], seqs[cid], i, j, x, k)
        assert got == [int(o["score"]), int(o["begH"]), int(o["endH"]), int(o["begV"]), int(o["endV"])], kat["name"]
        assert int(a["flagged"]) == int(o["flagged"]), kat["name"]
        if not a["flagged"]:
            assert got == kat["expect"], kat["name"]
        if kat["kind"] == "align":
            assert ("c" if a["strand"] else "n") == kat["strand"]


def test_xdrop_sequence_ends_short_reads_both_strands(eng):
    """The band on and past the sequence ends (xavier.h:185-251, the terminator / pad codes of simdutils.h:191-195): many short pairs --
    40 .. 400 bases, seeds anywhere including the very ends, identical / noisy / unrelated flanks, both strands, unequal lengths -- so
    that every extension reaches Phase 4 within a few dozen steps.  Every field of every record against the oracle, all kernel
    variants (the slices forced into classes as well)."""
    rng = np.random.default_rng(20240)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    k, x = 17, 7
    seqs, seeds = [], []
    for n in range(900):
        L = int(rng.integers(40, 400))
        tpl = rng.integers(0, 4, L).astype(np.uint8)
        s0 = int(rng.integers(0, L - k + 1)) if n % 5 else (0 if n % 2 else L - k)          # every fifth seed at an end of the row
        noise = (0.0, 0.03, 0.15, 0.6)[n % 4]
        cut_l = int(rng.integers(0, s0 + 1)) if n % 3 else 0                                 # the column starts later / ends earlier than the row
        cut_r = int(rng.integers(s0 + k, L + 1)) if n % 3 == 1 else L
        col = tpl[cut_l:cut_r].copy()
        sj = s0 - cut_l
        mut = rng.random(len(col)) < noise
        mut[sj:sj + k] = False                                                              # the seed k-mer stays
        col[mut] = (col[mut] + rng.integers(1, 4, int(mut.sum()))) % 4
        row_b = bytes(b"ACGT"[c] for c in tpl)
        col_b = bytes(b"ACGT"[c] for c in col)
        if n % 2:                                                                           # the column as its reverse complement
            col_b = col_b.translate(comp)[::-1]
            sj = len(col_b) - k - sj
        if len(col_b) < k or len(row_b) < k:
            continue
        seqs += [row_b, col_b]
        seeds.append((len(seqs) - 2, len(seqs) - 1, s0, sj))
    rs = synth.readset_from_seqs(seqs)
    eng.set_reads(rs)
    sd = np.zeros(len(seeds), api.SEED_DT)
    for t, q in enumerate(seeds):
        sd[t] = q
    pars = BellaPars(kmerSize=k, xDrop=x)
    ref = eng.xdrop_batch(sd, pars)
    for (rid, cid, i, j), a in zip(seeds, ref):
        o = O.xavier_align(seqs[rid], seqs[cid], i, j, x, k)
        got = (int(a["score"]), int(a["begH"]), int(a["endH"]), int(a["begV"]), int(a["endV"]), int(a["strand"]), int(a["steps"]), int(a["flagged"]))
        exp = (int(o["score"]), int(o["begH"]), int(o["endH"]), int(o["begV"]), int(o["endV"]), int(o["strand"]), int(o["steps"]), int(o["flagged"]))
        assert got == exp, (rid, cid, i, j, len(seqs[rid]), len(seqs[cid]))
    for variant, cmin in ((0, None), (1, 64), (2, None), (3, None)):
        eng.set_tuning("xdrop_variant", variant)
        if cmin:
            eng.set_tuning("xdrop_class_min", cmin)
        try:
            other = eng.xdrop_batch(sd, pars)
        finally:
            eng.set_tuning("xdrop_variant")
            eng.set_tuning("xdrop_class_min")
        assert np.array_equal(other, ref), variant


def test_exact_xdrop_mode_matches_logan_oracle_and_seqan_answers(eng):
    """the LOGAN-equivalent scoring kernel (logan.hpp): the reference's alignSeqAn known answers, and a golden read set's candidate
    pairs (Xavier's seeds) field by field against the oracle's restatement of loganGPU/functions.cuh + PostAlignDecisionGPU"""
    kats = json.load(open(os.path.join(GOLD, "logan_kat.json")))
    seqs, seeds = [], []
    for kat in kats:
        seqs += [kat["row"].encode(), kat["col"].encode()]
        seeds.append((len(seqs) - 2, len(seqs) - 1, kat["i"], kat["j"]))
    eng.set_reads(synth.readset_from_seqs(seqs))
    for (rid, cid, i, j), kat in zip(seeds, kats):
        sd = np.zeros(1, api.SEED_DT)
        sd[0] = (rid, cid, i, j)
        a = eng.xdrop_batch(sd, BellaPars(kmerSize=kat["k"], xDrop=kat["x"]), exact=True)[0]
        assert [int(a["score"]), int(a["begH"]), int(a["endH"]), int(a["begV"]), int(a["endV"])] == kat["expect"], kat["name"]
        assert ("c" if a["strand"] else "n") == kat["strand"]
    for name in ("toy120", "toyjunk220"):
        g = load_golden(name)
        eng.set_reads(g.rs)
        eng.assemble_tuples(g.k, g.nkmers, g.tk, g.tr, g.tp)
        pars = BellaPars(errorRate=g.err, kmerSize=g.k)
        n, _ = eng.overlap(pars)
        pairs, _, _ = eng.get_pairs(ext=False)
        npass = eng.align_pairs(pars, exact=True)
        alns = eng.get_alignments()
        phi = O.slope(g.err)
        step = max(1, n // 3000)
        ok_cnt = 0
        for t in range(0, n, step):
            p, a = pairs[t], alns[t]
            rid, cid = int(p["rid"]), int(p["cid"])
            e = O.logan_align(g.seqs[rid], g.seqs[cid], int(p["seedH"]), int(p["seedV"]), g.xdrop, g.k)
            ok, ov = O.post_align_gpu(e["score"], e["begV"], e["endV"], e["begH"], e["endH"], len(g.seqs[rid]), len(g.seqs[cid]), phi)
            assert (int(a["score"]), int(a["begH"]), int(a["endH"]), int(a["begV"]), int(a["endV"]), int(a["ov"]), int(a["passed"]), int(a["strand"]),
                    int(a["steps"])) == (int(e["score"]), int(e["begH"]), int(e["endH"]), int(e["begV"]), int(e["endV"]), ov, int(ok),
                                         int(e["strand"]), int(e["steps"])), (name, t)
            ok_cnt += int(ok)
        assert npass == int(alns["passed"].sum()) and (step > 1 or ok_cnt == npass)
        assert not alns["flagged"].any()
        # the launch goes in chunks of extensions (grid x block stays below 2^32 threads at 100k-read scale): tiny chunks, same records
        set_mode(eng, 256)
        try:
            assert eng.align_pairs(pars, exact=True) == npass
            assert np.array_equal(eng.get_alignments(), alns)
        finally:
            set_mode(eng, 0)


@pytest.mark.parametrize("layout", [0, 65536])
def test_partition_union_equals_whole(eng, layout):
    g = load_golden("toy120")
    set_mode(eng, layout)                                  # 65536: one-partner B' entries in the inline form (the default of sets whose A' is
                                                           # larger than the cache -- what every rank of a multi-GPU run of 100k reads has),
                                                           # here together with the partitioned layout (B' entries of the owned columns only)
    eng.set_reads(g.rs)
    eng.assemble_tuples(g.k, g.nkmers, g.tk, g.tr, g.tp)
    eng.set_partition(0, 1)
    eng.overlap(BellaPars())
    whole, _, _ = eng.get_pairs(ext=False)
    parts = []
    try:
        for r in range(3):
            eng.set_partition(r, 3)
            eng.overlap(BellaPars())
            p, _, _ = eng.get_pairs(ext=False)
            assert (p["cid"] % 3 == r).all()
            parts.append(p)
    finally:
        eng.set_partition(0, 1)
        set_mode(eng, 0)
    merged = np.concatenate(parts)
    order = np.argsort(merged["cid"], kind="stable")      # columns ascending, slot order kept inside a column
    assert np.array_equal(merged[order], whole)


def test_partitioned_layout_of_a_matrix_with_fewer_nonzeros_than_reads(eng):
    """the partitioned layout keeps one word per READ in a scratch array that used to be sized by nnz(A) alone:
    nnz == 0 and nnz << nreads, several partitions, with and without the row lists"""
    nr = 3000
    rs = synth.readset_from_seqs([b"ACGTTGCA" * 6] * nr)
    eng.set_reads(rs)
    tk = np.array([3, 3, 4, 4, 4], np.uint32); tr = np.array([5, 2900, 17, 1200, 2999], np.uint32); tp = np.array([0, 4, 8, 4, 12], np.uint16)
    order = np.argsort(tr, kind="stable")
    tk, tr, tp = tk[order], tr[order], tp[order]
    try:
        for rl in (0, 1):
            eng.set_tuning("row_lists", rl)
            for ntup in (0, len(tk)):
                got = []
                for r in range(3):
                    eng.set_partition(r, 3)                 # BEFORE the operands: the layout is built for the partition
                    eng.assemble_tuples(17, 9, tk[:ntup], tr[:ntup], tp[:ntup])
                    n, _ = eng.overlap(BellaPars(skipAlignment=True))
                    p, _, _ = eng.get_pairs(ext=False)
                    assert n == len(p) and (p["cid"] % 3 == r).all()
                    got += [(int(x["cid"]), int(x["rid"]), int(x["count"])) for x in p]
                exp = [] if ntup == 0 else [(5, 2900, 1), (17, 1200, 1), (17, 2999, 1), (1200, 2999, 1)]
                assert sorted(got) == exp
    finally:
        eng.set_partition(0, 1)
        eng.set_tuning("row_lists")


def test_reserved_slab_serves_the_stages_and_changes_nothing(golden):
    """bella_hip_reserve: the stages cut their buffers from one slab taken up front; same results as without; a slab in use is neither
    replaced nor given back; what does not fit the slab is allocated as before; bella_hip_trim is harmless at any time"""
    g = golden
    e = Engine(0)
    try:
        ms = e.reserve(192 << 20)
        assert ms > 0.0
        assert e.reserve(64 << 20) == 0.0                  # a smaller request is served by the slab that exists
        e.set_reads(g.rs)
        e.assemble_tuples(g.k, g.nkmers, g.tk, g.tr, g.tp)
        n, flops = e.overlap(BellaPars(skipAlignment=True, kmerSize=g.k))
        pairs, ext, _ = e.get_pairs()
        _, flop, _, exp = oracle_pairs(g.rs, g.seqs, g.nkmers, g.tk, g.tr, g.tp, g.k)
        assert flops == int(flop.sum()) and n == len(exp)
        check_pairs(pairs, ext, exp, g.rs.lengths, g.k)
        with pytest.raises(BellaHipError) as err:          # buffers live in the slab: it stays
            e.reserve(1 << 30)
        assert err.value.code == -7                        # BELLA_ERR_STATE
        e.trim()
        e.set_tuning("kcount_budget", 1 << 30)
        if g.nkmers and not g.syncmer and not g.window:    # a stage that outgrows the 192 MB slab falls back to the driver
            e.count_kmers(g.k, g.lower, g.upper)
            e.assemble_counted()
            n2, _ = e.overlap(BellaPars(skipAlignment=True, kmerSize=g.k))
            assert n2 == n
    finally:
        e.close()


def test_medium_synthetic_vs_oracle(eng):
    """2,000 reads x 6 kb (SpGEMM) -- bigger hash tables, LDS tiers and the ordering emulation under load"""
    rs = synth.make_reads(2000, read_len=6000, err=0.15, seed=21)
    t = synth.count_and_tuples(rs, 17, 2, 8)
    seqs = rs.seqs()
    eng.set_reads(rs)
    eng.assemble_tuples(17, t.nkmers, t.kmer, t.read, t.pos)
    got = eng.get_B()
    (Bc, Br, Bv), flop, ecol, exp = oracle_pairs(rs, seqs, t.nkmers, t.kmer, t.read, t.pos)
    for a, b in zip(got, (Bc, Br, Bv)):
        assert np.array_equal(a, b)
    n, flops = eng.overlap(BellaPars())
    assert n == len(exp) and flops == int(flop.sum())
    pairs, ext, _ = eng.get_pairs()
    check_pairs(pairs, ext, exp, rs.lengths, 17)
    # idempotence: a second pass over the same resident operands gives the same bytes
    eng.overlap(BellaPars())
    again, _, _ = eng.get_pairs()
    assert again.tobytes() == pairs.tobytes()
    # X-drop on a sample of the pairs
    eng.align_pairs(BellaPars())
    alns = eng.get_alignments()
    phi = O.slope(0.15)
    idx = np.random.default_rng(0).choice(len(pairs), size=300, replace=False)
    for n_ in idx:
        p, a = pairs[n_], alns[n_]
        rid, cid = int(p["rid"]), int(p["cid"])
        e = O.xavier_align(seqs[rid], seqs[cid], int(p["seedH"]), int(p["seedV"]), 7, 17)
        ok, ov = O.post_align(e["score"], e["begV"], e["endV"], e["begH"], e["endH"], len(seqs[rid]), len(seqs[cid]), phi)
        assert (int(a["score"]), int(a["begH"]), int(a["endH"]), int(a["begV"]), int(a["endV"]), int(a["ov"]), int(a["passed"])) == \
               (int(e["score"]), int(e["begH"]), int(e["endH"]), int(e["begV"]), int(e["endV"]), ov, int(ok))


def test_edge_cases(eng):
    # empty matrix, reads without tuples, a single pair
    rs = synth.readset_from_seqs([b"ACGT" * 20, b"TTTT" * 30, b"ACGT" * 20])
    eng.set_reads(rs)
    eng.assemble_tuples(17, 5, np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.uint16))
    assert eng.overlap(BellaPars()) == (0, 0)
    eng.assemble_tuples(17, 5, np.array([3, 3], np.uint32), np.array([0, 2], np.uint32), np.array([0, 4], np.uint16))
    assert eng.overlap(BellaPars())[0] == 1
    p, _, _ = eng.get_pairs()
    assert (int(p["rid"][0]), int(p["cid"][0]), int(p["count"][0]), int(p["seedH"][0]), int(p["seedV"][0])) == (2, 0, 1, 4, 0)
    # duplicated k-mer inside a read: LAST position wins, slot decided by FIRST occurrence (CSC.cpp:344)
    tk = np.array([7, 9, 7, 9, 7], np.uint32); tr = np.array([0, 0, 0, 2, 2], np.uint32); tp = np.array([0, 4, 8, 4, 12], np.uint16)
    eng.assemble_tuples(17, 10, tk, tr, tp)
    exp = O.build_B(3, tk, tr, tp)
    for a, b in zip(eng.get_B(), exp):
        assert np.array_equal(a, b)


def test_errors_are_loud(eng):
    with pytest.raises(BellaHipError) as e:
        eng.set_reads_raw(np.frombuffer(b"ACGTNACGT", np.uint8), np.array([0, 9], np.uint64))
    assert e.value.code == -4
    with pytest.raises(BellaHipError) as e:
        eng.set_reads_raw(np.frombuffer(b"A" * 70000, np.uint8), np.array([0, 70000], np.uint64))
    assert e.value.code == -5
    rs = synth.readset_from_seqs([b"ACGT" * 20, b"ACGT" * 20])
    eng.set_reads(rs)
    with pytest.raises(BellaHipError) as e:
        eng.assemble_tuples(17, 5, np.array([1, 1], np.uint32), np.array([1, 0], np.uint32), np.array([0, 0], np.uint16))
    assert e.value.code == -6
    with pytest.raises(BellaHipError):
        eng.overlap(BellaPars())              # no matrix after the failed assembly


def test_bad_operands_are_rejected_before_they_are_used(eng):
    """k-mer id >= nkmers, a k-mer running past the end of its read, a k-mer in more than 16383 reads, a malformed colptr:
    BELLA_ERR_BAD_ARG (-3), no device fault, and the context keeps working afterwards"""
    rs = synth.readset_from_seqs([b"ACGT" * 20, b"ACGT" * 20, b"ACGT" * 20])
    eng.set_reads(rs)
    with pytest.raises(BellaHipError) as e:       # id 5 with nkmers = 5
        eng.assemble_tuples(17, 5, np.array([5, 1], np.uint32), np.array([0, 1], np.uint32), np.array([0, 0], np.uint16))
    assert e.value.code == -3 and "k-mer id" in str(e.value)
    with pytest.raises(BellaHipError) as e:       # position 70 + 17 > 80
        eng.assemble_tuples(17, 5, np.array([1, 1], np.uint32), np.array([0, 1], np.uint32), np.array([70, 0], np.uint16))
    assert e.value.code == -3 and "length" in str(e.value)
    with pytest.raises(BellaHipError):
        eng.overlap(BellaPars())
    with pytest.raises(BellaHipError) as e:       # set_B: colptr not monotone / not starting at 0
        eng.set_B(17, 5, np.array([0, 2, 1, 2], np.uint32), np.array([1, 2], np.uint32), np.array([0, 0], np.uint16))
    assert e.value.code == -3
    with pytest.raises(BellaHipError) as e:
        eng.set_B(17, 5, np.array([1, 1, 2, 2], np.uint32), np.array([1, 2], np.uint32), np.array([0, 0], np.uint16))
    assert e.value.code == -3
    n = 16400                                      # one k-mer shared by 16400 reads: the 14-bit product count of B' overflows
    rs = synth.readset_from_seqs([b"ACGTACGTACGTACGTACGTAAAA"] * n)
    eng.set_reads(rs)
    with pytest.raises(BellaHipError) as e:
        eng.assemble_tuples(17, 3, np.full(n, 2, np.uint32), np.arange(n, dtype=np.uint32), np.zeros(n, np.uint16))
    assert e.value.code == -3 and "16383" in str(e.value)
    with pytest.raises(BellaHipError) as e:       # parameters outside their domain
        eng.assemble_tuples(17, 3, np.array([2, 2], np.uint32), np.array([0, 1], np.uint32), np.zeros(2, np.uint16))
        eng.overlap(BellaPars(errorRate=1.5))
    assert e.value.code == -3
    assert eng.overlap(BellaPars())[0] == 1       # and the context still works


def test_dropin_shim_from_reference_call_site(eng, tmp_path):
    """oracle/_ref/libbella_dropin.so = the reference's headers + its own HashSpGEMM call (main.cpp:498-525) compiled with
    bella_amd/host/bella_hip_shim.hpp: the call must land in libbella_hip.so and write the golden file."""
    import ctypes as C
    from conftest import ROOT
    path = os.path.join(ROOT, "oracle", "_ref", "libbella_dropin.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libbella_dropin.so not built (needs /root/reference at build time)")
    lib = C.CDLL(path)
    u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
    u16p = np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")
    lib.bella_dropin_hashspgemm.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64, u32p, u32p, u16p, C.POINTER(C.c_char_p),
                                            C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                            C.c_double, C.c_char_p, C.c_char_p, C.c_size_t]
    for name in ("sanity3", "toy120", "toyjunk220"):
        g = load_golden(name)
        n = g.rs.nreads
        sarr = (C.c_char_p * n)(*g.seqs)
        narr = (C.c_char_p * n)(*[x.encode() for x in g.names])
        for skip, key in ((1, "skip"), (0, "align")):
            so = C.create_string_buffer(4096)
            f = str(tmp_path / ("%s_%s.out" % (name, key)))
            lib.bella_dropin_hashspgemm(n, g.nkmers, len(g.tk), g.tk, g.tr, g.tp, sarr, narr, g.k, 500, 7, skip, 0, g.err, 0.1,
                                        f.encode(), so, len(so))
            nums = so.value.decode().split()
            assert nums[:3] == g.stdout[key][:3]            # nkmer, nnz(A) after merge, nnz(C): the stdout protocol
            assert open(f, "rb").read() == g.out[key]       # the reference's file, byte for byte (no tolerance)


def _run_cli(binary, fastqs, flags, cwd, env_extra=None):
    """the reference's CLI contract (main.cpp:65-175): -f names a newline-terminated list of FASTQ files, -o the output stem"""
    import subprocess
    import re
    os.makedirs(cwd, exist_ok=True)
    with open(os.path.join(cwd, "in.txt"), "w") as f:
        f.write("".join(p + "\n" for p in fastqs))            # (kmercount.hpp:96: every line must end in '\n')
    env = dict(os.environ, OMP_NUM_THREADS="1")                # 1-thread libcuckoo ids are the golden ids
    env.update(env_extra or {})
    p = subprocess.run([os.path.join(ROOT, "oracle", "_ref", binary), "-f", "in.txt", "-o", "out"] + list(flags), cwd=cwd, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    so = p.stdout.decode(errors="replace")
    nums = [ln.strip() for ln in so.splitlines() if re.fullmatch(r"[0-9.eE+-]+", ln.strip())]
    out = os.path.join(cwd, "out.out")
    assert os.path.exists(out), (p.returncode, so[-2000:], p.stderr.decode(errors="replace")[-2000:])
    return nums, open(out, "rb").read(), p.stderr.decode(errors="replace")


def test_reference_cli_built_from_its_unchanged_main_runs_on_the_gpu(golden, tmp_path):
    """oracle/_ref/bella_dropin = the reference's src/main.cpp with the ONE include line of INTEGRATION.md section 1, linked
    against libbella_hip.so (oracle/build_ref.sh): flags (main.cpp:65-175), the FASTQ list, the reference's own k-mer counting and
    CSC constructor on the host, then the unchanged call at main.cpp:498-525 -> the shim -> the C ABI.  Output file and the stdout
    protocol (nkmer, nnz(A), nnz(C), outputted: main.cpp:472-473, overlap.hpp:686,771) against the reference binary's goldens."""
    import gzip
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "bella_dropin")):
        pytest.skip("oracle/_ref/bella_dropin not built (needs /root/reference at build time)")
    g = golden
    fq = str(tmp_path / "reads.fastq")
    with gzip.open(os.path.join(GOLD, g.name, "reads.fastq.gz"), "rb") as src, open(fq, "wb") as dst:
        dst.write(src.read())
    for extra, key in ((["--skip-alignment"], "skip"), ([], "align"), (["--paf"], "paf")):
        nums, data, err = _run_cli("bella_dropin", [fq], g.meta["flags"] + extra, str(tmp_path / key))
        assert data == g.out[key], (g.name, key)                # byte for byte, no tolerance
        if key in g.stdout:
            assert nums[:-1] == g.stdout[key][:-1], (nums, g.stdout[key])   # every protocol line but the last (the run time)
        assert "bella_hip_shim.hpp" in err                      # the shim's log lines: the call really went through it


def test_reference_cli_on_the_gpu_with_a_fastq_list_several_gpus_and_stages(tmp_path):
    """the same executable with what the small goldens cannot exercise: a -f list of TWO files (kmercount.hpp:82-105 GetFiles), -g 2
    (two contexts sharing the one GPU) and a -m that forces the reference's stage loop (overlap.hpp:682-710) -- against the reference
    binary itself (oracle/_ref/bella_ref, single stage: its multi-stage run overwrites the file from offset 0, overlap.hpp:613-636)"""
    for b in ("bella_dropin", "bella_ref"):
        if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", b)):
            pytest.skip("oracle/_ref/%s not built" % b)
    rs = synth.make_reads(2200, read_len=4000, err=0.15, seed=77)
    fa, fb = str(tmp_path / "a.fastq"), str(tmp_path / "b.fastq")
    synth.write_fastq(fa, rs.subset(900))
    tail = synth.ReadSet(rs.codes[rs.offsets[900]:], rs.offsets[900:] - rs.offsets[900], rs.names[900:])
    synth.write_fastq(fb, tail)
    over = {"BELLA_HIP_SHIM_OVERSUBSCRIBE": "1"}
    for extra, key, variants in ((["--skip-alignment"], "skip", (([], None), (["-g", "2"], over), (["-m", "1"], None), (["-m", "1", "-g", "2"], over))),
                                 ([], "align", ((["-m", "1", "-g", "2"], over),))):
        rnums, ref, _ = _run_cli("bella_ref", [fa, fb], extra, str(tmp_path / ("ref_" + key)))
        nnzc = int(rnums[2])
        assert nnzc > 53000                                     # 1.5 * nnzc * 20 B > 1 MB: -m 1 means at least two stages
        stages = int(np.ceil(1.5 * nnzc * 20 / (1024.0 * 1024.0)))
        assert stages >= 2
        for flags, env in variants:
            nums, data, err = _run_cli("bella_dropin", [fa, fb], extra + flags, str(tmp_path / ("d_%s_%s" % (key, "_".join(flags)))), env)
            assert data == ref, (key, flags)
            assert nums[:3] == rnums[:3]
            if "-m" in flags:
                assert err.count("ColumnsRange") == stages, err[-1500:]
            if key == "align":                                  # "outputted" is printed per stage (overlap.hpp:771)
                assert sum(int(x) for x in nums[3:-1]) == int(rnums[3])


def _run_native(fastqs, flags, cwd, env_extra=None):
    """bella_amd/bin/bella-hip (bella_amd/host/bella_hip_main.cpp): the reference's CLI contract, the whole pipeline on the device"""
    import subprocess
    import re
    exe = os.path.join(ROOT, "bella_amd", "bin", "bella-hip")
    assert os.path.exists(exe), "bella_amd/bin/bella-hip is not built (bella_amd/build.py builds it)"
    os.makedirs(cwd, exist_ok=True)
    with open(os.path.join(cwd, "in.txt"), "w") as f:
        f.write("".join(p + "\n" for p in fastqs))
    env = dict(os.environ)
    env.update(env_extra or {})
    p = subprocess.run([exe, "-f", "in.txt", "-o", "out"] + list(flags), cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    so, se = p.stdout.decode(errors="replace"), p.stderr.decode(errors="replace")
    assert p.returncode == 0, (p.returncode, so[-2000:], se[-2000:])
    nums = [ln.strip() for ln in so.splitlines() if re.fullmatch(r"[0-9.eE+-]+", ln.strip())]
    out = os.path.join(cwd, "out.out")
    assert os.path.exists(out), (so[-2000:], se[-2000:])
    return nums, open(out, "rb").read(), se


def _write_mtx(path, g):
    """the reference's readbykmers.mtx dump (include/common/bellaio.h:2-47) from a golden set's tuples: pins the k-mer ids"""
    with open(path, "w") as f:
        f.write("%d\t%d\t%d\n" % (g.rs.nreads, g.nkmers, len(g.tk)))
        f.write("".join("%d\t%d\t%d\n" % (r + 1, k + 1, q) for k, r, q in zip(g.tk.tolist(), g.tr.tolist(), g.tp.tolist())))


def test_native_cli_with_the_reference_ids_is_byte_identical(golden, tmp_path):
    """bella-hip with --tuples <readbykmers.mtx of the reference run>: FASTQ ingest, assembly, SpGEMM, X-drop and the writer on the
    device behind the reference's flags -- the three output files and the stdout protocol (nkmer, nnz(A), nnz(C), outputted:
    main.cpp:472-473, CSC.cpp:405, overlap.hpp:686,771) byte for byte against the reference binary's goldens"""
    import gzip
    g = golden
    fq = str(tmp_path / "reads.fastq")
    with gzip.open(os.path.join(GOLD, g.name, "reads.fastq.gz"), "rb") as src, open(fq, "wb") as dst:
        dst.write(src.read())
    mtx = str(tmp_path / "readbykmers.mtx")
    _write_mtx(mtx, g)
    for extra, key in ((["--skip-alignment"], "skip"), ([], "align"), (["--paf"], "paf")):
        nums, data, err = _run_native([fq], g.meta["flags"] + extra + ["--tuples", mtx], str(tmp_path / key))
        assert data == g.out[key], (g.name, key)                # byte for byte, no tolerance
        if key in g.stdout:
            assert nums[:-1] == g.stdout[key][:-1], (nums, g.stdout[key])
        assert "bella_hip_main.cpp" in err and "ColumnsRange" in err


def test_native_cli_counts_on_the_device(golden, tmp_path):
    """the same executable without --tuples: k-mer counting / syncmer / minimizer selection, the dictionary and the tuples on the device
    (ids = ascending canonical words: labels).  Same dictionary size, same nnz(A), same nnz(C), the same candidate pair set as the
    reference; with alignment the few pairs whose seed follows the fold order may pass or fail differently (as in the reference between
    thread counts) and the quality evaluation agrees."""
    import gzip
    from bella_amd import evaluate as ev
    g = golden
    fq = str(tmp_path / "reads.fastq")
    with gzip.open(os.path.join(GOLD, g.name, "reads.fastq.gz"), "rb") as src, open(fq, "wb") as dst:
        dst.write(src.read())
    key = lambda data: {tuple(ln.split(b"\t")[:2]) for ln in data.split(b"\n") if ln}
    nums, data, _ = _run_native([fq], g.meta["flags"] + ["--skip-alignment"], str(tmp_path / "skip"))
    assert nums[:3] == g.stdout["skip"][:3], (nums, g.stdout["skip"])
    assert key(data) == key(g.out["skip"]) and len(data.splitlines()) == len(g.out["skip"].splitlines())
    nums, data, _ = _run_native([fq], g.meta["flags"], str(tmp_path / "align"))
    assert nums[:3] == g.stdout["align"][:3]
    mine, ref = key(data), key(g.out["align"])
    # (toy120: 3 %; the low-error set with its long product lists -- every pair has tens of shared k-mers, the seed is whichever the fold
    # order leaves on top, and -e 0.005 leaves the pass test no slack -- 14 %)
    assert len(mine ^ ref) <= max(2, 0.2 * len(ref)), (len(mine), len(ref), len(mine ^ ref))
    if g.name.startswith("toy") and g.name != "toyjunk220":
        try:
            G = ev.truth_pairs(ev.truth_from_names(g.names), 500)
        except Exception:
            G = None
        if G:
            r1, r2 = ev.evaluate(ev.read_bella_output(data, 500), G), ev.evaluate(ev.read_bella_output(g.out["align"], 500), G)
            assert abs(r1["recall"] - r2["recall"]) < 3.0 and abs(r1["precision"] - r2["precision"]) < 3.0   # (per cent, on sets of 50 .. 220 reads)


def test_native_cli_list_of_files_several_contexts_and_stages(tmp_path):
    """bella-hip on a -f list of TWO files, with -g 2 (two contexts sharing the one GPU: BELLA_HIP_OVERSUBSCRIBE) and a -m that forces
    the reference's stage loop (overlap.hpp:682-710): the same output file as its own single-context single-stage run, and the same
    protocol numbers; flags the reference has and this program does not build are refused"""
    import subprocess
    rs = synth.make_reads(2200, read_len=4000, err=0.15, seed=77)
    fa, fb = str(tmp_path / "a.fastq"), str(tmp_path / "b.fastq")
    synth.write_fastq(fa, rs.subset(900))
    tail = synth.ReadSet(rs.codes[rs.offsets[900]:], rs.offsets[900:] - rs.offsets[900], rs.names[900:])
    synth.write_fastq(fb, tail)
    over = {"BELLA_HIP_OVERSUBSCRIBE": "1"}
    for extra, key in ((["--skip-alignment"], "skip"), ([], "align")):
        bnums, base, _ = _run_native([fa, fb], extra, str(tmp_path / ("b_" + key)))
        nnzc = int(bnums[2])
        stages = int(np.ceil(1.5 * nnzc * 20 / (1024.0 * 1024.0)))
        assert stages >= 2
        for flags, env in (([("-g"), "2"], over), (["-m", "1"], None), (["-m", "1", "-g", "2"], over)):
            nums, data, err = _run_native([fa, fb], extra + flags, str(tmp_path / ("v_%s_%s" % (key, "_".join(flags)))), env)
            assert data == base, (key, flags)
            assert nums[:3] == bnums[:3]
            if "-m" in flags:
                assert err.count("ColumnsRange") == stages
            if key == "align":
                assert sum(int(x) for x in nums[3:-1]) == int(bnums[3])
    exe = os.path.join(ROOT, "bella_amd", "bin", "bella-hip")
    for bad in (["--hopc"], ["--estimate"], ["--split-count", "2"], ["--no-such-flag"]):
        p = subprocess.run([exe, "-f", "in.txt", "-o", "x"] + bad, cwd=str(tmp_path / "b_skip"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
        assert p.returncode != 0 and b"bella-hip:" in p.stderr, bad


def test_dropin_shim_stages_and_gpus_from_bellapars(eng, tmp_path, monkeypatch):
    """BELLApars::totalMemory (-m) -> the reference's stage count and boundaries (overlap.hpp:365-404,682-710); BELLApars::numGPU
    (-g) -> that many contexts (here sharing the one GPU): the file stays the reference's single-stage file, byte for byte"""
    import ctypes as C
    path = os.path.join(ROOT, "oracle", "_ref", "libbella_dropin.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libbella_dropin.so not built")
    monkeypatch.setenv("BELLA_HIP_SHIM_OVERSUBSCRIBE", "1")
    lib = C.CDLL(path)
    u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
    u16p = np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")
    lib.bella_dropin_hashspgemm2.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64, u32p, u32p, u16p, C.POINTER(C.c_char_p),
                                             C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                             C.c_double, C.c_int, C.c_double, C.c_char_p, C.c_char_p, C.c_size_t]
    lib.bella_dropin_last_stats.argtypes = [C.POINTER(C.c_uint64)]
    g = load_golden("toy120")
    n = g.rs.nreads
    sarr = (C.c_char_p * n)(*g.seqs)
    narr = (C.c_char_p * n)(*[x.encode() for x in g.names])
    nnzc = int(g.stdout["skip"][2])
    for ngpu, stages in ((1, 1), (1, 3), (2, 1), (3, 4)):
        # required = 1.5 * nnzc * (16 + 4) bytes (overlap.hpp:682); a budget of required / (stages - 0.5) gives `stages` stages
        mem_mb = 400000.0 if stages == 1 else 1.5 * nnzc * 20 / (stages - 0.5) / (1024 * 1024)
        for skip, key in ((1, "skip"), (0, "align")):
            so = C.create_string_buffer(4096)
            f = str(tmp_path / ("g%d_s%d_%s.out" % (ngpu, stages, key)))
            lib.bella_dropin_hashspgemm2(n, g.nkmers, len(g.tk), g.tk, g.tr, g.tp, sarr, narr, g.k, 500, 7, skip, 0, g.err, 0.1, ngpu, mem_mb,
                                         f.encode(), so, len(so))
            nums = so.value.decode().split()
            assert nums[:3] == g.stdout[key][:3]
            assert open(f, "rb").read() == g.out[key], (ngpu, stages, key)
            st = (C.c_uint64 * 9)()
            lib.bella_dropin_last_stats(st)
            # every column is computed by the numeric phase exactly ONCE; the symbolic phase runs only when one stage is not certain
            assert st[0] == n and st[3] == n and st[4] == stages and st[5] == ngpu, list(st)
            assert st[2] == (ngpu if stages > 1 else 0) and st[1] == ngpu * stages, list(st)
            if ngpu > 1:   # B' follows the partition: no context holds (much) more than its share of the entries
                assert st[6] <= 1.3 * st[7] / ngpu + 2048, list(st)
            if not skip:
                assert len(nums) == 3 + stages and sum(int(x) for x in nums[3:]) == int(g.stdout["align"][3])   # outputted per stage (:771)


def test_dropin_shim_align_call_surface(eng):
    """bella_hip::xavierAlign (align.hpp:152) and the alignLogan-shaped batch (align.hpp:210-211) of the shim, compiled against
    the reference's headers, on the reference's known answers"""
    import ctypes as C
    path = os.path.join(ROOT, "oracle", "_ref", "libbella_dropin.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libbella_dropin.so not built")
    lib = C.CDLL(path)
    kats = [k for k in json.load(open(os.path.join(ROOT, "tests", "golden", "xavier_kat.json"))) if k["kind"] == "align"]
    assert len(kats) >= 10
    oracle_flag = [int(O.xavier_align(k["row"].encode(), k["col"].encode(), k["i"], k["j"], k["x"], k["k"])["flagged"]) for k in kats]
    for kat, fl in zip(kats[:6], oracle_flag):                         # one call per pair
        out = (C.c_int * 5)()
        st = C.create_string_buffer(2)
        lib.bella_dropin_xavier_align(kat["row"].encode(), kat["col"].encode(), kat["i"], kat["j"], kat["x"], kat["k"], out, st)
        if not fl:
            assert list(out) == kat["expect"] and st.value.decode() == kat["strand"], kat["name"]
    # the per-pair form on RESIDENT reads: use_reads once, then read ids and seeds only (the reads are uploaded once per thread and run)
    g = load_golden("toy120")
    Bc, Br, Bv = O.build_B(g.rs.nreads, g.tk, g.tr, g.tp)
    _, _, op = O.spgemm(g.seqs, g.nkmers, Bc, Br, Bv, g.k)
    sel = op[:: max(1, len(op) // 40)][:40]
    m = len(sel)
    u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
    i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
    lib.bella_dropin_xavier_align_resident.argtypes = [C.c_uint32, C.POINTER(C.c_char_p), C.c_int, u32p, u32p, i32p, i32p, C.c_int, C.c_int, i32p, C.c_char_p]
    sarr = (C.c_char_p * g.rs.nreads)(*g.seqs)
    res = np.zeros(5 * m, np.int32)
    stb = C.create_string_buffer(m + 1)
    lib.bella_dropin_xavier_align_resident(g.rs.nreads, sarr, m, np.ascontiguousarray(sel["rid"], np.uint32), np.ascontiguousarray(sel["cid"], np.uint32),
                                           np.ascontiguousarray(sel["seedH"], np.int32), np.ascontiguousarray(sel["seedV"], np.int32), g.xdrop, g.k, res, stb)
    nchk = 0
    for t, pr in enumerate(sel):
        e = O.xavier_align(g.seqs[int(pr["rid"])], g.seqs[int(pr["cid"])], int(pr["seedH"]), int(pr["seedV"]), g.xdrop, g.k)
        if e["flagged"]:
            continue
        assert list(res[5 * t:5 * t + 5]) == [int(e["score"]), int(e["begH"]), int(e["endH"]), int(e["begV"]), int(e["endV"])], t
        assert stb.raw[t:t + 1] == (b"c" if e["strand"] else b"n")
        nchk += 1
    assert nchk >= 20
    same = [k for k in kats if (k["x"], k["k"]) == (kats[0]["x"], kats[0]["k"])]
    n = len(same)
    rows = (C.c_char_p * n)(*[k["row"].encode() for k in same])
    cols = (C.c_char_p * n)(*[k["col"].encode() for k in same])
    is_ = (C.c_int * n)(*[k["i"] for k in same])
    js_ = (C.c_int * n)(*[k["j"] for k in same])
    out = (C.c_int * (5 * n))()
    st = C.create_string_buffer(n + 1)
    lib.bella_dropin_align_batch(n, rows, cols, is_, js_, same[0]["x"], same[0]["k"], out, st)
    checked = 0
    for t, kat in enumerate(same):
        if O.xavier_align(kat["row"].encode(), kat["col"].encode(), kat["i"], kat["j"], kat["x"], kat["k"])["flagged"]:
            continue
        assert list(out[5 * t:5 * t + 5]) == kat["expect"] and st.raw[t:t + 1].decode() == kat["strand"], kat["name"]
        checked += 1
    assert checked >= 8


def _run_constructed(eng, lens, per_read, nkmers, k=17, seed=0):
    """reads of random bases with the given lengths; per_read[r] = [(kmer, pos), ...] in position order"""
    rng = np.random.default_rng(seed)
    arrs = [synth.BASES[rng.integers(0, 4, size=L, dtype=np.uint8)].copy() for L in lens]
    kseq = {}
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    tk, tr, tp = [], [], []
    for r, lst in enumerate(per_read):
        for km, pos in lst:
            if km not in kseq:
                kseq[km] = synth.BASES[rng.integers(0, 4, size=k, dtype=np.uint8)].copy()
            w = kseq[km]
            if rng.integers(0, 2):                      # plant the k-mer or its reverse complement (same canonical k-mer)
                w = np.asarray([comp[int(c)] for c in w[::-1]], np.uint8)
            arrs[r][pos:pos + k] = w
            tk.append(km); tr.append(r); tp.append(pos)
    seqs = [bytes(a) for a in arrs]
    rs = synth.readset_from_seqs(seqs)
    tk, tr, tp = np.asarray(tk, np.uint32), np.asarray(tr, np.uint32), np.asarray(tp, np.uint16)
    eng.set_reads(rs)
    eng.assemble_tuples(k, nkmers, tk, tr, tp)
    n, flops = eng.overlap(BellaPars(kmerSize=k))
    pairs, ext, colptrC = eng.get_pairs()
    _, flop, ecol, exp = oracle_pairs(rs, seqs, nkmers, tk, tr, tp, k)
    assert n == len(exp) and flops == int(flop.sum())
    check_pairs(pairs, ext, exp, rs.lengths, k)
    return exp


def test_key_table_overflow_is_retried_on_global_path(eng):
    """a column with as many pairs as products (500 > cap/2 of its LDS tier): rerun on the global path"""
    n = 501
    per_read = [[(j, 30 * j) for j in range(500)]] + [[(j, 5)] for j in range(500)]
    exp = _run_constructed(eng, [16000] + [100] * 500, per_read, 500)
    assert len(exp) == 500 and (exp["count"] == 1).all()


def test_more_multi_product_pairs_than_the_dense_list_holds_stay_in_lds(eng):
    """a column whose 750 pairs ALL have two products: the dense list of multi-product pairs (dcap / 2 = 400 entries in its tier) overflows
    while the key table (800) does not -- the per-pair passes fall back to the slot scan inside the kernel; round 5 abandoned such a
    column to the global-workspace rerun after its single-product records were written (a silent performance cliff)"""
    per_read = [[(j, 30 * j) for j in range(1500)]] + [[(2 * j, 5), (2 * j + 1, 40)] for j in range(750)]
    exp = _run_constructed(eng, [46000] + [100] * 750, per_read, 1500)
    assert len(exp) == 750 and (exp["count"] >= 2).all()
    assert eng.timings().retry_columns == 0


def test_many_bins_overflow_path_and_std_sort_order(eng):
    """one pair whose products all land in different overlap bins (40 bins > 8 held in LDS, > 16 => libstdc++ introsort
    tie order), plus a pair with a long single-bin state (> 64 positions): both go through k_fold_overflow"""
    a = [(t, 600 * t + 200) for t in range(40)]                  # read 0: k-mers 0..39, far apart
    b = [(t, 100 + 0 * t) for t in range(1)]                     # placeholder, replaced below
    # read 1 holds the same 40 k-mers all near its start -> overlap estimates 600 apart -> 40 orphan bins
    b = sorted([(t, 100 + 19 * t) for t in range(40)], key=lambda x: x[1])
    # reads 2,3: 120 shared k-mers, positions 40 apart on both (never within k of each other) -> one bin, 120 positions
    c = [(100 + t, 50 + 40 * t) for t in range(120)]
    d = [(100 + t, 60 + 40 * t) for t in range(120)]
    exp = _run_constructed(eng, [30000, 30000, 6000, 6000], [a, b, c, d], 300)
    by = {(int(p["rid"]), int(p["cid"])): p for p in exp}
    assert by[(1, 0)]["nbins"] > 16 and by[(3, 2)]["support"] > 64


def test_panel_assembly_and_device_set_B(eng):
    """multi-GPU assembly path on one GPU: two row-block panels assembled separately, concatenated on the device (what the
    all-gather does), handed back through bella_hip_set_B_device -> same B and same pairs as the one-shot assembly"""
    import torch
    from bella_amd import dist as bd
    g = load_golden("toylen80")
    eng.set_reads(g.rs)
    parts = []
    for r in range(2):
        lo, n = bd.block_range(r, 2, g.rs.nreads)
        sel = (g.tr >= lo) & (g.tr < lo + n)
        eng.assemble_panel(g.k, g.nkmers, lo, n, g.tk[sel], g.tr[sel], g.tp[sel])
        parts.append([t.clone() for t in eng.panel_tensors(0)])
    cnt = torch.cat([p[0] for p in parts]); ids = torch.cat([p[1] for p in parts]); val = torch.cat([p[2] for p in parts])
    colptr = torch.zeros(cnt.numel() + 1, dtype=torch.int32, device=cnt.device)
    colptr[1:] = torch.cumsum(cnt, 0).to(torch.int32)
    eng.set_B_device(g.k, g.nkmers, colptr, ids, val)
    exp = O.build_B(g.rs.nreads, g.tk, g.tr, g.tp)
    for a, b in zip(eng.get_B(), exp):
        assert np.array_equal(a, b)
    eng.overlap(BellaPars())
    pairs, ext, _ = eng.get_pairs()
    _, _, _, ep = oracle_pairs(g.rs, g.seqs, g.nkmers, g.tk, g.tr, g.tp, g.k)
    check_pairs(pairs, ext, ep, g.rs.lengths, g.k)


def test_ecsample_shaped_set_full_parity(eng):
    """BASELINE configs[0] shape: 15,152 reads with the interval lengths of the reference's E. coli sample (mean 8.2 kb,
    max 26.9 kb) on a random 4.64 Mb genome.  Whole SpGEMM output vs the oracle, X-drop on a sample."""
    z = np.load(os.path.join(GOLD, "ecsample_intervals.npz"))
    rs = synth.make_reads_from_intervals(z["start"], z["end"])
    assert rs.nreads == 15152 and int(rs.lengths.max()) < 65536
    t = synth.count_and_tuples(rs, 17, 2, 8, device="cuda:0")
    seqs = rs.seqs()
    eng.set_reads(rs)
    eng.assemble_tuples(17, t.nkmers, t.kmer, t.read, t.pos)
    (Bc, Br, Bv), flop, ecol, exp = oracle_pairs(rs, seqs, t.nkmers, t.kmer, t.read, t.pos)
    for a, b in zip(eng.get_B(), (Bc, Br, Bv)):
        assert np.array_equal(a, b)
    pars = BellaPars()
    n, flops = eng.overlap(pars)
    assert n == len(exp) and flops == int(flop.sum()) and n > 500000
    pairs, ext, colptrC = eng.get_pairs()
    assert np.array_equal(colptrC, ecol.astype(np.uint64))
    check_pairs(pairs, ext, exp, rs.lengths, 17)
    eng.align_pairs(pars)
    alns = eng.get_alignments()
    phi = O.slope(0.15)
    for n_ in np.random.default_rng(1).choice(len(pairs), size=400, replace=False):
        p, a = pairs[n_], alns[n_]
        rid, cid = int(p["rid"]), int(p["cid"])
        e = O.xavier_align(seqs[rid], seqs[cid], int(p["seedH"]), int(p["seedV"]), 7, 17)
        ok, ov = O.post_align(e["score"], e["begV"], e["endV"], e["begH"], e["endH"], len(seqs[rid]), len(seqs[cid]), phi)
        assert (int(a["score"]), int(a["begH"]), int(a["endH"]), int(a["begV"]), int(a["endV"]), int(a["ov"]), int(a["passed"]),
                int(a["steps"])) == (int(e["score"]), int(e["begH"]), int(e["endH"]), int(e["begV"]), int(e["endV"]), ov, int(ok),
                                     int(e["steps"]))


def test_baseline_config1_full_size_parity(eng):
    """BASELINE configs[1] at full size (10k reads x 10 kb, the bench workload): every pair record vs the oracle, plus
    size-independent properties (strict lower triangle, unique pairs, count==1 <=> single shared k-mer record)."""
    rs = synth.make_reads(10000, read_len=10000, coverage=30.0, err=0.15, seed=1)
    seqs = rs.seqs()
    eng.set_reads(rs)
    # the bench's own pipeline: the engine counts the k-mers, builds the dictionary and the tuples, and assembles from them; the
    # test kit's independent (torch) counter must agree with it tuple for tuple (ids = ranks of the canonical k-mers)
    nk, nt, _ = eng.count_kmers(17, 2, 8)
    t = synth.Tuples(*eng.get_tuples(), nk)
    t2 = synth.count_and_tuples(rs, 17, 2, 8, device="cuda:0")
    assert nk == t2.nkmers and np.array_equal(t.kmer, t2.kmer) and np.array_equal(t.read, t2.read) and np.array_equal(t.pos, t2.pos)
    del t2
    eng.assemble_counted()
    n, flops = eng.overlap(BellaPars(skipAlignment=True))
    pairs, ext, colptrC = eng.get_pairs()
    assert (pairs["rid"] > pairs["cid"]).all()
    key = pairs["cid"].astype(np.uint64) << np.uint64(32) | pairs["rid"].astype(np.uint64)
    assert len(np.unique(key)) == len(key)
    assert (np.diff(pairs["cid"].astype(np.int64)) >= 0).all()                       # column-major
    assert ((ext["nbins"] >= 1) & (ext["support"] >= 1)).all()
    _, flop, ecol, exp = oracle_pairs(rs, seqs, t.nkmers, t.kmer, t.read, t.pos)
    assert n == len(exp) == 755378 or n == len(exp)
    assert flops == int(flop.sum())
    check_pairs(pairs, ext, exp, rs.lengths, 17)
    # BASELINE configs[2]: X-drop on the same pairs; 20,000 of them (junk pairs included) against the oracle's scalar Xavier,
    # spread over the host cores
    eng.align_pairs(BellaPars())
    alns = eng.get_alignments()
    rng = np.random.default_rng(7)
    sample = np.sort(rng.choice(n, size=20000, replace=False))
    global _XSEQS
    _XSEQS = seqs
    import multiprocessing as mp
    jobs = [(int(pairs["rid"][i]), int(pairs["cid"][i]), int(pairs["seedH"][i]), int(pairs["seedV"][i])) for i in sample]
    with mp.get_context("fork").Pool(min(32, os.cpu_count() or 1)) as pool:
        res = pool.map(_xavier_job, jobs, chunksize=64)
    bad = 0
    for i, e in zip(sample, res):
        al = alns[i]
        got = (int(al["score"]), int(al["begH"]), int(al["endH"]), int(al["begV"]), int(al["endV"]))
        bad += got != e
    assert bad == 0, bad


def test_baseline_config3_100k_read_set_parity(eng):
    """BASELINE configs[3]'s read set (100k reads x 10 kb) on one GPU, the size the roofline target is quoted on and a different
    regime from 10k reads (~500 pairs per column, 92 % of them single-product, half-size key tables, four LDS classes):
      * every column's product count and pair count (colptrC) against the oracle's symbolic phase (all 100k columns, host cores),
      * every record of EVERY column (~50 M records) against the oracle's numeric phase on the host cores,
      * size-independent properties of all ~50 M records,
      * X-drop on ALL pairs, 200,000 of them (mostly chance pairs) against the oracle's scalar Xavier;
      * the engine's tuples against the test kit's independent counter."""
    import time
    t0 = time.time()
    rs = synth.make_reads_fast(100000, read_len=10000, coverage=30.0, err=0.15, seed=1)
    seqs = rs.seqs()
    eng.set_reads(rs)
    nk, nt, _ = eng.count_kmers(17, 2, 8)
    tk, tr, tp = eng.get_tuples()
    # the tuples the oracle is fed come from the engine's own counter: the test kit's independent (torch) counter must agree with
    # it tuple for tuple at this size too (ids = ranks of the canonical k-mers)
    t2 = synth.count_and_tuples(rs, 17, 2, 8, device="cuda:0")
    assert nk == t2.nkmers and np.array_equal(tk, t2.kmer) and np.array_equal(tr, t2.read) and np.array_equal(tp, t2.pos)
    del t2
    eng.assemble_counted()
    pars = BellaPars()
    n, flops = eng.overlap(pars)
    pairs, ext, colptrC = eng.get_pairs()
    assert n > 40_000_000
    assert (pairs["rid"] > pairs["cid"]).all()
    assert (np.diff(pairs["cid"].astype(np.int64)) >= 0).all()                       # column-major
    key = pairs["cid"].astype(np.uint64) << np.uint64(32) | pairs["rid"].astype(np.uint64)
    assert (np.diff(key.reshape(-1)) != 0).all() and len(np.unique(key)) == len(key)  # every pair once
    del key
    assert ((ext["nbins"] >= 1) & (ext["support"] >= 1)).all()
    t1 = time.time()
    Bc, Br, Bv = O.build_B(rs.nreads, tk, tr, tp)
    for a_, b_ in zip(eng.get_B(), (Bc, Br, Bv)):
        assert np.array_equal(a_, b_)
    per_row = np.diff(colptrC.astype(np.int64))
    sample = np.arange(rs.nreads, dtype=np.uint32)                                   # ALL columns (round 4 compared a quarter of them)
    flop, nnzc, per_col = O.spgemm_parallel(seqs, nk, Bc, Br, Bv, sample, 17, procs=min(192, os.cpu_count() or 1))
    t2 = time.time()
    assert flops == int(flop.astype(np.int64).sum())
    assert np.array_equal(per_row, nnzc.astype(np.int64))
    exp = np.concatenate([per_col[int(c)] for c in sample])                          # column-major, slot order inside: the engine's order
    del per_col
    check_pairs(pairs, ext, exp, rs.lengths, 17)
    # configs[3] aligns: X-drop on all pairs
    npass = eng.align_pairs(pars)
    alns = eng.get_alignments()
    assert 0 < npass < n
    rng = np.random.default_rng(11)
    pick = np.sort(rng.choice(n, size=200000, replace=False))                        # 0.4 % of the pairs, mostly chance pairs
    global _XSEQS
    _XSEQS = seqs
    import multiprocessing as mp
    jobs = list(zip(pairs["rid"][pick].tolist(), pairs["cid"][pick].tolist(), pairs["seedH"][pick].tolist(), pairs["seedV"][pick].tolist()))
    with mp.get_context("fork").Pool(min(128, os.cpu_count() or 1)) as pool:
        res = pool.map(_xavier_job, jobs, chunksize=256)
    bad = 0
    for i, e in zip(pick, res):
        al = alns[i]
        bad += (int(al["score"]), int(al["begH"]), int(al["endH"]), int(al["begV"]), int(al["endV"])) != e
    assert bad == 0, bad
    print("100k parity: %d pairs, %d sampled columns (%d records), gpu part %.0f s, oracle %.0f s, total %.0f s"
          % (n, len(sample), len(exp), t1 - t0, t2 - t1, time.time() - t0))


_XSEQS = None
_BIG_SETS = {}


def _xavier_job(job):
    rid, cid, sh, sv = job
    e = O.xavier_align(_XSEQS[rid], _XSEQS[cid], sh, sv, 7, 17)
    return (int(e["score"]), int(e["begH"]), int(e["endH"]), int(e["begV"]), int(e["endV"]))


# ---- k-mer counting, reliable dictionary, tuples on the device ---------------------------------------------

def _relabel_equal(a, b):
    if len(a) != len(b):
        return False
    if len(a) == 0:
        return True
    pa, pb = np.unique(a, return_inverse=True)[1], np.unique(b, return_inverse=True)[1]
    # same partition <=> the pairs (label_a, label_b) are as many as the labels on either side
    both = np.unique(np.stack([pa, pb], 1), axis=0)
    return len(both) == pa.max() + 1 == pb.max() + 1


def test_count_kmers_matches_oracle_and_reference_dump(eng, golden):
    g = golden
    eng.set_reads(g.rs)
    nk, nt, nd = eng.count_kmers(g.k, g.lower, g.upper, g.syncmer, g.window)
    codes, counts, tk, tr, tp, ndist = O.count_kmers(g.seqs, g.k, g.lower, g.upper, g.syncmer, g.window)
    assert (nk, nt, nd) == (len(codes), len(tk), ndist)
    dc, dn = eng.get_dictionary()
    assert np.array_equal(dc, codes) and np.array_equal(dn, counts)
    gk, gr, gp = eng.get_tuples()
    assert np.array_equal(gk, tk) and np.array_equal(gr, tr) and np.array_equal(gp, tp)
    # the reference's own dump: same reliable set, same tuples up to the k-mer numbering
    assert nk == g.nkmers and np.array_equal(gr, g.tr) and np.array_equal(gp, g.tp) and _relabel_equal(gk, g.tk)


@pytest.mark.parametrize("k,lower,upper", [(17, 2, 8), (11, 2, 4), (32, 2, 8), (3, 2, 65535), (5, 3, 60000), (21, 2, 2)])
def test_count_kmers_parameter_sweep(eng, k, lower, upper):
    rs = synth.make_reads(70, read_len=900, coverage=12.0, err=0.1, seed=11)
    eng.set_reads(rs)
    codes, counts, tk, tr, tp, ndist = O.count_kmers(rs.seqs(), k, lower, upper)
    # the sorted words carry their positions where word + position index fit 64 bits (all of these but k = 32); debug bit 14 takes the
    # hash-table look-up path that the other cases use; one pass over all words, then several passes over the bins
    for dbg, budget in ((0, 0), (16384, 0), (0, 9000), (16384, 9000)):
        set_mode(eng, dbg)
        eng.set_tuning("kcount_budget", budget)
        try:
            nk, nt, nd = eng.count_kmers(k, lower, upper)
        finally:
            set_mode(eng, 0)
            eng.set_tuning("kcount_budget")
        assert (nk, nt, nd) == (len(codes), len(tk), ndist)
        dc, dn = eng.get_dictionary()
        gk, gr, gp = eng.get_tuples()
        assert np.array_equal(dc, codes) and np.array_equal(dn, counts)
        assert np.array_equal(gk, tk) and np.array_equal(gr, tr) and np.array_equal(gp, tp)


@pytest.mark.parametrize("k,lower,upper", [(17, 2, 8), (6, 2, 60000), (11, 2, 30), (21, 2, 4), (32, 2, 8), (7, 3, 65535)])
def test_count_syncmers_parameter_sweep(eng, k, lower, upper):
    """the reference's -s mode (SyncmerCount + canonical-lookup tuple loop), low-error reads so that syncmers repeat"""
    rs = synth.make_reads(60, read_len=1200, coverage=10.0, err=0.01, seed=13, mix=(1 / 3, 1 / 3, 1 / 3))
    eng.set_reads(rs)
    codes, counts, tk, tr, tp, ndist = O.count_kmers(rs.seqs(), k, lower, upper, True)
    for budget in (0, 3000):                                         # one pass, then several passes over the bins
        eng.set_tuning("kcount_budget", budget)
        nk, nt, nd = eng.count_kmers(k, lower, upper, True)
        assert (nk, nt, nd) == (len(codes), len(tk), ndist)
        dc, dn = eng.get_dictionary()
        gk, gr, gp = eng.get_tuples()
        assert np.array_equal(dc, codes) and np.array_equal(dn, counts)
        assert np.array_equal(gk, tk) and np.array_equal(gr, tr) and np.array_equal(gp, tp)
    eng.set_tuning("kcount_budget")
    with pytest.raises(BellaHipError):
        eng.count_kmers(5, 2, 8, True)


@pytest.mark.parametrize("k,window,lower,upper", [(17, 10, 2, 8), (15, 1, 2, 30), (21, 50, 2, 8), (32, 7, 2, 8), (5, 3, 2, 65535),
                                                  (17, 3000, 2, 8)])
def test_count_minimizers_parameter_sweep(eng, k, window, lower, upper):
    """the reference's -w mode (getMinimizers' deque incl. its size_t range test, MinimizerCount, minimizer tuple branch);
    window 3000 exceeds every read: nothing is ever sampled"""
    rs = synth.make_reads(60, read_len=1200, coverage=10.0, err=0.02, seed=29, mix=(1 / 3, 1 / 3, 1 / 3))
    eng.set_reads(rs)
    codes, counts, tk, tr, tp, ndist = O.count_kmers(rs.seqs(), k, lower, upper, False, window)
    for budget, dbg in ((0, 0), (2000, 0), (0, 16384)):                # (debug bit 14: the look-up path instead of positions in the sort keys)
        eng.set_tuning("kcount_budget", budget)
        set_mode(eng, dbg)
        try:
            nk, nt, nd = eng.count_kmers(k, lower, upper, False, window)
        finally:
            set_mode(eng, 0)
        assert (nk, nt, nd) == (len(codes), len(tk), ndist)
        dc, dn = eng.get_dictionary()
        gk, gr, gp = eng.get_tuples()
        assert np.array_equal(dc, codes) and np.array_equal(dn, counts)
        assert np.array_equal(gk, tk) and np.array_equal(gr, tr) and np.array_equal(gp, tp)
    eng.set_tuning("kcount_budget")


def test_count_kmers_multi_pass_equals_single_pass(eng):
    rs = synth.make_reads(120, read_len=1500, coverage=15.0, err=0.12, seed=3)
    eng.set_reads(rs)
    eng.count_kmers(17, 2, 8)
    one = (eng.get_dictionary(), eng.get_tuples())
    eng.set_tuning("kcount_budget", 20000)                           # ~ten passes over the bins of the canonical word
    eng.count_kmers(17, 2, 8)
    many = (eng.get_dictionary(), eng.get_tuples())
    eng.set_tuning("kcount_budget")
    for a, b in zip(one[0] + one[1], many[0] + many[1]):
        assert np.array_equal(a, b)


def test_count_kmers_edge_cases(eng):
    # reads shorter than k contribute nothing; a u16 count wraps (kmercount.hpp:632-641): 65538 copies of one k-mer count as 2
    rs = synth.ReadSet.from_strings(["ACGT", "A" * 40000, "A" * 25570, "ACGTACGTTTGACCA"], ["a", "b", "c", "d"])
    eng.set_reads(rs)
    k = 17
    nk, nt, nd = eng.count_kmers(k, 2, 8)
    codes, counts, tk, tr, tp, ndist = O.count_kmers(rs.seqs(), k, 2, 8)
    assert (nk, nt, nd) == (len(codes), len(tk), ndist) == (1, 65538, 1) and counts[0] == 2
    gk, gr, gp = eng.get_tuples()
    assert np.array_equal(gk, tk) and np.array_equal(gr, tr) and np.array_equal(gp, tp)
    with pytest.raises(BellaHipError):
        eng.count_kmers(17, 1, 8)
    with pytest.raises(BellaHipError):
        eng.count_kmers(33, 2, 8)


def test_counted_pipeline_equals_tuple_pipeline(eng):
    """count -> assemble -> overlap entirely on the device == the same with host-side tuples (ids agree: ascending canonical order)"""
    rs = synth.make_reads(300, read_len=2500, coverage=20.0, err=0.15, seed=21)
    t = synth.count_and_tuples(rs, 17, 2, 8)
    eng.set_reads(rs)
    eng.assemble_tuples(17, t.nkmers, t.kmer, t.read, t.pos)
    n1, f1 = eng.overlap(BellaPars(skipAlignment=True))
    p1, e1, c1 = eng.get_pairs()
    nk, nt, _ = eng.count_kmers(17, 2, 8)
    assert nk == t.nkmers and nt == len(t.kmer)
    eng.assemble_counted()
    n2, f2 = eng.overlap(BellaPars(skipAlignment=True))
    p2, e2, c2 = eng.get_pairs()
    assert (n1, f1) == (n2, f2) and np.array_equal(p1, p2) and np.array_equal(e1, e2) and np.array_equal(c1, c2)
    assert eng.timings().kcount_ms > 0


# ---- FASTQ ingest through the library -----------------------------------------------------------------------

def test_load_fastq_equals_set_reads(eng, golden, tmp_path):
    import gzip
    g = golden
    p = tmp_path / "reads.fastq"
    p.write_bytes(gzip.open(os.path.join(GOLD, g.name, "reads.fastq.gz"), "rb").read())
    n, nb = eng.load_fastq(str(p))
    assert n == g.rs.nreads and nb == int(g.rs.offsets[-1])
    assert eng.names == g.names and np.array_equal(eng.lengths, g.rs.lengths)
    eng.count_kmers(g.k, g.lower, g.upper)
    a = eng.get_tuples()
    eng.set_reads(g.rs)
    eng.count_kmers(g.k, g.lower, g.upper)
    b = eng.get_tuples()
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_load_fastq_list_numbers_the_reads_through_the_files(eng, tmp_path):
    """the reference's -f list (kmercount.hpp:82-105): several files, read ids in list order; an empty file in the middle; chunks of
    the base stream that span file boundaries"""
    rs = synth.make_reads(260, read_len=3000, coverage=20.0, err=0.15, seed=9)
    cuts = [0, 1, 1, 120, 259, 260]                        # one read, an EMPTY file, 119 reads, 139 reads, one read
    paths = []
    for f in range(len(cuts) - 1):
        lo, hi = cuts[f], cuts[f + 1]
        part = synth.ReadSet(rs.codes[rs.offsets[lo]:rs.offsets[hi]], rs.offsets[lo:hi + 1] - rs.offsets[lo], rs.names[lo:hi])
        p = str(tmp_path / ("part%d.fastq" % f))
        synth.write_fastq(p, part)
        paths.append(p)
    n, nb = eng.load_fastq(paths)
    assert n == rs.nreads and nb == int(rs.offsets[-1])
    assert eng.names == rs.names and np.array_equal(eng.lengths, rs.lengths)
    st = eng.ingest_stats()
    assert int(st["file_bytes"]) == sum(os.path.getsize(p) for p in paths)
    eng.count_kmers(17, 2, 8)
    a = eng.get_tuples()
    eng.set_reads(rs)
    eng.count_kmers(17, 2, 8)
    for x, y in zip(a, eng.get_tuples()):
        assert np.array_equal(x, y)
    n0, _ = eng.load_fastq([])                             # an empty list is an empty read set, not an error
    assert n0 == 0


def test_load_fastq_streams_a_file_larger_than_the_pinned_chunk(eng, tmp_path):
    """configs[1]'s 10k reads as a 208 MB FASTQ: 104 M bases = two 64 MB chunks through the pinned buffers, the index on several
    threads; the packed reads on the device are those of set_reads (same tuples), names and lengths come back"""
    rs = synth.make_reads(10000, read_len=10000, coverage=30.0, err=0.15, seed=1)
    p = tmp_path / "big.fastq"
    synth.write_fastq(str(p), rs)
    n, nb = eng.load_fastq(str(p))
    st = eng.ingest_stats()
    assert (n, nb) == (rs.nreads, int(rs.offsets[-1])) and st["reads"] == n and st["bases"] == nb and st["file_bytes"] == p.stat().st_size
    assert nb > (64 << 20) and st["threads"] >= 1 and st["index_ms"] > 0 and st["upload_ms"] > 0
    assert eng.names == rs.names and np.array_equal(eng.lengths, rs.lengths)
    eng.count_kmers(17, 2, 8)
    a = eng.get_tuples()
    eng.set_reads(rs)
    eng.count_kmers(17, 2, 8)
    b = eng.get_tuples()
    assert len(a[0]) > 1000000
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_file_to_file_pipeline_reproduces_reference_up_to_kmer_labels(eng, tmp_path):
    """FASTQ file -> load_fastq -> count_kmers -> assemble_counted -> HashSpGEMM-shaped driver -> output file.  K-mer ids
    differ from the reference's (labels): the candidate pair set is identical, seeds -- and so a few borderline alignment
    decisions -- follow the fold order (as they do in the reference between thread counts); the quality evaluation agrees."""
    import gzip
    from bella_amd import evaluate as ev, hash_spgemm
    import io
    g = load_golden("toy120")
    p = tmp_path / "reads.fastq"
    p.write_bytes(gzip.open(os.path.join(GOLD, g.name, "reads.fastq.gz"), "rb").read())
    eng.load_fastq(str(p))
    eng.count_kmers(17, 2, 8)
    eng.assemble_counted()
    out = tmp_path / "o.out"
    key = lambda data: {tuple(ln.split(b"\t")[:2]) for ln in data.split(b"\n") if ln}
    hash_spgemm(eng, BellaPars(skipAlignment=True), str(out), stdout=io.StringIO())
    assert key(out.read_bytes()) == key(g.out["skip"])              # the candidate pairs do not depend on the labels
    hash_spgemm(eng, BellaPars(skipAlignment=False), str(out), stdout=io.StringIO())
    mine = out.read_bytes()
    pairs_mine, pairs_ref = key(mine), key(g.out["align"])
    assert len(pairs_mine ^ pairs_ref) <= 0.03 * len(pairs_ref)     # seeds (hence borderline pass/fail) follow the fold order
    G = ev.truth_pairs(ev.truth_from_names(g.names), 500)
    r1, r2 = ev.evaluate(ev.read_bella_output(mine, 500), G), ev.evaluate(ev.read_bella_output(g.out["align"], 500), G)
    assert abs(r1["recall"] - r2["recall"]) < 0.5 and abs(r1["precision"] - r2["precision"]) < 0.5


# ---- columns with >= 65,536 products (wide.hpp) -----------------------------------------------------------------------------

@pytest.mark.parametrize("budget,layout,passdbg,rowlists", [(0, 0, 0, 1), (300000, 0, 0, 1), (0, 1024, 0, 0), (300000, 2048, 0, 0), (0, 0, 0, 0), (0, 0, 4096, 1), (300000, 0, 4096, 1),
                                                             (0, 2048, 0, 1), (0, 0, 4096, 0)])
def test_wide_columns_bit_exact(eng, budget, layout, passdbg, rowlists):
    """70 near-identical reads with -u 80: every k-mer is shared by all of them, column 0 has ~200k products (the row kernels
    index products with 16 bits); strands mixed, two reads carry a 700-base deletion plus a random tail (a second overlap bin)"""
    eng.set_tuning("wide_budget", budget)                            # 300000: several batches of wide columns
    set_mode(eng, layout)                                            # 1024: A' in order of first appearance; 2048: the row lists asked for "do not fit"
    eng.set_tuning("row_lists", rowlists)                            # 1: grouping of the wide columns in LDS from the row lists; 0 (default): a long-list input like this one
                                                                     # gets them anyway when they fit (2048: "no room", 1024: other layout -> a list expanded per batch)
    rng = np.random.default_rng(17)
    base = rng.integers(0, 4, size=3000, dtype=np.uint8)
    comp = (3 - base)[::-1]
    seqs = []
    for i in range(70):
        s = base if i % 3 else comp
        if i in (11, 40):
            s = np.concatenate([s[:900], s[1600:], rng.integers(0, 4, size=700, dtype=np.uint8)])   # two diagonals, estimates 700 apart
        seqs.append(synth.BASES[s].tobytes())
    rs = synth.ReadSet.from_strings(seqs)
    eng.set_reads(rs)
    nk, nt, _ = eng.count_kmers(17, 2, 80)
    tk, tr, tp = eng.get_tuples()
    eng.assemble_counted()
    set_mode(eng, passdbg)                                           # 4096: the sort-based grouping of the wide columns also with row lists
    n, flops = eng.overlap(BellaPars(skipAlignment=True))
    set_mode(eng, 0)
    eng.set_tuning("row_lists")
    pairs, ext, colptrC = eng.get_pairs()
    _, flop, ecol, exp = oracle_pairs(rs, rs.seqs(), nk, tk, tr, tp, 17)
    assert int(flop.max()) >= 65536 and flops == int(flop.sum()) and n == len(exp)
    assert np.array_equal(colptrC, ecol.astype(np.uint64))
    check_pairs(pairs, ext, exp, rs.lengths, 17)
    assert (ext["nbins"] > 1).any()


def test_wide_columns_with_more_partners_than_the_grouping_table_holds(eng):
    """1,700 copies of one 26-base read (strands mixed), -u 2000: ten k-mers shared by all reads, column i has 10 * (1699 - i) products
    on 1699 - i partners -- ~600 columns above the LDS tiers, the first ~160 of them with more than the 1,536 partners the LDS grouping
    table of the wide path holds: the first grouping pass raises its flag, the append pass (already enqueued behind it) leaves by
    itself and the batch takes the sort-based path"""
    rng = np.random.default_rng(23)
    base = rng.integers(0, 4, size=26, dtype=np.uint8)
    comp = (3 - base)[::-1]
    seqs = [synth.BASES[base if i % 3 else comp].tobytes() for i in range(1700)]
    rs = synth.ReadSet.from_strings(seqs)
    eng.set_tuning("row_lists", 1)
    eng.set_reads(rs)
    nk, nt, _ = eng.count_kmers(17, 2, 2000)
    tk, tr, tp = eng.get_tuples()
    eng.assemble_counted()
    n, flops = eng.overlap(BellaPars(skipAlignment=True))
    eng.set_tuning("row_lists")
    pairs, ext, colptrC = eng.get_pairs()
    _, flop, ecol, exp = oracle_pairs(rs, rs.seqs(), nk, tk, tr, tp, 17)
    assert int(flop.max()) >= 11009 and int(np.diff(ecol.astype(np.int64)).max()) > 1536
    assert flops == int(flop.sum()) and n == len(exp)
    assert np.array_equal(colptrC, ecol.astype(np.uint64))
    check_pairs(pairs, ext, exp, rs.lengths, 17)


def test_hifi_syncmer_medium_set_parity(eng):
    """BASELINE configs[4]'s regime at medium size: 1,200 HiFi-like reads (15 kb, 0.5 % error, 30x), syncmer selection (-s) with
    -u 40: ~30,000 products per column (the columns of the 11k-65k global-workspace path and of the >= 65,536 wide path), pairs
    with hundreds to thousands of products (long fold walks).  Every column's product and pair count against the oracle's symbolic
    phase, every record of 150 sampled columns against its numeric phase, X-drop on 3,000 sampled pairs."""
    rs = synth.make_reads(1200, read_len=15000, coverage=30.0, err=0.005, seed=2, mix=(1 / 3, 1 / 3, 1 / 3))
    seqs = rs.seqs()
    eng.set_reads(rs)
    nk, nt, _ = eng.count_kmers(17, 2, 40, syncmer=True)
    tk, tr, tp = eng.get_tuples()
    eng.assemble_counted()
    pars = BellaPars(errorRate=0.005)
    n, flops = eng.overlap(pars)
    pairs, ext, colptrC = eng.get_pairs()
    Bc, Br, Bv = O.build_B(rs.nreads, tk, tr, tp)
    per_row = np.diff(colptrC.astype(np.int64))
    sample = np.unique(np.concatenate([np.arange(0, rs.nreads, 9), np.argsort(per_row)[-20:]])).astype(np.uint32)
    flop, nnzc, per_col = O.spgemm_parallel(seqs, nk, Bc, Br, Bv, sample, 17)
    assert flops == int(flop.astype(np.int64).sum()) and int(flop.max()) > 11008          # beyond the LDS tiers
    assert flops / max(n, 1) > 200                                                         # long product lists
    assert np.array_equal(per_row, nnzc.astype(np.int64))
    got_idx = np.concatenate([np.arange(int(colptrC[c]), int(colptrC[c + 1])) for c in sample])
    exp = np.concatenate([per_col[int(c)] for c in sample])
    check_pairs(pairs[got_idx], ext[got_idx], exp, rs.lengths, 17)
    eng.align_pairs(pars)
    alns = eng.get_alignments()
    pick = np.sort(np.random.default_rng(5).choice(n, size=min(3000, n), replace=False))
    global _XSEQS
    _XSEQS = seqs
    import multiprocessing as mp
    jobs = [(int(pairs["rid"][i]), int(pairs["cid"][i]), int(pairs["seedH"][i]), int(pairs["seedV"][i])) for i in pick]
    with mp.get_context("fork").Pool(min(64, os.cpu_count() or 1)) as pool:
        res = pool.map(_xavier_job, jobs, chunksize=16)
    bad = sum((int(alns[i]["score"]), int(alns[i]["begH"]), int(alns[i]["endH"]), int(alns[i]["begV"]), int(alns[i]["endV"])) != e
              for i, e in zip(pick, res))
    assert bad == 0, bad


@pytest.mark.parametrize("upper", [8, 40])
def test_baseline_config4_hifi_10k_reads_parity(eng, upper):
    """BASELINE configs[4] at single-GPU scale: 10,000 HiFi reads (15 kb, 0.5 % error, 30x), syncmer selection (-s), with the
    reference's default bounds (-u 8: almost every true syncmer exceeds the bound) and with -u 40 (hundreds of
    products per pair, columns above the LDS tiers).  Every column's product and pair count against the oracle's symbolic phase,
    every record of more than a tenth of the columns (every ninth and the 100 pair-richest) against its numeric phase,
    size-independent properties of all records."""
    rs = synth.make_reads(10000, read_len=15000, coverage=30.0, err=0.005, seed=2, mix=(1 / 3, 1 / 3, 1 / 3))
    seqs = rs.seqs()
    eng.set_reads(rs)
    nk, nt, _ = eng.count_kmers(17, 2, upper, syncmer=True)
    tk, tr, tp = eng.get_tuples()
    eng.assemble_counted()
    pars = BellaPars(errorRate=0.005)                                                # the full pipeline: SpGEMM + X-drop + the pass test
    n, flops = eng.overlap(pars)
    pairs, ext, colptrC = eng.get_pairs()
    assert n > 1000
    assert (pairs["rid"] > pairs["cid"]).all() and (np.diff(pairs["cid"].astype(np.int64)) >= 0).all()
    key = pairs["cid"].astype(np.uint64) << np.uint64(32) | pairs["rid"].astype(np.uint64)
    assert len(np.unique(key)) == len(key)
    Bc, Br, Bv = O.build_B(rs.nreads, tk, tr, tp)
    for a_, b_ in zip(eng.get_B(), (Bc, Br, Bv)):
        assert np.array_equal(a_, b_)
    per_row = np.diff(colptrC.astype(np.int64))
    sample = np.unique(np.concatenate([np.arange(0, rs.nreads, 9), np.argsort(per_row)[-100:]])).astype(np.uint32)
    assert len(sample) >= rs.nreads // 10
    flop, nnzc, per_col = O.spgemm_parallel(seqs, nk, Bc, Br, Bv, sample, 17)
    assert flops == int(flop.astype(np.int64).sum())
    assert np.array_equal(per_row, nnzc.astype(np.int64))
    got_idx = np.concatenate([np.arange(int(colptrC[c]), int(colptrC[c + 1])) for c in sample])
    exp = np.concatenate([per_col[int(c)] for c in sample])
    check_pairs(pairs[got_idx], ext[got_idx], exp, rs.lengths, 17)
    # configs[4] aligns: X-drop on all pairs (HiFi: true overlaps of up to 15 kb per direction), 6,000 of them against the oracle's scalar
    # Xavier and PostAlignDecision (errorRate 0.005)
    npass = eng.align_pairs(pars)
    alns = eng.get_alignments()
    assert 0 < npass <= n
    pick = np.sort(np.random.default_rng(13).choice(n, size=min(6000, n), replace=False))
    global _XSEQS
    _XSEQS = seqs
    import multiprocessing as mp
    jobs = list(zip(pairs["rid"][pick].tolist(), pairs["cid"][pick].tolist(), pairs["seedH"][pick].tolist(), pairs["seedV"][pick].tolist()))
    with mp.get_context("fork").Pool(min(128, os.cpu_count() or 1)) as pool:
        res = pool.map(_xavier_job, jobs, chunksize=16)
    phi = O.slope(0.005)
    bad = 0
    for i, e in zip(pick, res):
        al = alns[i]
        bad += (int(al["score"]), int(al["begH"]), int(al["endH"]), int(al["begV"]), int(al["endV"])) != e
        ok, ov = O.post_align(e[0], e[3], e[4], e[1], e[2], int(rs.lengths[pairs["rid"][i]]), int(rs.lengths[pairs["cid"][i]]), phi)
        bad += (int(al["passed"]), int(al["ov"])) != (int(ok), int(ov))
    assert bad == 0, bad
    print("HiFi 10k reads -u %d: %d reliable syncmers, %d tuples, %d products, %d pairs (%d pass), %d records and %d alignments compared"
          % (upper, nk, nt, flops, n, npass, len(exp), len(pick)))


@pytest.mark.parametrize("upper,stages", [(8, 1), (40, 4)])
def test_baseline_config4_hifi_100k_reads(eng, upper, stages):
    """BASELINE configs[4]'s regime at 100,000 HiFi reads (15 kb, 0.5 % error, 30x; a tenth of the configuration's 1M reads, one GPU),
    syncmer selection, -u 8 and -u 40.  With -u 40 that is ~3e9 products: the expanding default layout and the path of the columns
    above the LDS tiers at a size where they are the product, formed in four column stages as a -m budget would (the symbolic phase
    alone gives the boundaries).  Every column's product and pair count against the symbolic phase of the engine AND of the oracle,
    every record of > 2 % of the columns against the oracle's numeric phase, size-independent properties of all records, X-drop on a
    stage's pairs with 3,000 of them against the oracle."""
    import time
    t0 = time.time()
    if "hifi100k" not in _BIG_SETS:                                                  # (both parameter sets run on the same reads)
        _BIG_SETS.clear()
        r_ = synth.make_reads(100000, read_len=15000, coverage=30.0, err=0.005, seed=2, mix=(1 / 3, 1 / 3, 1 / 3))
        _BIG_SETS["hifi100k"] = (r_, r_.seqs())
    rs, seqs = _BIG_SETS["hifi100k"]
    t1 = time.time()
    eng.set_reads(rs)
    nk, nt, _ = eng.count_kmers(17, 2, upper, syncmer=True)
    tk, tr, tp = eng.get_tuples()
    eng.assemble_counted()
    pars = BellaPars(errorRate=0.005)
    colS, nS, fS = eng.count_pairs(pars)                                             # estimateFLOP + estimateNNZ_Hash + prefixsum
    bounds = [0] + [int(np.searchsorted(colS, np.uint64(b * int(nS) // stages), side="right")) - 1 for b in range(1, stages)] + [rs.nreads]
    parts, exts, alns_last, n, flops = [], [], None, 0, 0
    try:
        for b in range(stages):
            eng.set_column_range(bounds[b], bounds[b + 1] - bounds[b])
            nb, fb = eng.overlap(pars)
            p, e, cp = eng.get_pairs()
            assert nb == int(colS[bounds[b + 1]] - colS[bounds[b]])
            parts.append(p); exts.append(e); n += nb; flops += fb
            if b == stages - 1:                                                          # the alignment stage on the last stage's pairs
                npass = eng.align_pairs(pars)
                alns_last = eng.get_alignments()
                assert 0 < npass <= nb
    finally:
        eng.set_column_range(0, 0xFFFFFFFF)
    t2 = time.time()
    pairs, ext = np.concatenate(parts), np.concatenate(exts)
    assert n == nS == len(pairs) and flops == fS and n > 10000
    assert (pairs["rid"] > pairs["cid"]).all() and (np.diff(pairs["cid"].astype(np.int64)) >= 0).all()
    key = pairs["cid"].astype(np.uint64) << np.uint64(32) | pairs["rid"].astype(np.uint64)
    assert (np.diff(key) != 0).all() and len(np.unique(key)) == len(key)
    del key
    assert np.array_equal(np.bincount(pairs["cid"], minlength=rs.nreads), np.diff(colS.astype(np.int64)))
    assert ((ext["nbins"] >= 1) & (ext["support"] >= 1)).all()
    Bc, Br, Bv = O.build_B(rs.nreads, tk, tr, tp)
    for a_, b_ in zip(eng.get_B(), (Bc, Br, Bv)):
        assert np.array_equal(a_, b_)
    per_row = np.diff(colS.astype(np.int64))
    sample = np.unique(np.concatenate([np.arange(0, rs.nreads, 45), np.argsort(per_row)[-100:]])).astype(np.uint32)
    assert len(sample) >= rs.nreads // 50
    flop, nnzc, per_col = O.spgemm_parallel(seqs, nk, Bc, Br, Bv, sample, 17)
    t3 = time.time()
    assert flops == int(flop.astype(np.int64).sum())
    assert np.array_equal(per_row, nnzc.astype(np.int64))
    got_idx = np.concatenate([np.arange(int(colS[c]), int(colS[c + 1])) for c in sample])
    exp = np.concatenate([per_col[int(c)] for c in sample])
    check_pairs(pairs[got_idx], ext[got_idx], exp, rs.lengths, 17)
    last = parts[-1]
    pick = np.sort(np.random.default_rng(3).choice(len(last), size=min(3000, len(last)), replace=False))
    global _XSEQS
    _XSEQS = seqs
    import multiprocessing as mp
    jobs = list(zip(last["rid"][pick].tolist(), last["cid"][pick].tolist(), last["seedH"][pick].tolist(), last["seedV"][pick].tolist()))
    with mp.get_context("fork").Pool(min(128, os.cpu_count() or 1)) as pool:
        res = pool.map(_xavier_job, jobs, chunksize=16)
    bad = sum((int(alns_last[i]["score"]), int(alns_last[i]["begH"]), int(alns_last[i]["endH"]), int(alns_last[i]["begV"]), int(alns_last[i]["endV"])) != e
              for i, e in zip(pick, res))
    assert bad == 0, bad
    print("HiFi 100k reads -u %d: %d reliable syncmers, %d tuples, %d products, %d pairs in %d stages, %d records compared; reads %.0f s, engine %.0f s, oracle %.0f s, total %.0f s"
          % (upper, nk, nt, flops, n, stages, len(exp), t1 - t0, t2 - t1, t3 - t2, time.time() - t0))


class _PackedSeqs:
    """read r's bases out of one ASCII buffer (what a million-read set keeps instead of a million bytes objects)"""
    def __init__(self, asc, offsets):
        self.asc, self.offsets = asc, offsets

    def __getitem__(self, r):
        return self.asc[int(self.offsets[r]):int(self.offsets[r + 1])].tobytes()


def test_baseline_config4_hifi_1M_reads(eng):
    """BASELINE configs[4] at ITS OWN size on one GPU: 1,000,000 synthetic HiFi reads (15 kb, 0.5 % error, 30x: 15 G bases), syncmer
    selection, the reference's default bound -u 8 (-u 40 stays at 100k reads: ~3e10 products are a many-stage run).
    Reads from the device generator (bella_testkit.synth.make_reads_torch), then the whole device pipeline -- count, assemble, the
    symbolic phase alone, the numeric phase, X-drop on every candidate pair -- and: size-independent properties of ALL records; the
    pair and product count of EVERY column against the engine's own symbolic phase and against the oracle's; every record of > 1 % of
    the columns against the oracle's numeric phase; > 5,000 alignments against the oracle.  BELLA_TEST_HIFI_READS overrides the size."""
    import time
    import multiprocessing as mp
    nreads = int(os.environ.get("BELLA_TEST_HIFI_READS", "1000000"))
    t0 = time.time()
    _BIG_SETS.clear()
    rs = synth.make_reads_fast(nreads, read_len=15000, coverage=30.0, err=0.005, seed=4, mix=(1 / 3, 1 / 3, 1 / 3))
    asc = np.ascontiguousarray(api._ACGT[rs.codes])
    rs.codes = None                                             # (15 GB: the ASCII buffer is what everything below reads)
    offs = np.ascontiguousarray(rs.offsets, dtype=np.uint64)
    t1 = time.time()
    eng.set_reads_raw(asc, offs, names=rs.names)
    nk, nt, _ = eng.count_kmers(17, 2, 8, syncmer=True)
    eng.assemble_counted()
    pars = BellaPars(errorRate=0.005)
    colS, nS, fS = eng.count_pairs(pars)                        # estimateFLOP + estimateNNZ_Hash + prefixsum on all columns
    n, flops = eng.overlap(pars)
    pairs, ext, colptrC = eng.get_pairs()
    npass = eng.align_pairs(pars)
    alns = eng.get_alignments()
    t2 = time.time()
    assert n == nS == len(pairs) and flops == fS and n > nreads // 20 and 0 < npass <= n
    assert np.array_equal(colptrC, colS)
    assert (pairs["rid"] > pairs["cid"]).all() and (np.diff(pairs["cid"].astype(np.int64)) >= 0).all()
    key = pairs["cid"].astype(np.uint64) << np.uint64(32) | pairs["rid"].astype(np.uint64)
    assert len(np.unique(key)) == len(key)
    del key
    per_row = np.diff(colS.astype(np.int64))
    assert np.array_equal(np.bincount(pairs["cid"], minlength=nreads), per_row)
    assert ((ext["nbins"] >= 1) & (ext["support"] >= 1)).all()
    lens = np.diff(rs.offsets)
    assert (pairs["seedH"].astype(np.int64) + 17 <= lens[pairs["rid"]]).all() and (pairs["seedV"].astype(np.int64) + 17 <= lens[pairs["cid"]]).all()
    passed = alns["passed"] != 0
    assert int(passed.sum()) == npass
    # the oracle: B from the engine's tuples (host, one thread), the symbolic phase of all columns and the numeric phase of a sample
    tk, tr, tp = eng.get_tuples()
    Bc, Br, Bv = O.build_B(nreads, tk, tr, tp)
    del tk, tr, tp
    for a_, b_ in zip(eng.get_B(), (Bc, Br, Bv)):
        assert np.array_equal(a_, b_)
    sample = np.unique(np.concatenate([np.arange(0, nreads, 90), np.argsort(per_row)[-200:]])).astype(np.uint32)
    assert len(sample) >= nreads // 100
    flop, nnzc, per_col = O.spgemm_parallel((asc, rs.offsets), nk, Bc, Br, Bv, sample, 17, procs=min(128, os.cpu_count() or 1))
    t3 = time.time()
    assert flops == int(flop.astype(np.int64).sum()) and np.array_equal(per_row, nnzc.astype(np.int64))
    got_idx = np.concatenate([np.arange(int(colS[c]), int(colS[c + 1])) for c in sample])
    exp = np.concatenate([per_col[int(c)] for c in sample])
    check_pairs(pairs[got_idx], ext[got_idx], exp, rs.lengths, 17)
    pick = np.sort(np.random.default_rng(5).choice(len(pairs), size=min(6000, len(pairs)), replace=False))
    global _XSEQS
    _XSEQS = _PackedSeqs(asc, rs.offsets)
    jobs = list(zip(pairs["rid"][pick].tolist(), pairs["cid"][pick].tolist(), pairs["seedH"][pick].tolist(), pairs["seedV"][pick].tolist()))
    with mp.get_context("fork").Pool(min(128, os.cpu_count() or 1)) as pool:
        res = pool.map(_xavier_job, jobs, chunksize=16)
    bad = sum((int(alns[i]["score"]), int(alns[i]["begH"]), int(alns[i]["endH"]), int(alns[i]["begV"]), int(alns[i]["endV"])) != e for i, e in zip(pick, res))
    assert bad == 0, bad
    _XSEQS = None
    print("HiFi %d reads -u 8: %d reliable syncmers, %d tuples, %d products, %d pairs, %d passed; %d columns / %d records / %d alignments against the oracle; reads %.0f s, engine %.0f s, oracle %.0f s, total %.0f s"
          % (nreads, nk, nt, flops, n, npass, len(sample), len(exp), len(pick), t1 - t0, t2 - t1, t3 - t2, time.time() - t0))


@pytest.mark.parametrize("name,dbg", [("toy120", 32), ("toyrep90", 32), ("toyhifi50", 32), ("toysync60", 32), ("toyrep90", 32 | 64),
                                      ("toy120", 32 | 65536), ("toyrep90", 32 | 65536), ("toy120", 32 | 4096 | 65536)])
def test_columns_above_the_lds_tiers_on_the_sort_based_path(name, dbg):
    """the path of the columns with more products than the largest LDS tier when a pass has many of them (wide.hpp: expand, radix
    sort, slot order, one workgroup per pair with the closed-form fold): with a 64-product LDS tier and debug bit 5 most
    columns of a golden set take it -- single- and multi-bin pairs, long lists, both strands; then the many-bins set (serial
    fold for > 16 bins)"""
    g = load_golden(name)
    e = Engine(0)
    try:
        e.set_tuning("lds_tiers", 64)                                # (per context: no process-wide state)
        set_mode(e, dbg)                                             # bit 6: 64-bit sort keys (the default here is 32-bit); bit 16: one-partner B' entries carry
                                                                     # the partner (k_wide_expand / the batch's own product lists read them); bit 12: radix-sort grouping
        e.set_reads(g.rs)
        e.assemble_tuples(g.k, g.nkmers, g.tk, g.tr, g.tp)
        n, flops = e.overlap(BellaPars(skipAlignment=True, kmerSize=g.k))
        pairs, ext, colptrC = e.get_pairs()
        _, flop, ecol, exp = oracle_pairs(g.rs, g.seqs, g.nkmers, g.tk, g.tr, g.tp, g.k)
        assert flops == int(flop.sum()) and n == len(exp) and int((flop > 64).sum()) > 10
        assert np.array_equal(colptrC, ecol.astype(np.uint64))
        check_pairs(pairs, ext, exp, g.rs.lengths, g.k)
        if name == "toyrep90":
            assert (ext["nbins"] > 1).any()
            a = [(t, 600 * t + 200) for t in range(40)]
            b = sorted([(t, 100 + 19 * t) for t in range(40)], key=lambda x: x[1])
            c = [(100 + t, 50 + 40 * t) for t in range(120)]
            d = [(100 + t, 60 + 40 * t) for t in range(120)]
            exp2 = _run_constructed(e, [30000, 30000, 6000, 6000], [a, b, c, d], 300)
            by = {(int(p["rid"]), int(p["cid"])): p for p in exp2}
            assert by[(1, 0)]["nbins"] > 16 and by[(3, 2)]["support"] > 64
    finally:
        e.close()


@pytest.mark.parametrize("tier", ["8192", "11008"])
def test_big_lds_tiers_bit_exact(tier):
    """every column through the 8192-product LDS tier (the 16-positions-per-thread instance of the row kernel)"""
    g = load_golden("toyrep90")
    e = Engine(0)
    try:
        e.set_tuning("lds_tiers", int(tier))
        e.set_reads(g.rs)
        e.assemble_tuples(g.k, g.nkmers, g.tk, g.tr, g.tp)
        n, flops = e.overlap(BellaPars(skipAlignment=True))
        pairs, ext, colptrC = e.get_pairs()
        _, flop, ecol, exp = oracle_pairs(g.rs, g.seqs, g.nkmers, g.tk, g.tr, g.tp, g.k)
        assert flops == int(flop.sum()) and n == len(exp)
        check_pairs(pairs, ext, exp, g.rs.lengths, g.k)
    finally:
        e.close()


def test_counted_panels_equal_one_shot_assembly(eng):
    """multi-GPU path without a host copy of the tuples: count on the device, assemble the rows of each read block from the
    device-resident tuples, concatenate (= the all-gather), set_B_device -> the same B and pairs as assemble_counted"""
    import torch
    from bella_amd import dist as bd
    rs = synth.make_reads(150, read_len=2000, coverage=15.0, err=0.15, seed=41)
    eng.set_reads(rs)
    nk, nt, _ = eng.count_kmers(17, 2, 8)
    eng.assemble_counted()
    B1 = eng.get_B()
    eng.overlap(BellaPars(skipAlignment=True))
    p1 = eng.get_pairs()
    parts = []
    for r in range(3):
        lo, n = bd.block_range(r, 3, rs.nreads)
        eng.assemble_counted_panel(lo, n)
        parts.append([t.clone() for t in eng.panel_tensors(0)])
    cnt = torch.cat([p[0] for p in parts]); ids = torch.cat([p[1] for p in parts]); val = torch.cat([p[2] for p in parts])
    colptr = torch.zeros(cnt.numel() + 1, dtype=torch.int32, device=cnt.device)
    colptr[1:] = torch.cumsum(cnt, 0).to(torch.int32)
    eng.set_B_device(17, nk, colptr, ids, val)
    for a, b in zip(eng.get_B(), B1):
        assert np.array_equal(a, b)
    eng.overlap(BellaPars(skipAlignment=True))
    p2 = eng.get_pairs()
    for a, b in zip(p1, p2):
        assert np.array_equal(a, b)


def test_library_communicator_allgather_single_rank(eng):
    """the C-ABI multi-GPU entry points on the one GPU there is: a 1-rank RCCL communicator, the whole read set as this rank's
    panel, bella_hip_allgather_panels -> the same B and pairs as the one-shot assembly; and the argument checks"""
    rs = synth.make_reads(150, read_len=2000, coverage=15.0, err=0.15, seed=43)
    eng.set_reads(rs)
    nk, nt, _ = eng.count_kmers(17, 2, 8)
    eng.assemble_counted()
    B1 = eng.get_B()
    eng.overlap(BellaPars(skipAlignment=True))
    p1 = eng.get_pairs()
    with pytest.raises(BellaHipError):
        eng.allgather_panels()                     # no communicator yet
    cid = eng.comm_id()
    assert len(cid) == 128 and any(cid)
    eng.comm_init(1, 0, cid)
    with pytest.raises(BellaHipError):
        eng.allgather_panels()                     # no panel
    eng.assemble_counted_panel(0, rs.nreads - 10)
    with pytest.raises(BellaHipError) as e:
        eng.allgather_panels()                     # the panels do not cover the read set
    assert e.value.code == -3
    eng.assemble_counted_panel(0, rs.nreads)
    eng.allgather_panels()
    for a, b in zip(eng.get_B(), B1):
        assert np.array_equal(a, b)
    eng.overlap(BellaPars(skipAlignment=True))
    for a, b in zip(p1, eng.get_pairs()):
        assert np.array_equal(a, b)
    # distributed counting on the 1-rank communicator: the whole code space here, tuples for a read block only
    nk2, nt2, nd2 = eng.count_kmers_dist(40, 70, 17, 2, 8)
    assert nk2 == nk
    tk, tr, tp = eng.get_tuples()
    eng.count_kmers(17, 2, 8)
    fk, fr, fp = eng.get_tuples()
    keep = (fr >= 40) & (fr < 110)
    assert nt2 == int(keep.sum()) and np.array_equal(tk, fk[keep]) and np.array_equal(tr, fr[keep]) and np.array_equal(tp, fp[keep])
    eng.count_kmers_dist(40, 70, 17, 2, 8)
    with pytest.raises(BellaHipError):
        eng.assemble_counted()                     # the tuples cover a block only
    with pytest.raises(BellaHipError):
        eng.assemble_counted_panel(30, 20)         # outside the block
    eng.assemble_counted_panel(40, 70)
    eng.comm_destroy()


def test_half_size_key_tables_layout_bit_exact(eng, monkeypatch):
    """the LDS layout of pair-rich inputs (key tables of cap/2 slots, Gaux inside T2's upper half) on the multi-bin golden set"""
    set_mode(eng, 16)
    for name in ("toyrep90", "toy120"):
        g = load_golden(name)
        eng.set_reads(g.rs)
        eng.assemble_tuples(g.k, g.nkmers, g.tk, g.tr, g.tp)
        n, flops = eng.overlap(BellaPars(skipAlignment=True))
        pairs, ext, colptrC = eng.get_pairs()
        _, flop, ecol, exp = oracle_pairs(g.rs, g.seqs, g.nkmers, g.tk, g.tr, g.tp, g.k)
        assert n == len(exp)
        check_pairs(pairs, ext, exp, g.rs.lengths, g.k)


def test_out_of_order_lists_fall_back_to_the_repairing_path(eng):
    """the LDS tiers only CHECK that the ordered scatter produced product order (it always does on gfx950) and hand a column
    to the global path, which repairs, otherwise: inject the failure for every fifth column -> same results"""
    g = load_golden("toyrep90")
    set_mode(eng, 4)
    try:
        eng.set_reads(g.rs)
        eng.assemble_tuples(g.k, g.nkmers, g.tk, g.tr, g.tp)
        n, flops = eng.overlap(BellaPars(skipAlignment=True))
        pairs, ext, colptrC = eng.get_pairs()
        _, flop, ecol, exp = oracle_pairs(g.rs, g.seqs, g.nkmers, g.tk, g.tr, g.tp, g.k)
        assert n == len(exp)
        check_pairs(pairs, ext, exp, g.rs.lengths, g.k)
        assert eng.timings().retry_columns >= 5                          # the redone columns are counted and reported
        set_mode(eng, 0)
        eng.overlap(BellaPars(skipAlignment=True))
        assert eng.timings().retry_columns == 0                         # gfx950 keeps product order: nothing is redone
    finally:
        set_mode(eng, 0)


def test_staged_output_equals_single_stage(eng, golden, tmp_path):
    """the reference forms its output in stages of consecutive columns under a memory budget (overlap.hpp:682-789): three
    stages through bella_hip_set_column_range write the same files"""
    import io
    g = golden
    eng.set_reads(g.rs)
    eng.assemble_tuples(g.k, g.nkmers, g.tk, g.tr, g.tp)
    f = str(tmp_path / "o.out")
    so = io.StringIO()
    api.hash_spgemm(eng, BellaPars(skipAlignment=True, errorRate=g.err, kmerSize=g.k), f, stdout=so, stages=3)
    assert open(f, "rb").read() == g.out["skip"] and so.getvalue().split()[0] == g.stdout["skip"][2]
    one = str(tmp_path / "one.out")
    api.hash_spgemm(eng, BellaPars(errorRate=g.err, kmerSize=g.k), one, stdout=io.StringIO())
    so = io.StringIO()
    api.hash_spgemm(eng, BellaPars(errorRate=g.err, kmerSize=g.k), f, stdout=so, stages=3)
    assert open(f, "rb").read() == open(one, "rb").read()
    assert int(so.getvalue().split()[0]) == int(g.stdout["align"][2])


def _run_ranks(nranks, body, timeout=300):
    """one host thread per context (ctypes releases the GIL): the collective entry points rendezvous inside the library"""
    import threading
    out, err = [None] * nranks, []

    def work(r):
        try:
            out[r] = body(r)
        except Exception as e:  # noqa: BLE001
            err.append((r, e))
    th = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(nranks)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout)
    assert not any(t.is_alive() for t in th), "a rank is still inside a collective call: the ranks did not leave it together"
    return out, err


def test_in_process_transport_waits_have_a_deadline(monkeypatch):
    """a rank that never joins (init) or never enters the exchange: its peers leave with an error after BELLA_HIP_COMM_TIMEOUT_S instead
    of waiting forever (comm.hpp: every rendezvous wait of the in-process transport has the deadline comm_sync gives the stream waits)"""
    import time
    monkeypatch.setenv("BELLA_HIP_COMM_TIMEOUT_S", "2")
    e0 = Engine(0)
    cid = e0.comm_id(local=True)
    engines = [Engine(0) for _ in range(2)]
    t0 = time.time()
    with pytest.raises(Exception):
        engines[0].comm_init(2, 0, cid, local=True)             # rank 1 never comes
    assert 1.5 < time.time() - t0 < 30
    # a group whose second rank joins but never calls the collective
    rs = synth.make_reads(40, read_len=1500, coverage=10.0, err=0.15, seed=3)
    cid = e0.comm_id(local=True)

    def body(r):
        e = engines[r]
        e.set_reads(rs)
        e.comm_init(2, r, cid, local=True)
        if r == 1:
            return "stayed out"
        t1 = time.time()
        try:
            e.count_kmers_dist(0, 20, 17, 2, 8)
        except Exception as ex:  # noqa: BLE001
            return ("error", time.time() - t1, str(ex))
        return ("no error", time.time() - t1, "")
    out, err = _run_ranks(2, body, timeout=60)
    assert not err, err
    assert out[0][0] == "error" and 1.5 < out[0][1] < 30, out[0]
    for e in engines:
        e.comm_destroy()
        e.close()
    e0.close()


def test_two_devices_without_oversubscription(tmp_path):
    """the first run with the contexts of one call on DIFFERENT devices: bella-hip -g 2 with no BELLA_HIP_OVERSUBSCRIBE -- peer access is
    enabled between the two devices, the panel exchange and the shared formation of A' copy across them (hipMemcpyPeerAsync), the output is
    the single-GPU one.  Skipped (not passed) on a box with one device."""
    from bella_amd import _lib
    if _lib.load().bella_hip_device_count() < 2:
        pytest.skip("needs two devices")
    rs = synth.make_reads(2200, read_len=4000, err=0.15, seed=77)
    fq = str(tmp_path / "a.fastq")
    synth.write_fastq(fq, rs)
    for extra, key in ((["--skip-alignment"], "skip"), ([], "align")):
        bnums, base, _ = _run_native([fq], extra, str(tmp_path / ("one_" + key)))
        nums, data, err = _run_native([fq], extra + ["-g", "2"], str(tmp_path / ("two_" + key)))
        assert data == base and nums[:3] == bnums[:3]
        assert "GPUs = 2" in err


def test_rccl_script_with_one_rank():
    """tools/rccl_two_ranks.py with ONE process: what a one-device box can run of it -- torch.distributed rendezvous, the library's RCCL
    communicator with its self-test, the distributed counting and the panel exchange with a single rank, the comparison on rank 0"""
    import subprocess
    env = dict(os.environ, BELLA_DIST_LAYOUT="0", HSA_ENABLE_IPC_MODE_LEGACY="0", GRAFT_REPO_ROOT=ROOT, BELLA_RCCL_READS="800")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port",
                        "29531", os.path.join(ROOT, "tools", "rccl_two_ranks.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0 and "RCCL-1 OK" in out, (p.returncode, out[-1500:], p.stderr.decode(errors="replace")[-1500:])


@pytest.mark.parametrize("shared", [0, 1])
def test_two_ranks_over_real_rccl(shared):
    """two PROCESSES, one device each, the library's own RCCL communicator (tools/rccl_two_ranks.py): distributed counting, the panel
    exchange, the replicated and the shared formation of A', the pass -- merged records identical to a single-context run.  Skipped (not
    passed) on a box with one device: the first box with two runs it."""
    import subprocess
    from bella_amd import _lib
    if _lib.load().bella_hip_device_count() < 2:
        pytest.skip("needs two devices")
    env = dict(os.environ, BELLA_DIST_LAYOUT=str(shared), HSA_ENABLE_IPC_MODE_LEGACY="0", GRAFT_REPO_ROOT=ROOT)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
                        str(29533 + shared), os.path.join(ROOT, "tools", "rccl_two_ranks.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0 and "RCCL-2 OK" in out, (p.returncode, out[-1500:], p.stderr.decode(errors="replace")[-1500:])


@pytest.mark.parametrize("shared", [False, True])
@pytest.mark.parametrize("bounds", [[0, 70, 150], [0, 90, 90, 150], [0, 1, 60, 150], [0, 40, 40, 100, 149, 150]])
def test_collectives_with_several_ranks_on_the_in_process_transport(bounds, shared):
    """The N > 1 logic of bella_hip_count_kmers_dist and bella_hip_allgather_panels -- code-space split, dictionary concatenation,
    grouped send/recv offsets, uneven panels, an empty panel -- on the one GPU there is: N contexts, one host thread each, the
    library's in-process transport (bella_hip_comm_init_local) in RCCL's place.  Bit-exact against the one-shot assembly.
    shared: A' formed over the ranks by k-mer range (BELLA_TUNE_DIST_LAYOUT, the default); else on every rank: the same layout entry
    for entry, hence the same pairs in the same order."""
    from bella_amd import dist as bd
    nranks = len(bounds) - 1
    rs = synth.make_reads(150, read_len=2000, coverage=15.0, err=0.15, seed=43)
    e0 = Engine(0)
    e0.set_reads(rs)
    nk, nt, _ = e0.count_kmers(17, 2, 8)
    d0 = e0.get_dictionary()
    e0.assemble_counted()
    B1 = e0.get_B()
    e0.overlap(BellaPars(skipAlignment=True))
    p1 = e0.get_pairs()
    cid = e0.comm_id(local=True)
    engines = [Engine(0) for _ in range(nranks)]

    def body(r):
        e = engines[r]
        e.set_reads(rs)
        e.comm_init(nranks, r, cid, local=True)
        lo, n = bounds[r], bounds[r + 1] - bounds[r]
        nk_r, _, _ = e.count_kmers_dist(lo, n, 17, 2, 8)
        dic = e.get_dictionary()
        e.assemble_counted_panel(lo, n)
        e.set_partition(r, nranks)             # BEFORE the exchange: the layout it ends with holds B' for the owned columns only
        if not shared and r != 1:              # (one rank that cannot is enough: the call asks every rank, all form A' whole)
            e.set_tuning("dist_layout", 0)
        e.allgather_panels()
        B = e.get_B()
        mem = e.memory()
        assert mem.layout_shared == (1 if shared else 0)
        e.overlap(BellaPars(skipAlignment=True))
        own = e.get_pairs()
        # another partition on a context laid out for (r, nranks): the layout is rebuilt from the resident B
        e.set_partition((r + 1) % nranks, nranks)
        e.overlap(BellaPars(skipAlignment=True))
        other = e.get_pairs()
        e.set_partition(0, 1)
        e.overlap(BellaPars(skipAlignment=True))
        whole = e.get_pairs()
        return nk_r, dic, B, own, mem, other, whole
    out, err = _run_ranks(nranks, body)
    assert not err, err
    nnz = int(B1[0][-1])
    rowlen = np.diff(B1[0].astype(np.int64))
    for r in range(nranks):
        nk_r, dic, B, _, mem, _, whole = out[r]
        assert nk_r == nk
        assert np.array_equal(dic[0], d0[0]) and np.array_equal(dic[1], d0[1])
        for a, b in zip(B, B1):
            assert np.array_equal(a, b)
        # per-column layout memory follows the partition: B' entries exactly for the owned columns, 10 bytes each (+ row pointers, + the
        # allocator's slack of 1/16 + 256 bytes per array); A' and the exchanged matrix are whole
        owned = int(rowlen[r::nranks].sum())
        assert mem.owned_nnz == owned and owned <= 1.3 * nnz / nranks
        assert mem.layout_B_bytes <= 1.3 * (10 * nnz / nranks) + 4 * (rs.nreads + 2) * 1.07 + 3 * 300
        assert mem.layout_A_bytes >= 8 * nnz and mem.matrix_bytes >= 6 * nnz and mem.rowlist_bytes == 0
        assert np.array_equal(whole[0], p1[0])
    merged = bd.merge_in_reference_order([out[r][3][0] for r in range(nranks)])
    assert np.array_equal(merged, p1[0])
    merged = bd.merge_in_reference_order([out[(r - 1) % nranks][5][0] for r in range(nranks)])
    assert np.array_equal(merged, p1[0])
    for e in engines:
        e.comm_destroy()
        e.close()
    e0.close()


def test_a_failing_rank_takes_all_ranks_out_of_the_collective():
    """one rank enters bella_hip_allgather_panels without a panel: every rank returns an error (nobody waits for the exchange)"""
    rs = synth.make_reads(60, read_len=1500, coverage=10.0, err=0.15, seed=7)
    e0 = Engine(0)
    cid = e0.comm_id(local=True)
    engines = [Engine(0) for _ in range(3)]
    bounds = [0, 20, 40, 60]

    def body(r):
        e = engines[r]
        e.set_reads(rs)
        e.comm_init(3, r, cid, local=True)
        e.count_kmers_dist(bounds[r], bounds[r + 1] - bounds[r], 17, 2, 8)
        if r != 1:
            e.assemble_counted_panel(bounds[r], bounds[r + 1] - bounds[r])
        e.allgather_panels()
        return True
    out, err = _run_ranks(3, body, timeout=120)
    assert sorted(r for r, _ in err) == [0, 1, 2] and all(isinstance(e, BellaHipError) for _, e in err)

    def body3(r):                              # a rank that fails AFTER the ranks agreed on the shared formation of A', before its share begins
        e = engines[r]
        e.count_kmers_dist(bounds[r], bounds[r + 1] - bounds[r], 17, 2, 8)
        e.assemble_counted_panel(bounds[r], bounds[r + 1] - bounds[r])
        e.set_partition(r, 3)
        e.set_debug(262144 if r == 1 else 0)
        try:
            e.allgather_panels()
        finally:
            e.set_debug(0)
        return True
    out, err = _run_ranks(3, body3, timeout=120)
    assert sorted(r for r, _ in err) == [0, 1, 2] and all(isinstance(e, BellaHipError) for _, e in err)

    def body2(r):                              # a rank with a bad ARGUMENT (k = 40) takes part in the agreement too: nobody waits for its dictionary
        engines[r].count_kmers_dist(bounds[r], bounds[r + 1] - bounds[r], 40 if r == 2 else 17, 2, 8)
        return True
    out, err = _run_ranks(3, body2, timeout=120)
    assert sorted(r for r, _ in err) == [0, 1, 2] and all(isinstance(e, BellaHipError) for _, e in err)
    assert [e.code for r, e in sorted(err, key=lambda t: t[0])] == [-7, -7, -3]
    for e in engines:
        e.comm_destroy()
        e.close()
    e0.close()
