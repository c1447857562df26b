#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- regenerates tests/golden/ from the *reference binary*.

Runs only in the build container (needs /root/reference and oracle/_ref/ built by
oracle/build_ref.sh).  For every fixture set it writes a FASTQ, runs the reference
(`bella_ref_dump` = src/main.cpp with -DWRITEDATAMATRIX, `bella_ref`) at OMP_NUM_THREADS=1
(the only reproducible setting) and stores *data only*:

  reads.fastq.gz   input
  tuples.npz       (kmer, read, pos) from readbykmers.mtx (include/common/bellaio.h:2-47) -- pins k-mer ids
  skip.out.gz      bella_ref --skip-alignment          (overlap.hpp:580-585, 6 columns)
  align.out.gz     bella_ref (default, Xavier X-drop)  (overlap.hpp:472-473, 12 columns)
  paf.out.gz       bella_ref --paf                     (overlap.hpp:476-489)
  stdout.json      stdout protocol numbers (nkmer, nnzA, nnzC, outputted)
  meta.json        flags used

plus tests/golden/xavier_kat.json: XavierXDrop/xavierAlign known answers (xavier/demo.cpp inputs and
hand-made edge cases) computed by the reference through oracle/_ref/libbella_ref.so.
"""
import ctypes
import gzip
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bella_testkit import synth  # noqa: E402

REF = os.environ.get("BELLA_REFERENCE", "/root/reference")
RB = os.path.join(ROOT, "oracle", "_ref")
GOLD = os.path.join(ROOT, "tests", "golden")


def run_ref(binary, fastq, extra, cwd, no_aslr=False):
    with open(os.path.join(cwd, "in.txt"), "w") as f:
        f.write(fastq + "\n")  # list file must end in '\n' (kmercount.hpp:96)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    pre = ["setarch", "x86_64", "-R"] if no_aslr else []      # fixed stack addresses: xavier.h:165's uninitialised maxpos reads 0-ish
    p = subprocess.run(pre + [os.path.join(RB, binary), "-f", "in.txt", "-o", "out"] + extra, cwd=cwd, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    # exit status is meaningless (teardown abort): judge by the final stdout line
    out = p.stdout.decode(errors="replace")
    with open(os.path.join(cwd, "out.out"), "rb") as f:
        data = f.read()
    return out, data


def stdout_numbers(txt):
    nums = [ln.strip() for ln in txt.splitlines() if re.fullmatch(r"[0-9.eE+-]+", ln.strip())]
    return nums


def flagged_pairs(rs, tup, k, xdrop=7):
    """how many candidate pairs hit the reference's uninitialised-maxpos case, by the oracle's restatement"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    seqs = rs.seqs()
    Bc, Br, Bv = O.build_B(rs.nreads, tup.kmer, tup.read, tup.pos)
    _, _, pairs = O.spgemm(seqs, tup.nkmers, Bc, Br, Bv, k)
    fl = 0
    for p in pairs:
        e = O.xavier_align(seqs[int(p["rid"])], seqs[int(p["cid"])], int(p["seedH"]), int(p["seedV"]), xdrop, k)
        fl += int(e["flagged"])
    return len(pairs), fl


def make_set(name, rs, flags, aslr_check=False):
    d = os.path.join(GOLD, name)
    os.makedirs(d, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        fq = os.path.join(tmp, "reads.fastq")
        synth.write_fastq(fq, rs)
        so, align = run_ref("bella_ref_dump", fq, flags, tmp)
        tup = synth.read_mtx_tuples(os.path.join(tmp, "readbykmers.mtx"))
        # the reference emits tuples read by read at one thread; keep its order
        np.savez_compressed(os.path.join(d, "tuples.npz"), kmer=tup.kmer, read=tup.read, pos=tup.pos,
                            nkmers=np.int64(tup.nkmers))
        so2, align2 = run_ref("bella_ref", fq, flags, tmp)
        assert align2 == align, "reference not reproducible at 1 thread?"
        so3, skip = run_ref("bella_ref", fq, flags + ["--skip-alignment"], tmp)
        so4, paf = run_ref("bella_ref", fq, flags + ["--paf"], tmp)
        with open(fq, "rb") as f, gzip.open(os.path.join(d, "reads.fastq.gz"), "wb", compresslevel=9) as g:
            g.write(f.read())
        for nm, data in (("align.out.gz", align), ("skip.out.gz", skip), ("paf.out.gz", paf)):
            with gzip.open(os.path.join(d, nm), "wb", compresslevel=9) as g:
                g.write(data)
        with open(os.path.join(d, "stdout.json"), "w") as f:
            json.dump({"align": stdout_numbers(so2), "skip": stdout_numbers(so3)}, f, indent=1)
        meta = {"flags": flags, "nreads": rs.nreads, "threads": 1}
        if aslr_check:
            # the same binary with address-space randomisation off: the garbage the reference reads at xavier.h:165 changes, the
            # output files must not (those alignments never pass the threshold)
            _, align_r = run_ref("bella_ref", fq, flags, tmp, no_aslr=True)
            _, paf_r = run_ref("bella_ref", fq, flags + ["--paf"], tmp, no_aslr=True)
            assert align_r == align and paf_r == paf, "reference output depends on the uninitialised maxpos"
            k = int(flags[flags.index("-k") + 1]) if "-k" in flags else 17
            npairs, nflag = flagged_pairs(rs, tup, k)
            meta.update({"aslr_off_output_identical": True, "candidate_pairs": npairs, "flagged_pairs": nflag})
        with open(os.path.join(d, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1)
    print(name, "reads", rs.nreads, "tuples", tup.kmer.shape[0], "nkmers", tup.nkmers, "align lines",
          align.count(b"\n"), "skip lines", skip.count(b"\n"))


def repeat_genome_reads(seed):
    """reads from a genome with a duplicated 1.5 kb segment -> pairs with several overlap bins"""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, size=7000, dtype=np.uint8)
    g[4000:5500] = g[500:2000]
    rs = synth.make_reads(90, read_len=2200, err=0.12, seed=seed, genome_len=7000)
    # make_reads draws its own genome; rebuild the reads from OUR genome with the same recipe
    rng2 = np.random.default_rng(seed + 1)
    seqs, names = [], []
    for r in range(90):
        L = 2200
        st = int(rng2.integers(0, 7000 - L))
        t = g[st:st + L].copy()
        if rng2.integers(0, 2):
            t = (3 - t)[::-1]
        u = rng2.random(L)
        out = []
        for b, x in zip(t, u):
            if x < 0.036:
                continue
            if x < 0.048:
                b = (b + int(rng2.integers(1, 4))) & 3
            out.append(int(b))
            if 0.048 <= x < 0.12:
                out.append(int(rng2.integers(0, 4)))
        seqs.append(bytes(synth.BASES[np.asarray(out, dtype=np.uint8)]))
        names.append("rep%d_%d" % (r, st))
    return synth.readset_from_seqs(seqs, names)


def xavier_kats():
    lib = ctypes.CDLL(os.path.join(RB, "libbella_ref.so"))
    lib.bella_ref_xavier_xdrop.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    lib.bella_ref_xavier_align.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_char_p]
    kats = []
    # 1) the reference's own demo inputs (xavier/demo.cpp:108-121): sequences are data held by the demo
    src = open(os.path.join(REF, "xavier", "demo.cpp")).read()
    s1 = re.search(r'^\s*seq1 = "([ACGT]+)', src, re.M).group(1)
    s2 = re.search(r'^\s*seq2 = "([ACGT]+)', src, re.M).group(1)
    out = (ctypes.c_int * 6)()
    lib.bella_ref_xavier_xdrop(s1.encode(), s2.encode(), 56, 1916, 17, 15, out)
    kats.append({"kind": "xdrop", "name": "demo.cpp", "target": s1, "query": s2, "begH": 56, "begV": 1916, "k": 17,
                 "x": 15, "expect": list(out)[:5], "exit_score": out[5]})
    rng = np.random.default_rng(42)

    def rnd(n):
        return bytes(synth.BASES[rng.integers(0, 4, size=n, dtype=np.uint8)]).decode()

    def mutate(s, e):
        o = []
        for ch in s:
            u = rng.random()
            if u < e * 0.3:
                continue
            if u < e * 0.4:
                ch = "ACGT"[(("ACGT".index(ch)) + int(rng.integers(1, 4))) & 3]
            o.append(ch)
            if e * 0.4 <= u < e:
                o.append("ACGT"[int(rng.integers(0, 4))])
        return "".join(o)

    def rc(s):
        return s[::-1].translate(str.maketrans("ACGT", "TGCA"))

    def add_align(name, row, col, i, j, x=7, k=17):
        o = (ctypes.c_int * 5)()
        st = ctypes.create_string_buffer(2)
        lib.bella_ref_xavier_align(row.encode(), col.encode(), len(row), i, j, x, k, o, st)
        kats.append({"kind": "align", "name": name, "row": row, "col": col, "i": i, "j": j, "k": k, "x": x,
                     "expect": list(o), "strand": st.value.decode()[:1]})

    base = rnd(3000)
    seed = base[1500:1517]
    # identical / noisy copies, both strands; seed positions vary to hit short-prefix / short-suffix paths
    a = mutate(base[:1500], 0.15) + seed + mutate(base[1517:], 0.15)
    b = mutate(base[:1500], 0.15) + seed + mutate(base[1517:], 0.15)
    ia, ib = a.index(seed), b.index(seed)
    add_align("noisy_n", a, b, ia, ib)
    ra = rc(a)
    add_align("noisy_c", ra, b, len(a) - ia - 17, ib)
    add_align("identical_rebase", base, base, 1500, 1500)                # long exact match -> re-basing > CUTOFF
    add_align("identical_x15", base, base, 700, 700, x=15)
    add_align("short_prefix", a[ia - 10:], b[ib - 3:], 10, 3)            # prefix < 32 on both
    add_align("short_prefix_one", a[ia - 10:], b, 10, ib)                 # prefix < 32 on H only
    add_align("short_suffix", a[:ia + 17 + 9], b[:ib + 17 + 40], ia, ib)  # suffix < 32 on H (begin-instead-of-end quirk)
    add_align("short_both", a[ia - 5:ia + 17 + 5], b[ib - 5:ib + 17 + 5], 5, 5)
    add_align("exact_32_prefix", a[ia - 15:], b[ib - 15:], 15, 15)       # prefix length == 32 exactly
    add_align("exact_31_prefix", a[ia - 14:], b[ib - 14:], 14, 14)
    # unrelated sequences sharing only the seed: drifts, phase-4 exits, x-drop exits
    for t in range(6):
        u1, u2 = rnd(900 + 97 * t), rnd(1100 - 53 * t)
        p1, p2 = 100 + 60 * t, 500 - 70 * t
        u1 = u1[:p1] + seed + u1[p1 + 17:]
        u2 = u2[:p2] + seed + u2[p2 + 17:]
        add_align("junk%d" % t, u1, u2, p1, p2)
        add_align("junk%d_x2" % t, u1, u2, p1, p2, x=2)
    # unequal lengths -> one sequence ends first (phase 4), both directions
    add_align("h_ends_first", a[ia - 200:ia + 217], b, 200, ib)
    add_align("v_ends_first", a, b[ib - 150:ib + 167], ia, 150)
    add_align("k15", a, b, ia, ib, k=15)
    for e in (0.02, 0.08, 0.25, 0.4):
        aa = mutate(base[:1500], e) + seed + mutate(base[1517:], e)
        bb = mutate(base[:1500], e) + seed + mutate(base[1517:], e)
        add_align("err%g" % e, aa, bb, aa.index(seed), bb.index(seed))
    with open(os.path.join(GOLD, "xavier_kat.json"), "w") as f:
        json.dump(kats, f, indent=0)
    print("xavier KATs:", len(kats), "demo ->", kats[0]["expect"], kats[0]["exit_score"])


def logan_kats():
    """known answers of alignSeqAn (include/align.hpp:93: SeqAn's gapped X-drop extendSeed, the CPU algorithm the reference's CUDA
    kernel loganGPU/functions.cuh:223-408 ports) computed by the reference through oracle/_ref/libbella_ref.so"""
    lib = ctypes.CDLL(os.path.join(RB, "libbella_ref.so"))
    lib.bella_ref_seqan_align.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.POINTER(ctypes.c_int), ctypes.c_char_p]
    rng = np.random.default_rng(77)
    comp = bytes.maketrans(b"ACGT", b"TGCA")

    def rnd(n):
        return bytes(synth.BASES[rng.integers(0, 4, size=n, dtype=np.uint8)])

    def mutate(s, e):
        o = bytearray()
        for ch in s:
            u = rng.random()
            if u < e * 0.3:
                continue
            if u < e * 0.4:
                ch = b"ACGT"[(b"ACGT".index(ch) + int(rng.integers(1, 4))) & 3]
            o.append(ch)
            if e * 0.4 <= u < e:
                o.append(b"ACGT"[int(rng.integers(0, 4))])
        return bytes(o)

    kats = []
    for trial in range(48):
        L = int(rng.integers(80, 1400))
        base = rnd(L)
        e = [0.0, 0.02, 0.15, 0.15, 0.3][trial % 5]
        a, b = mutate(base, e), mutate(base, e)
        if trial % 6 == 5:
            b = rnd(len(b))                                # unrelated sequences sharing only the planted k-mer
        k = 17
        x = [7, 7, 3, 15, 50][trial % 5]
        i = int(rng.integers(0, len(a) - k))
        j = int(rng.integers(0, len(b) - k))
        if trial % 8 == 0:
            i = 0                                          # empty left prefix on H
        if trial % 8 == 1:
            j = len(b) - k                                 # empty right suffix on V
        b = b[:j] + a[i:i + k] + b[j + k:]
        if trial % 2:                                      # the H read on the other strand
            a, i = a[::-1].translate(comp), len(a) - i - k
        o = (ctypes.c_int * 5)()
        st = ctypes.create_string_buffer(2)
        lib.bella_ref_seqan_align(a, b, len(a), i, j, x, k, o, st)
        kats.append({"name": "t%d_e%g_x%d" % (trial, e, x), "row": a.decode(), "col": b.decode(), "i": i, "j": j, "k": k, "x": x,
                     "expect": list(o), "strand": st.value.decode()[:1]})
    with open(os.path.join(GOLD, "logan_kat.json"), "w") as f:
        json.dump(kats, f, indent=0)
    print("logan KATs:", len(kats))


def eval_kats(sets=("toy120", "toylen80", "toyhifi50"), min_overlaps=(300, 500, 1000)):
    """known answers of the reference's quality evaluator (benchmark/evaluation.cpp built as oracle/_ref/bella_eval): recall,
    precision, F1 of each golden aligned output against the truth its read names encode (r<idx>_<start>_<len>_<strand>)"""
    import gzip
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name in sets:
            d = os.path.join(GOLD, name)
            rs = synth.read_fastq(os.path.join(d, "reads.fastq.gz"))
            truth = os.path.join(tmp, "truth.txt")
            with open(truth, "w") as f:                     # evaluation.h:93-97 (not simulated): ref read start end
                for n in rs.names:
                    _, st, ln, _ = n.rsplit("_", 3)
                    f.write("genome %s %d %d\n" % (n, int(st), int(st) + int(ln)))
            bella = os.path.join(tmp, "bella.out")
            with open(bella, "wb") as f:
                f.write(gzip.open(os.path.join(d, "align.out.gz"), "rb").read())
            for mo in min_overlaps:
                r = subprocess.run([os.path.join(RB, "bella_eval"), "-G", truth, "-B", bella, "-l", str(mo)], stdout=subprocess.PIPE,
                                   check=True, env=dict(os.environ, OMP_NUM_THREADS="1"))
                lines = [ln.strip() for ln in r.stdout.decode().splitlines()]
                nums = [ln for ln in lines if re.fullmatch(r"-?[0-9.]+|-?nan", ln)]
                g = int(next(ln for ln in lines if "in the ground truth" in ln).split()[0])
                s2 = int(next(ln for ln in lines if ln.startswith("* ") and "overlaps longer" in ln).split()[1])
                t2 = int(next(ln for ln in lines if ln.startswith("* ") and "true positives" in ln).split()[1])
                out["%s/%d" % (name, mo)] = {"truth": g, "reported_x2": s2, "true_positives_x2": t2, "recall": nums[-3],
                                             "precision": nums[-2], "f1": nums[-1]}
    json.dump(out, open(os.path.join(GOLD, "eval_kat.json"), "w"), indent=1)
    return out


def junk_rich_set():
    """mostly unrelated reads (coverage 2) and k = 11: almost every candidate pair is a chance k-mer hit, a third of the
    alignments start with no positive cell (the reference's uninitialised maxpos)"""
    return synth.make_reads(220, read_len=2500, err=0.15, seed=31, coverage=2.0)


def main():
    os.makedirs(GOLD, exist_ok=True)
    if "--eval-only" in sys.argv:
        print(eval_kats())
        return
    if "--logan-only" in sys.argv:
        logan_kats()
        return
    if "--junk-only" in sys.argv:
        make_set("toyjunk220", junk_rich_set(), ["-k", "11"], aslr_check=True)
        return
    # the reference's own 3-read sanity input is a data file (sanitytests/reversecomptest.fastq)
    rs = synth.read_fastq(os.path.join(REF, "sanitytests", "reversecomptest.fastq"))
    make_set("sanity3", rs, [])
    make_set("toy120", synth.make_reads(120, read_len=2000, err=0.15, seed=7), [])
    make_set("toylen80", synth.make_reads(80, read_len=1800, err=0.15, seed=11, len_jitter=0.5, coverage=25.0), [])
    make_set("toyhifi50", synth.make_reads(50, read_len=2500, err=0.005, seed=3, mix=(1 / 3, 1 / 3, 1 / 3), coverage=5.0),
             ["-e", "0.005"])
    make_set("toyrep90", repeat_genome_reads(5), ["-e", "0.12", "-u", "30"])
    make_set("toysync60", synth.make_reads(60, read_len=2500, err=0.02, seed=9, mix=(1 / 3, 1 / 3, 1 / 3), coverage=7.0),
             ["-s", "-e", "0.02"])                                                 # syncmer selection
    make_set("toymin70", synth.make_reads(70, read_len=2200, err=0.06, seed=23, coverage=9.0), ["-w", "7", "-e", "0.06"])   # minimizers
    make_set("toyjunk220", junk_rich_set(), ["-k", "11"], aslr_check=True)
    xavier_kats()
    logan_kats()
    eval_kats()
    # read intervals of the reference's E. coli sample (dataset/ecsample-truth.txt, columns 3-4): the FASTQ itself is not in
    # the reference snapshot; the intervals shape the "ecsample-like" synthetic set of BASELINE configs[0]
    iv = np.loadtxt(os.path.join(REF, "dataset", "ecsample-truth.txt"), usecols=(2, 3), dtype=np.int64)
    np.savez_compressed(os.path.join(GOLD, "ecsample_intervals.npz"), start=iv[:, 0].astype(np.int32), end=iv[:, 1].astype(np.int32))


if __name__ == "__main__":
    main()
