"""CPU tests of the drop-in boundary: the C-ABI library builds/loads, exports every symbol include/bella_hip.h
declares, and fails loudly without a GPU (no compute calls here)."""
import os
import re

import numpy as np
import pytest

from bella_amd import _lib, api

from bella_testkit import synth
from conftest import ROOT, load_golden
import _oracle as O


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "bella_hip.h")).read()
    declared = set(re.findall(r"\b(bella_hip_[a-z_A-Z0-9]+)\s*\(", hdr))
    assert len(declared) >= 18
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == {s[0] for s in _lib.SIGNATURES}
    assert lib.bella_hip_abi_version() == 6


def test_struct_layouts_match_header():
    assert _lib.PAIR_DT.itemsize == 16 and _lib.EXT_DT.itemsize == 8 and _lib.ALN_DT.itemsize == 32 and _lib.SEED_DT.itemsize == 12
    import ctypes
    assert ctypes.sizeof(_lib.Params) == 24 and ctypes.sizeof(_lib.Timings) == 80 and ctypes.sizeof(_lib.WriteStats) == 80 and ctypes.sizeof(_lib.IngestStats) == 40
    assert ctypes.sizeof(_lib.Memory) == 80


def test_ctypes_structs_equal_the_headers_as_a_c_compiler_sees_them(tmp_path):
    """sizeof / offsetof of every struct of include/bella_hip.h, from gcc, against the ctypes mirrors"""
    import ctypes
    import subprocess
    structs = {"bella_params": _lib.Params, "bella_timings": _lib.Timings, "bella_write_stats": _lib.WriteStats, "bella_ingest_stats": _lib.IngestStats,
               "bella_memory": _lib.Memory}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "bella_hip.h"', 'int main(void) {']
    for cname, cls in structs.items():
        src.append('printf("%s size %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            src.append('printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    for cname, dt in (("bella_pair", _lib.PAIR_DT), ("bella_pair_ext", _lib.EXT_DT), ("bella_aln", _lib.ALN_DT), ("bella_seed", _lib.SEED_DT)):
        src.append('printf("%s size %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname in dt.names:
            src.append('printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    src.append('return 0; }')
    cfile = tmp_path / "layout.c"
    cfile.write_text("\n".join(src))
    exe = str(tmp_path / "layout")
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(cfile), "-o", exe])
    got = {}
    for ln in subprocess.check_output([exe]).decode().splitlines():
        a, b, v = ln.split()
        got[(a, b)] = int(v)
    for cname, cls in structs.items():
        assert got[(cname, "size")] == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, (cname, fname)
    for cname, dt in (("bella_pair", _lib.PAIR_DT), ("bella_pair_ext", _lib.EXT_DT), ("bella_aln", _lib.ALN_DT), ("bella_seed", _lib.SEED_DT)):
        assert got[(cname, "size")] == dt.itemsize, cname
        for fname in dt.names:
            assert got[(cname, fname)] == dt.fields[fname][1], (cname, fname)


def test_no_cpu_fallback_without_gpu():
    lib = _lib.load()
    if lib.bella_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(api.BellaHipError) as e:
        api.Engine(0)
    assert e.value.code == -1


def test_product_never_imports_the_oracle():
    for dp, _, files in os.walk(os.path.join(ROOT, "bella_amd")):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "_oracle" not in txt and "liboracle" not in txt and "bella_ref" not in txt, f
                assert "bella_testkit" not in txt, f               # host-side numpy helpers are test tooling, not product


def test_host_writers_against_golden(golden, tmp_path):
    """the product's output formatting (api.format_* and the library's multi-threaded writer bella_hip_write_output, plain host
    code) fed with ORACLE results reproduces the golden files"""
    g = golden
    Bc, Br, Bv = O.build_B(g.rs.nreads, g.tk, g.tr, g.tp)
    _, _, op = O.spgemm(g.seqs, g.nkmers, Bc, Br, Bv, g.k)
    pairs = np.zeros(len(op), _lib.PAIR_DT)
    for f in ("rid", "cid", "count", "seedH", "seedV"):
        pairs[f] = op[f]
    for n, p in enumerate(op):
        a, b = g.seqs[p["rid"]], g.seqs[p["cid"]]
        pairs["flags"][n] = int(a[p["seedH"]:p["seedH"] + g.k] == b[p["seedV"]:p["seedV"] + g.k])
    assert api.format_skip(g.names, g.rs.lengths, pairs, g.k) == g.out["skip"]
    f = str(tmp_path / "o.out")
    for nt in (1, 3, 0):
        open(f, "wb").close()
        st = api.write_output(f, api.BellaPars(skipAlignment=True, kmerSize=g.k), g.names, g.rs.lengths, pairs, nthreads=nt)
        assert open(f, "rb").read() == g.out["skip"] and st.lines == len(pairs) and st.bytes == len(g.out["skip"])
    txt, oal, passed = O.align_lines(g.names, g.seqs, op, g.xdrop, g.k, g.err)
    alns = np.zeros(len(op), _lib.ALN_DT)
    phi = O.slope(g.err)
    for n, (p, a) in enumerate(zip(op, oal)):
        ok, ov = O.post_align(a["score"], a["begV"], a["endV"], a["begH"], a["endH"], len(g.seqs[p["rid"]]), len(g.seqs[p["cid"]]), phi)
        alns[n] = (a["score"], a["begH"], a["endH"], a["begV"], a["endV"], ov, a["strand"], ok, a["steps"], a["flagged"])
    assert api.format_aligned(g.names, g.rs.lengths, pairs, alns) == txt
    ptxt, _, _ = O.align_lines(g.names, g.seqs, op, g.xdrop, g.k, g.err, paf=True)
    assert api.format_aligned(g.names, g.rs.lengths, pairs, alns, paf=True) == ptxt
    for paf, want in ((False, txt), (True, ptxt)):
        open(f, "wb").write(b"head\n")                                     # the writer appends (overlap.hpp:613)
        st = api.write_output(f, api.BellaPars(kmerSize=g.k, errorRate=g.err, outputPaf=paf), g.names, g.rs.lengths, pairs, alns, nthreads=5)
        assert open(f, "rb").read() == b"head\n" + want
        assert st.lines == int(alns["passed"].sum()) and st.aligned_pairs == len(pairs)
        assert (st.bases_passed + st.bases_failed) % 2 ** 64 == st.aligned_bases      # (size_t sums of int differences, as in the reference)


def test_synth_dictionary_matches_reference_counts():
    """count_and_tuples (caller-side numpy) gives the reference's reliable-k-mer COUNT and tuple multiset on a
    golden set (ids differ: the reference's are cuckoo-table order)."""
    g = load_golden("toy120")
    t = synth.count_and_tuples(g.rs, 17, 2, 8)
    assert t.nkmers == g.nkmers and len(t.kmer) == len(g.tk)
    assert np.array_equal(t.read, g.tr) and np.array_equal(t.pos, g.tp)
    # same partition of tuples into k-mer classes
    a = np.unique(np.stack([t.kmer, g.tk]), axis=1)
    assert a.shape[1] == g.nkmers


def test_native_cli_flags_without_a_device(tmp_path):
    """bella_amd/bin/bella-hip (built by bella_amd/build.py): the reference's option surface (main.cpp:65-175).  What needs no device:
    --help and a missing -f / -o print the usage and exit 0 as the reference does (:97-145); flags the reference has and this program
    does not build (--hopc, --estimate, --split-count > 1), unknown options and malformed values are refused with a message; a list
    file whose last line has no newline names no file (kmercount.hpp:96)."""
    import subprocess
    from bella_amd import build as b
    exe = b.build_cli()
    run = lambda args: subprocess.run([exe] + args, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    for args in (["--help"], ["-h"], [], ["-f", "in.txt"], ["-o", "x"], ["-f", "in.txt", "-o", "x", "--score-deviation", "1.5"]):
        p = run(args)
        assert p.returncode == 0 and b"--skip-alignment" in p.stdout and b"-u, --upper-freq" in p.stdout, args
    for bad, msg in ((["--hopc"], b"--hopc"), (["--estimate"], b"--estimate"), (["--split-count", "2"], b"--split-count"), (["--bogus"], b"does not exist"),
                     (["-k", "seventeen"], b"not an integer"), (["-k"], b"missing an argument"), (["--paf=1"], b"takes no value"), (["-k", "40"], b"[1,32]")):
        p = run(["-f", "in.txt", "-o", "x"] + bad)
        assert p.returncode == 1 and b"bella-hip:" in p.stderr and msg in p.stderr, (bad, p.stderr)
    open(tmp_path / "in.txt", "w").write("/nonexistent/a.fastq")           # no trailing newline
    p = run(["-f", "in.txt", "-o", "x", "-k17", "--kmer=17", "--xdrop", "7"])
    assert p.returncode == 1 and b"no FASTQ file" in p.stderr
    p = run(["-f", "missing.txt", "-o", "x"])
    assert p.returncode == 1 and b"Could not open missing.txt" in p.stderr


def test_bench_xdrop_instruction_constants_match_the_isa_listing():
    """bench.py's in-run X-drop utilisation figure (xdrop.valu_issue_frac) multiplies the measured anti-diagonal steps by per-step
    instruction counts: those constants are the "every step" line of the committed ISA listing (tools/xdrop_isa.py)"""
    import re
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    txt = open(os.path.join(ROOT, "profiles", "r06_xdrop_isa.txt")).read()
    m = re.search(r"every step: (\d+) VALU = (\d+) at ~4\.15 cycles \+ (\d+) at ~2\.4 cycles", txt)
    o = re.search(r"outer loop \(once per 16 steps\): \d+ instructions, (\d+) VALU", txt)
    assert m and o
    src = open(os.path.join(ROOT, "bench.py")).read()
    get = lambda name: int(re.search(r"^%s = (\d+)" % name, src, re.M).group(1))
    assert get("XDROP_VALU_HALF_RATE") == int(m.group(2)) and get("XDROP_VALU_FULL_RATE") == int(m.group(3))
    assert int(m.group(1)) == int(m.group(2)) + int(m.group(3))
    assert get("XDROP_VALU_CHECKPOINT") == round(int(o.group(1)) / 16.0)
