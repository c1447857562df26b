# development aid (GPU box): the round's new paths against the older ones at a size above the tests' (default: 200,000 PacBio-like reads:
# two counting passes, 31 position bits, inline B' entries) -- tuples, pair records and a sample of alignments must be identical.
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import numpy as np
from bella_amd import Engine, BellaPars
from bella_testkit import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
t0 = time.time()
rs = synth.make_reads_fast(n, read_len=10000, coverage=30.0, err=0.15, seed=5)
print("reads %d made in %.0f s" % (n, time.time() - t0), flush=True)
out = {}
for name, dbg in (("new", 0), ("old", 16384 | 32768)):
    e = Engine(0); e.set_debug(dbg); e.set_reads(rs)
    nk, nt, nd = e.count_kmers(17, 2, 8)
    tk, tr, tp = e.get_tuples()
    e.assemble_counted()
    pars = BellaPars(skipAlignment=True)
    npairs, flops = e.overlap(pars)
    pairs, ext, colptr = e.get_pairs()
    tm = e.timings()
    print(name, "nk", nk, "nt", nt, "pairs", npairs, "flops", flops, "kcount %.1f ms assemble %.1f ms overlap %.2f ms" % (tm.kcount_ms, tm.assemble_ms, tm.overlap_total_ms), flush=True)
    out[name] = (nk, nt, nd, tk.copy(), tr.copy(), tp.copy(), npairs, flops, pairs.tobytes(), ext.tobytes() if ext is not None else b"", colptr.copy())
    del e
a, b = out["new"], out["old"]
assert a[:3] == b[:3], (a[:3], b[:3])
for i in (3, 4, 5, 10):
    assert np.array_equal(a[i], b[i]), i
assert a[6:8] == b[6:8] and a[8] == b[8] and a[9] == b[9]
print("identical: %d tuples, %d pairs" % (a[1], a[6]))
