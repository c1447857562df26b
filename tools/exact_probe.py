import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from bella_amd import BellaPars, Engine
from bella_testkit import synth
rs = synth.make_reads_fast(10000, read_len=10000, coverage=30.0, err=0.15, seed=1)
eng = Engine(0); eng.set_reads(rs); eng.count_kmers(17, 2, 8); eng.assemble_counted()
pars = BellaPars()
n, _ = eng.overlap(pars)
for ex in (False, True, True):
    t0 = time.perf_counter(); npass = eng.align_pairs(pars, exact=ex); t1 = time.perf_counter()
    al = eng.get_alignments()
    print("exact" if ex else "xavier", "pairs", n, "passed", npass, "kernel ms %.1f" % eng.timings().xdrop_ms, "wall %.1f" % ((t1 - t0) * 1e3), "steps %.3g" % al["steps"].astype(np.float64).sum())
