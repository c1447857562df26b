# development aid (GPU box): timings of one stage on the synthetic set
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import numpy as np
from bella_amd import Engine, BellaPars
from bella_testkit import synth
n = int(sys.argv[1])
rs = synth.make_reads_fast(n, read_len=10000, coverage=30.0, err=0.15, seed=1)
eng = Engine(0); eng.set_reads(rs)
nk, nt, nd = eng.count_kmers(17, 2, 8)
for _ in range(2):
    eng.assemble_counted(); tm = eng.timings()
    print("reads", n, "assemble_ms %.2f rows_ms %.2f layout_ms %.2f kcount %.2f" % (tm.assemble_ms, tm.rows_ms, tm.layout_ms, tm.kcount_ms))
pars = BellaPars(skipAlignment=True)
eng.set_debug(2)
for _ in range(3):
    eng.overlap(pars); tm = eng.timings()
    print("overlap total %.3f rows %.3f sym %.3f order %.3f" % (tm.overlap_total_ms, tm.spgemm_ms, tm.symbolic_ms, tm.compact_ms))
