import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from bella_amd import BellaPars, Engine
from bella_testkit import synth
n = int(sys.argv[1]); upper = int(sys.argv[2]); sync = int(sys.argv[3])
rs = synth.make_reads_fast(n, read_len=15000, coverage=30.0, err=0.005, seed=2, mix=(1 / 3, 1 / 3, 1 / 3))
eng = Engine(0)
eng.set_reads(rs)
nk, nt, nd = eng.count_kmers(17, 2, upper, syncmer=bool(sync))
eng.assemble_counted()
eng.set_debug(2)
for _ in range(3):
    t2 = time.perf_counter()
    npairs, flops = eng.overlap(BellaPars(skipAlignment=True))
    t3 = time.perf_counter()
tm = eng.timings()
pairs, _, colptrC = eng.get_pairs(ext=False)
print("reads %d upper %d sync %d: reliable %d tuples %d | overlap %.2f ms wall (rows %.2f) pairs %d products %d prod/pair %.1f; pairs/col max %d; launches %d"
      % (n, upper, sync, nk, nt, (t3 - t2) * 1e3, tm.spgemm_ms, npairs, flops, flops / max(npairs, 1), int(np.diff(colptrC.astype(np.int64)).max()), tm.spgemm_launches))
