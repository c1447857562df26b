# development aid (GPU box): A' in first-appearance order vs k-mer order: assembly and step times at 100k reads
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from bella_amd import BellaPars, Engine
from bella_testkit import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
rs = synth.make_reads_fast(n, read_len=10000, coverage=30.0, err=0.15, seed=1)
eng = Engine(0); eng.set_reads(rs)
eng.count_kmers(17, 2, 8)
ref = None
for dbg in (1024, 0, 1024, 0):
    eng.set_debug(dbg)
    eng.assemble_counted()
    tm = eng.timings(); asm = (tm.assemble_ms, tm.rows_ms, tm.layout_ms)
    eng.set_debug(2 | dbg)
    best = 1e9
    for _ in range(6):
        npairs, flops = eng.overlap(BellaPars(skipAlignment=True))
        t = eng.timings(); best = min(best, t.overlap_total_ms)
    eng.set_debug(dbg)
    eng.overlap(BellaPars(skipAlignment=True))
    pairs, _, colptr = eng.get_pairs(ext=False)
    h = (int(pairs["rid"].astype(np.uint64).sum()), int(pairs["count"].astype(np.uint64).sum()), int(pairs["seedH"].astype(np.uint64).sum()), len(pairs))
    if ref is None: ref = h
    print("debug %3d assemble %.2f rows %.2f layout %.2f | step %.3f ms (spgemm %.3f) | same %s" % (dbg, asm[0], asm[1], asm[2], best, t.spgemm_ms, h == ref), flush=True)
