"""Development aid: what ONE rank of the N-GPU STRONG-scaling run computes (the fixed 100k-read set of configs[3], columns
i % N == r), on one GPU: python tools/rank_probe.py [N ...]   (default 2 4 8)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bella_amd import BellaPars, Engine
from bella_testkit import synth
Ns = [int(x) for x in sys.argv[1:]] or [2, 4, 8]
rs = synth.make_reads(100000, read_len=10000, coverage=30.0, err=0.15, seed=1)
eng = Engine(0)
eng.set_reads(rs)
eng.count_kmers(17, 2, 8)
eng.assemble_counted()
eng.set_debug(2)
pars = BellaPars(skipAlignment=True)
for N in [1] + Ns:
    for r in sorted({0, N - 1}):
        eng.set_partition(r, N)
        for _ in range(3):
            eng.overlap(pars)
        t3 = time.perf_counter()
        for _ in range(6):
            npairs, flops = eng.overlap(pars)
        t4 = time.perf_counter()
        tm = eng.timings()
        print("N=%d rank %d: step %.3f ms (rows %.3f symbolic %.3f compaction %.3f) pairs %d products %d"
              % (N, r, (t4 - t3) * 1e3 / 6, tm.spgemm_ms, tm.symbolic_ms, tm.compact_ms, npairs, flops))
