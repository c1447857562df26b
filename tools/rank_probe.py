"""Development aid: what ONE rank of an N-GPU weak-scaling run computes (N x 10k reads, columns i % N == r), on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bella_amd import BellaPars, Engine
from bella_testkit import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rs = synth.make_reads(10000 * N, read_len=10000, coverage=30.0, err=0.15, seed=1)
eng = Engine(0)
eng.set_reads(rs)
t0 = time.perf_counter()
nk, nt, nd = eng.count_kmers(17, 2, 8)
t1 = time.perf_counter()
eng.assemble_counted()
t2 = time.perf_counter()
eng.set_debug(2)
tot = 0
for r in (0, N - 1):
    eng.set_partition(r, N)
    for _ in range(3):
        eng.overlap(BellaPars(skipAlignment=True))
    t3 = time.perf_counter()
    for _ in range(10):
        npairs, flops = eng.overlap(BellaPars(skipAlignment=True))
    t4 = time.perf_counter()
    tm = eng.timings()
    print("N=%d rank %d: count %.0f ms assemble %.0f ms | step %.3f ms (rows %.3f symbolic %.3f) pairs %d products %d"
          % (N, r, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t4 - t3) * 100, tm.spgemm_ms, tm.symbolic_ms, npairs, flops))
