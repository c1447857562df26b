"""Development aid: HiFi-like set (long product lists, wide columns with a raised -u) through the device pipeline."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bella_amd import BellaPars, Engine
from bella_testkit import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
upper = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rs = synth.make_reads(n, read_len=15000, coverage=30.0, err=0.005, seed=2, mix=(1 / 3, 1 / 3, 1 / 3))
eng = Engine(0)
eng.set_reads(rs)
t0 = time.perf_counter()
nk, nt, nd = eng.count_kmers(17, 2, upper)
eng.assemble_counted()
t1 = time.perf_counter()
for _ in range(2):
    t2 = time.perf_counter()
    npairs, flops = eng.overlap(BellaPars(skipAlignment=True))
    t3 = time.perf_counter()
    tm = eng.timings()
    print("reads %d upper %d: reliable %d tuples %d | count+assemble %.1f ms | overlap %.1f ms wall (rows %.2f symbolic %.2f) pairs %d products %d"
          % (n, upper, nk, nt, (t1 - t0) * 1e3, (t3 - t2) * 1e3, tm.spgemm_ms, tm.symbolic_ms, npairs, flops))
