"""Development aid: what ONE rank of the N-GPU STRONG-scaling run does on the fixed 100k-read set of configs[3] (columns i % N == r), on
one GPU: the device layout built FOR the rank's partition (B' for its own columns only), its bytes, and the step.
python tools/rank_probe.py [N ...]   (default 2 4 8).  Excludes the exchange and launch skew: a probe, not a scaling measurement."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bella_amd import BellaPars, Engine
from bella_testkit import synth
Ns = [int(x) for x in sys.argv[1:]] or [2, 4, 8]
rs = synth.make_reads_fast(100000, read_len=10000, coverage=30.0, err=0.15, seed=1)
eng = Engine(0)
eng.set_reads(rs)
eng.count_kmers(17, 2, 8)
pars = BellaPars(skipAlignment=True)
for N in [1] + Ns:
    for r in sorted({0, N - 1}):
        eng.set_partition(r, N)                      # before the operands are installed: the layout follows the partition
        eng.assemble_counted()
        tm = eng.timings()
        mem = eng.memory()
        eng.set_debug(2)
        for _ in range(3):
            eng.overlap(pars)
        t3 = time.perf_counter()
        for _ in range(6):
            npairs, flops = eng.overlap(pars)
        t4 = time.perf_counter()
        t2 = eng.timings()
        print("N=%d rank %d: layout %.2f ms (rows of B %.2f ms), B' %.0f MB of %d entries, A' %.0f MB | step %.3f ms (rows %.3f symbolic %.3f order %.3f) pairs %d products %d"
              % (N, r, tm.layout_ms, tm.rows_ms, mem.layout_B_bytes / 1e6, mem.owned_nnz, mem.layout_A_bytes / 1e6, (t4 - t3) * 1e3 / 6, t2.spgemm_ms,
                 t2.symbolic_ms, t2.compact_ms, npairs, flops))
