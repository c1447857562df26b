# development aid (GPU box): timings of one stage on the synthetic set
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import numpy as np
from bella_amd import Engine, BellaPars
from bella_testkit import synth
n = int(sys.argv[1])
rs = synth.make_reads_fast(n, read_len=10000, coverage=30.0, err=0.15, seed=1)
eng = Engine(0); eng.set_reads(rs)
eng.count_kmers(17, 2, 8); eng.assemble_counted()
pars = BellaPars()
XDBG = int(os.environ.get("XDBG", "0"))
eng.overlap(pars)
eng.set_debug(XDBG)
for variant in [int(v) for v in sys.argv[2:]] or [0]:
    eng.set_tuning("xdrop_variant", variant)
    for _ in range(2):
        t0 = time.perf_counter(); npass = eng.align_pairs(pars); w = (time.perf_counter() - t0) * 1e3
        print("variant", variant, "xdrop_ms %.2f wall %.2f passed %d" % (eng.timings().xdrop_ms, w, npass), flush=True)
