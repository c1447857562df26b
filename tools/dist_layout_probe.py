"""Development aid: bella_hip_allgather_panels with the device layout formed on every rank (default) and shared over the ranks
(BELLA_TUNE_DIST_LAYOUT), N contexts on ONE GPU over the library's in-process transport, fixed 100k-read set of configs[3].
The N ranks' kernels share the one GPU, so wall time ~ the SUM of the ranks' work; per-rank figures come from the kernel trace
(rocprofv3 --kernel-trace --stats over this script: per-kernel totals / N).  Checks that both layouts give the same pairs.
python tools/dist_layout_probe.py [N] [reads]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bella_amd import BellaPars, Engine
from bella_testkit import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nreads = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
modes = [int(x) for x in os.environ.get("PROBE_MODES", "0,1,0,1").split(",")]
rs = synth.make_reads_fast(nreads, read_len=10000, coverage=30.0, err=0.15, seed=1)
e0 = Engine(0)
cid = e0.comm_id(local=True)
engines = [Engine(0) for _ in range(N)]
bounds = [nreads * r // N for r in range(N + 1)]
pars = BellaPars(skipAlignment=True)
res = {}


def body(r):
    e = engines[r]
    e.set_reads(rs)
    e.comm_init(N, r, cid, local=True)
    e.count_kmers_dist(bounds[r], bounds[r + 1] - bounds[r], 17, 2, 8)
    for it, shared in enumerate(modes):
        e.assemble_counted_panel(bounds[r], bounds[r + 1] - bounds[r])
        e.set_partition(r, N)
        e.set_tuning("dist_layout", shared)
        t0 = time.perf_counter()
        e.allgather_panels()
        t1 = time.perf_counter()
        tm = e.timings()
        mem = e.memory()
        e.overlap(pars)
        pairs = e.get_pairs()[0]
        res[(it, r)] = ((t1 - t0) * 1e3, tm.assemble_ms, tm.layout_ms, int(mem.layout_shared), pairs)


th = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(N)]
for t in th:
    t.start()
for t in th:
    t.join(1200)
assert not any(t.is_alive() for t in th)
for it, shared in enumerate(modes):
    w = [res[(it, r)] for r in range(N)]
    print("N=%d shared=%d (taken %s): allgather_panels wall %.1f ms max over ranks; exchange+layout %.1f ms, layout %.1f ms (max over ranks; all ranks on one GPU)"
          % (N, shared, sorted({x[3] for x in w}), max(x[0] for x in w), max(x[1] for x in w), max(x[2] for x in w)))
    if it:
        for r in range(N):
            assert np.array_equal(res[(it, r)][4], res[(0, r)][4]), "pairs differ between the layouts"
print("pairs equal over the modes on every rank:", sum(len(res[(0, r)][4]) for r in range(N)))
