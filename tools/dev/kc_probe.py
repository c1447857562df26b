# development aid (GPU box): k-mer counting + assembly times over several calls.  usage: python tools/dev/kc_probe.py 100000 4
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
from bella_amd import Engine
from bella_testkit import synth
n = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rs = synth.make_reads_fast(n, read_len=10000, coverage=30.0, err=0.15, seed=1)
e = Engine(0); e.set_reads(rs)
for it in range(reps):
    t0 = time.perf_counter(); nk, nt, nd = e.count_kmers(17, 2, 8); w = (time.perf_counter() - t0) * 1e3
    kc = e.timings().kcount_ms
    t0 = time.perf_counter(); e.assemble_counted(); w2 = (time.perf_counter() - t0) * 1e3
    tm = e.timings()
    print("pass %d: kcount %.2f ms (wall %.1f), assemble %.2f ms (wall %.1f; rows %.2f, layout %.2f) nk %d nt %d" % (it, kc, w, tm.assemble_ms, w2, tm.rows_ms, tm.layout_ms, nk, nt), flush=True)
if len(sys.argv) > 3:
    e.set_tuning("kcount_budget", int(sys.argv[3]))
    for it in range(3):
        t0 = time.perf_counter(); nk2, nt2, nd2 = e.count_kmers(17, 2, 8); w = (time.perf_counter() - t0) * 1e3
        print("budget %s pass %d: kcount %.2f ms (wall %.1f) nk %d nt %d same %s" % (sys.argv[3], it, e.timings().kcount_ms, w, nk2, nt2, (nk2, nt2) == (nk, nt)), flush=True)
