"""development aid (GPU box): configs[4] at full size -- 1M HiFi reads through the device pipeline, stage by stage with wall times and
host memory, to size the parity test.  usage: python tools/dev/hifi_big_probe.py NREADS [UPPER]"""
import os, sys, time, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401  (first: one HIP runtime)
from bella_amd import BellaPars, Engine
from bella_testkit import synth
n = int(sys.argv[1]); upper = int(sys.argv[2]) if len(sys.argv) > 2 else 8
def rss(): return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6
t0 = time.time(); rs = synth.make_reads_fast(n, read_len=15000, coverage=30.0, err=0.005, seed=2, mix=(1 / 3, 1 / 3, 1 / 3))
print("reads %.1f s, bases %.2f G, rss %.1f GB" % (time.time() - t0, rs.offsets[-1] / 1e9, rss()), flush=True)
eng = Engine(0)
t0 = time.time(); eng.set_reads(rs); print("set_reads %.1f s rss %.1f" % (time.time() - t0, rss()), flush=True)
t0 = time.time(); nk, nt, nd = eng.count_kmers(17, 2, upper, syncmer=True); print("count %.1f s (device %.0f ms): nk %d nt %d" % (time.time() - t0, eng.timings().kcount_ms, nk, nt), flush=True)
t0 = time.time(); eng.assemble_counted(); print("assemble %.1f s (%.0f ms)" % (time.time() - t0, eng.timings().assemble_ms), flush=True)
pars = BellaPars(errorRate=0.005)
t0 = time.time(); colS, nS, fS = eng.count_pairs(pars); print("count_pairs %.1f s: pairs %d products %d" % (time.time() - t0, nS, fS), flush=True)
t0 = time.time(); npairs, flops = eng.overlap(pars); print("overlap %.1f s (%.1f ms): %d %d" % (time.time() - t0, eng.timings().overlap_total_ms, npairs, flops), flush=True)
t0 = time.time(); npass = eng.align_pairs(pars); print("align %.1f s (%.0f ms): passed %d" % (time.time() - t0, eng.timings().xdrop_ms, npass), flush=True)
t0 = time.time(); tk, tr, tp = eng.get_tuples(); print("get_tuples %.1f s rss %.1f" % (time.time() - t0, rss()), flush=True)
