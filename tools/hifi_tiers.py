# development aid (GPU box): the HiFi-like probe under different LDS-tier limits (columns above the last tier take the wide path)
# usage: python tools/hifi_tiers.py 10000 40
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from bella_amd import BellaPars, Engine
from bella_testkit import synth
n = int(sys.argv[1]); upper = int(sys.argv[2])
rs = synth.make_reads_fast(n, read_len=15000, coverage=30.0, err=0.005, seed=2, mix=(1 / 3, 1 / 3, 1 / 3))
eng = Engine(0)
eng.set_reads(rs)
eng.count_kmers(17, 2, upper, syncmer=True)
eng.assemble_counted()
eng.set_debug(2)
base = [384, 689, 1000, 1394, 2048, 2752, 3712, 4600, 5568, 8192, 11008]
for top in (None, 2752, 2048, 1394, 1000, 689, 384, None):
    if top is None: eng.set_tuning("lds_tiers")
    else: eng.set_tuning("lds_tiers", *[c for c in base if c <= top])
    ts = []
    for _ in range(5):
        t2 = time.perf_counter(); npairs, flops = eng.overlap(BellaPars(skipAlignment=True)); ts.append((time.perf_counter() - t2) * 1e3)
    print("tiers up to %s: %s ms (pairs %d)" % (top, " ".join("%.2f" % t for t in ts), npairs), flush=True)
