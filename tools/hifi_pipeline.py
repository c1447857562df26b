"""Development aid: BASELINE configs[4]'s pipeline (HiFi-like reads, syncmer selection, -u 40) at a size one GPU holds comfortably: device
times of every stage.  python tools/hifi_pipeline.py [reads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bella_amd import BellaPars, Engine
from bella_testkit import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
rs = synth.make_reads_fast(n, read_len=15000, coverage=30.0, err=0.005, seed=2, mix=(1 / 3, 1 / 3, 1 / 3))
eng = Engine(0)
eng.set_reads(rs)
nk, nt, nd = eng.count_kmers(17, 2, 40, syncmer=True)
kc = eng.timings().kcount_ms
eng.assemble_counted()
asm = eng.timings().assemble_ms
pars = BellaPars(errorRate=0.005)
for _ in range(2):
    t0 = time.perf_counter(); npairs, flops = eng.overlap(pars); t1 = time.perf_counter()
ov = (t1 - t0) * 1e3
npass = eng.align_pairs(pars)
xd = eng.timings().xdrop_ms
print("reads %d (%.0f Mb): syncmer dictionary %d, tuples %d | count %.1f ms, assemble %.1f ms, overlap %.1f ms (%d pairs, %d products, %.1f G products/s), "
      "X-drop %.1f ms (%d passed, %.2f M pairs/s) | device total %.1f ms"
      % (n, rs.codes.size / 1e6, nk, nt, kc, asm, ov, npairs, flops, flops / ov / 1e6, xd, npass, npairs / xd / 1e3, kc + asm + ov + xd))
