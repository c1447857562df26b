"""Reads -> overlaps -> recall / precision / F1 on the synthetic bench set (run on the GPU box): the whole device pipeline
(count_kmers, assemble_counted, overlap, align_pairs) followed by the evaluator mirror (bella_amd/evaluate.py)."""
import io
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bella_amd import BellaPars, Engine, evaluate as ev, hash_spgemm

from bella_testkit import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
rs = synth.make_reads_fast(n, read_len=10000, coverage=30.0, err=0.15, seed=1)
eng = Engine(0)
t0 = time.perf_counter()
eng.set_reads(rs)
t1 = time.perf_counter()
eng.count_kmers(17, 2, 8)
eng.assemble_counted()
t2 = time.perf_counter()
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "o.out")
    t3 = time.perf_counter()
    hash_spgemm(eng, BellaPars(skipAlignment=False), out, stdout=io.StringIO())
    t4 = time.perf_counter()
    data = open(out, "rb").read()
tm = eng.timings()
print("reads %d: set_reads %.1f ms, count+assemble %.1f ms (device %.1f + %.1f), HashSpGEMM incl. alignment and file %.1f ms "
      "(spgemm %.2f ms, xdrop %.1f ms), %d lines" % (n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, tm.kcount_ms, tm.assemble_ms, (t4 - t3) * 1e3,
                                                     tm.overlap_total_ms, tm.xdrop_ms, data.count(b"\n")))
for mo in (2000, 1000):
    G = ev.truth_pairs(ev.truth_from_names(rs.names), mo)
    S = ev.read_bella_output(data, mo)
    r = ev.evaluate(S, G)
    print("min overlap %d: truth %d (both orders), reported %d, recall %.2f precision %.2f F1 %.2f" % (mo, r["truth"], r["reported"],
                                                                                                   r["recall"], r["precision"], r["f1"]))
