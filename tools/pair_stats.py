"""Development aid: distribution of products per pair / fold steps on the bench workload (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bella_testkit import synth
from bella_amd.api import Engine, BellaPars

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
rs = synth.make_reads_fast(n, read_len=10000, coverage=30.0, err=0.15, seed=1)
tup = synth.count_and_tuples(rs, 17, 2, 8, device="cuda:0")
eng = Engine(0)
eng.set_reads(rs)
eng.assemble_tuples(17, tup.nkmers, tup.kmer, tup.read, tup.pos)
npairs, flops = eng.overlap(BellaPars(skipAlignment=True))
pairs, ext, _ = eng.get_pairs(True)
cnt = pairs["count"].astype(np.int64)
print("pairs", npairs, "products", flops, "sum(count)", cnt.sum(), "=> surviving comparisons", cnt.sum() - flops)
print("count percentiles", np.percentile(cnt, [50, 90, 99, 99.9, 100]))
print("nbins>1:", (ext["nbins"] > 1).sum(), "support percentiles", np.percentile(ext["support"], [50, 90, 99, 100]))
cid = pairs["cid"]
d = np.bincount(cid, minlength=n)
print("pairs per column percentiles", np.percentile(d, [50, 90, 99, 100]))
