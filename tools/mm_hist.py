# development aid (GPU box): distribution of products per pair on the HiFi-like probe
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, scipy.sparse as sp
from bella_amd import Engine
from bella_testkit import synth
n = int(sys.argv[1]); upper = int(sys.argv[2]); sync = int(sys.argv[3])
rs = synth.make_reads_fast(n, read_len=15000, coverage=30.0, err=0.005, seed=2, mix=(1 / 3, 1 / 3, 1 / 3))
eng = Engine(0); eng.set_reads(rs)
nk, nt, nd = eng.count_kmers(17, 2, upper, syncmer=bool(sync))
eng.assemble_counted()
colptr, rowids, values = eng.get_B()
A = sp.csr_matrix((np.ones(len(rowids), np.float32), rowids.astype(np.int64), colptr.astype(np.int64)), shape=(n, nk))
C = sp.triu(A @ A.T, k=1).tocoo()
mm = C.data.astype(np.int64)
print("pairs", len(mm), "products", mm.sum(), "mean", mm.mean())
edges = [0, 1, 2, 8, 32, 96, 256, 512, 1024, 2048, 4096, 8192, 32768, 1 << 30]
for a, b in zip(edges[:-1], edges[1:]):
    s = (mm > a) & (mm <= b)
    print("mm in (%6d, %6d]: pairs %7d (%.1f%%) products %10d (%.1f%%)" % (a, b, s.sum(), 100 * s.mean(), mm[s].sum(), 100 * mm[s].sum() / mm.sum()))
col = np.diff(colptr.astype(np.int64))
print("entries per read: mean %.0f max %d" % (col.mean(), col.max()))
