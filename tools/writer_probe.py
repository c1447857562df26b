# development aid: writer thread sweep on the GPU box's host (no GPU work).  usage: python tools/writer_probe.py [npairs]
import numpy as np, time, os, sys, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bella_amd import BellaPars
from bella_amd.api import write_output
from bella_amd._lib import PAIR_DT
n = 100000; np_ = int(sys.argv[1]) if len(sys.argv) > 1 else 50000000
rng = np.random.default_rng(1)
names = ["read%d_%d_%d" % (i, i * 7, i * 13) for i in range(n)]
lens = rng.integers(5000, 15000, n).astype(np.uint32)
pairs = np.zeros(np_, PAIR_DT)
pairs["rid"] = rng.integers(0, n, np_); pairs["cid"] = rng.integers(0, n, np_); pairs["count"] = rng.integers(1, 9, np_)
pairs["seedH"] = rng.integers(0, 5000, np_); pairs["seedV"] = rng.integers(0, 5000, np_)
pars = BellaPars(skipAlignment=True)
d = tempfile.mkdtemp()
f = os.path.join(d, "wb.out")
for mode in os.environ.get("MODES", "0 1").split():
    os.environ["BELLA_WRITER_MODE"] = mode
    for T in [int(x) for x in os.environ.get("THREADS", "1 8 16 32 64 128 256").split()]:
        best = None
        for rep in range(2):
            open(f, "wb").close()
            st = write_output(f, pars, names, lens, pairs, nthreads=T)
            if best is None or st.seconds < best.seconds: best = st
        print("mode %s T %3d ms %8.1f measure %6.1f GB/s %.2f thr %d" % (mode, T, best.seconds * 1e3, best.format_seconds * 1e3, best.bytes / best.seconds / 1e9, best.threads), flush=True)
os.remove(f)
