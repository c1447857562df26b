import gzip
import json
import os
import sys

import numpy as np
import pytest


# the reference (oracle/_ref) is only reproducible at one OpenMP thread; libgomp reads this at load
os.environ.setdefault("OMP_NUM_THREADS", "1")

try:
    # PyTorch ships its own copy of the HIP runtime (torch/lib/libamdhip64.so, no soname); libbella_hip.so links /opt/rocm's.  A
    # process that initialises /opt/rocm's first and torch's afterwards holds two runtimes and torch finds "no HIP GPUs"; with torch
    # loaded first both bind to one runtime (what `pytest tests/` does anyway: collecting test_multiproc_cpu.py imports torch).
    # The tests that hand device memory to torch (panel tensors) need that order whatever subset of the suite runs.
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLD = os.path.join(ROOT, "tests", "golden")
GOLDEN_SETS = ["sanity3", "toy120", "toylen80", "toyhifi50", "toyrep90", "toysync60", "toymin70", "toyjunk220"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    def __init__(self, name):
        from bella_testkit import synth
        d = os.path.join(GOLD, name)
        self.name = name
        self.rs = synth.read_fastq(os.path.join(d, "reads.fastq.gz"))
        self.seqs = self.rs.seqs()
        self.names = self.rs.names
        z = np.load(os.path.join(d, "tuples.npz"))
        self.tk, self.tr, self.tp = z["kmer"], z["read"], z["pos"]
        self.nkmers = int(z["nkmers"])
        self.meta = json.load(open(os.path.join(d, "meta.json")))
        self.stdout = json.load(open(os.path.join(d, "stdout.json")))
        self.out = {k: gzip.open(os.path.join(d, k + ".out.gz"), "rb").read() for k in ("skip", "align", "paf")}
        fl = self.meta["flags"]
        self.err = float(fl[fl.index("-e") + 1]) if "-e" in fl else 0.15
        self.k = int(fl[fl.index("-k") + 1]) if "-k" in fl else 17
        self.window = int(fl[fl.index("-w") + 1]) if "-w" in fl else 0             # main.cpp:165
        self.syncmer = "-s" in fl                                            # main.cpp:169
        self.lower = int(fl[fl.index("-l") + 1]) if "-l" in fl else 2          # main.cpp:91-92 defaults
        self.upper = int(fl[fl.index("-u") + 1]) if "-u" in fl else 8
        self.xdrop = 7


_cache = {}


def load_golden(name):
    if name not in _cache:
        _cache[name] = Golden(name)
    return _cache[name]


@pytest.fixture(params=GOLDEN_SETS)
def golden(request):
    return load_golden(request.param)


def set_mode(eng, bits: int):
    """The parameter tables of the GPU tests name a layout / path combination by the bits earlier rounds passed to
    bella_hip_set_debug.  The layout and path CHOICES among them now go through their named tuning parameters
    (BELLA_TUNE_LAYOUT_ORDER, BELLA_TUNE_INLINE_ENTRIES, BELLA_TUNE_ROW_PATH); what is left in the debug word is fault injection
    and test switches.  set_mode(eng, 0) restores the defaults."""
    eng.set_tuning("row_path", 1 if bits & 1 else 0)
    eng.set_tuning("layout_order", 1 if bits & 1024 else 0)
    eng.set_tuning("inline_entries", 1 if bits & 32768 else 2 if bits & 65536 else 0)
    eng.set_debug(bits & ~(1 | 1024 | 32768 | 65536))
