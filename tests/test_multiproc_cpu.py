"""CPU test of the N>1 plumbing: 2 processes, gloo.  Each rank plays a GPU that computed the columns of its partition
(the oracle's pairs filtered by cid % 2 == rank stand in for the kernel output), gathers, and rank 0 must hold the whole
result in the reference's order."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import _oracle as O
    from bella_amd import _lib, dist as bd
    from conftest import load_golden
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = load_golden("toy120")
    Bc, Br, Bv = O.build_B(g.rs.nreads, g.tk, g.tr, g.tp)
    _, _, op = O.spgemm(g.seqs, g.nkmers, Bc, Br, Bv, g.k)
    whole = np.zeros(len(op), _lib.PAIR_DT)
    for f in ("rid", "cid", "count", "seedH", "seedV"):
        whole[f] = op[f]
    first, stride = bd.partition(rank, world)
    mine = whole[whole["cid"] % stride == first]
    merged = bd.gather_pairs(mine)
    ok = True
    if rank == 0:
        ok = merged.tobytes() == whole.tobytes() and len(merged) > 1000
    else:
        ok = merged is None
    # the panel exchange: every rank assembles the rows of B of its read block (oracle stands in for k_asm_rows) and
    # all-gathers; every rank must end up with the reference's whole B
    import torch
    lo, n = bd.block_range(rank, world, g.rs.nreads)
    sel = (g.tr >= lo) & (g.tr < lo + n)
    pc, pr, pv = O.build_B(n, g.tk[sel], (g.tr[sel] - lo).astype(np.uint32), g.tp[sel])
    colptr, ids, val = bd.allgather_panels(torch.from_numpy(np.diff(pc.astype(np.int64)).astype(np.int32)),
                                           torch.from_numpy(pr.astype(np.int32)), torch.from_numpy(pv.astype(np.int16)))
    ok = ok and np.array_equal(colptr.numpy().astype(np.uint32), Bc) and np.array_equal(ids.numpy().astype(np.uint32), Br) \
        and np.array_equal(val.numpy().astype(np.uint16), Bv)
    q.put((rank, bool(ok), len(mine)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_partition_gather_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=300) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    assert sum(n for _, _, n in res) > 1000 and min(n for _, _, n in res) > 100      # both ranks had work
