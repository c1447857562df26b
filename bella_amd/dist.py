"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in CPU tests).

Output columns of C = A*A^T are independent: rank r computes the columns i with i % world == r
(`Engine.set_partition`) -- cyclic, because the strictly-lower-triangular work of column i falls with i.  The timed step has
no collective; gathering the per-rank pair records to rank 0 (16 B per pair) is the only exchange after it."""
from __future__ import annotations

import numpy as np


def partition(rank: int, world: int):
    """(first, stride) for Engine.set_partition"""
    return rank % world, world


def merge_in_reference_order(parts):
    """Per-rank pair arrays (each: its columns ascending, slot order inside a column) -> the reference's 1-thread order."""
    merged = np.concatenate(parts) if len(parts) else np.zeros(0)
    order = np.argsort(merged["cid"], kind="stable")
    return merged[order]


def gather_pairs(pairs_local: np.ndarray, device=None, group=None):
    """all ranks call; returns the merged array on rank 0 (None elsewhere).  Records travel as raw bytes, padded to the
    longest rank so that one all_gather over xGMI (or gloo in tests) carries them."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = torch.device(device) if device is not None else torch.device("cpu")
    n = torch.tensor([pairs_local.nbytes], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    mx = int(max(int(s.item()) for s in sizes))
    buf = torch.zeros(max(mx, 1), dtype=torch.uint8, device=dev)
    if pairs_local.nbytes:
        buf[: pairs_local.nbytes] = torch.from_numpy(pairs_local.view(np.uint8).reshape(-1).copy()).to(dev)
    outs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf, group=group)
    if rank != 0:
        return None
    parts = []
    for o, s in zip(outs, sizes):
        nb = int(s.item())
        parts.append(np.frombuffer(o[:nb].cpu().numpy().tobytes(), dtype=pairs_local.dtype))
    return merge_in_reference_order(parts)


def block_range(rank: int, world: int, nreads: int):
    """contiguous read block owned by `rank` for ASSEMBLY (the compute partition is cyclic, see partition())"""
    lo = (nreads * rank) // world
    hi = (nreads * (rank + 1)) // world
    return lo, hi - lo


def allgather_panels(rowcnt, rowids, values, group=None):
    """The one exchange of the multi-GPU path: every rank contributes the rows of B of its read block (torch tensors on the
    rank's GPU -- RCCL over xGMI -- or on the CPU with gloo in tests) and receives the whole matrix:
    returns (colptr int32[nreads+1], rowids int32[nnz], values int16[nnz]) on the same device."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = rowcnt.device
    sizes = torch.tensor([rowcnt.numel(), rowids.numel()], dtype=torch.int64, device=dev)
    allsz = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(allsz, sizes, group=group)
    allsz = [(int(s[0].item()), int(s[1].item())) for s in allsz]
    mr = max(1, max(a for a, _ in allsz))
    mn = max(1, max(b for _, b in allsz))

    def gather(x, pad_to, dtype):
        buf = torch.zeros(pad_to, dtype=dtype, device=dev)
        buf[: x.numel()] = x
        raw = buf.view(torch.uint8)                      # bytes: every backend moves them (gloo has no int16)
        outs = [torch.zeros_like(raw) for _ in range(world)]
        dist.all_gather(outs, raw, group=group)
        return [o.view(dtype) for o in outs]

    g_cnt = gather(rowcnt, mr, torch.int32)
    g_ids = gather(rowids, mn, torch.int32)
    g_val = gather(values, mn, torch.int16)
    cnt = torch.cat([g[:a] for g, (a, _) in zip(g_cnt, allsz)])
    ids = torch.cat([g[:b] for g, (_, b) in zip(g_ids, allsz)])
    val = torch.cat([g[:b] for g, (_, b) in zip(g_val, allsz)])
    colptr = torch.zeros(cnt.numel() + 1, dtype=torch.int32, device=dev)
    colptr[1:] = torch.cumsum(cnt, 0).to(torch.int32)
    return colptr, ids, val


def init_comm(eng, device_index: int, backend: str = "nccl") -> bool:
    """All ranks call once: creates the library's own RCCL communicator (bella_hip_comm_init); the 128-byte id travels over
    torch.distributed.  Returns False (and leaves the context without a communicator) when that is not possible -- "gloo"
    backend in CPU-rendezvous tests, or librccl missing: the callers then use the torch.distributed paths.
    The sequence of collectives is the same on every rank whatever fails where: (1) all-reduce "librccl loads here" -- the only
    way the collective bella_hip_comm_init can fail on SOME ranks, so nobody enters it unless everybody can; (2) broadcast of
    (flag, id) from rank 0 -- a rank 0 that could not make the id says so in the flag; (3) comm_init; (4) all-reduce of its status."""
    import sys
    import torch
    import torch.distributed as dist
    if backend != "nccl":
        return False
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = torch.device("cuda", device_index)
    try:
        can = 1 if eng.comm_available() else 0
    except Exception:
        can = 0
    flag = torch.tensor([can], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        if rank == 0:
            print("[bella_amd.dist] librccl is not loadable on every rank; using torch.distributed", file=sys.stderr)
        return False
    raw, ok_id = bytes(128), 1
    if rank == 0:
        try:
            raw = eng.comm_id()
        except Exception as e:
            print("[bella_amd.dist] bella_hip_comm_id failed (%r); using torch.distributed" % (e,), file=sys.stderr)
            ok_id = 0
    t = torch.tensor([ok_id] + list(raw), dtype=torch.uint8, device=dev)
    dist.broadcast(t, src=0)                                 # unconditional: every rank is here
    vals = t.cpu().tolist()
    if vals[0] == 0:
        return False
    ok = 1
    try:
        eng.comm_init(world, rank, bytes(vals[1:]))
    except Exception as e:
        print("[bella_amd.dist] rank %d: bella_hip_comm_init failed (%r)" % (rank, e), file=sys.stderr)
        ok = 0
    flag = torch.tensor([ok], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)              # all ranks take the same path
    if int(flag.item()) == 0:
        try:
            eng.comm_destroy()
        except Exception:
            pass
        return False
    return True


def exchange_panels(eng, device_index: int, backend: str = "nccl", k: int = 17, have_comm: bool = False):
    """All ranks call after Engine.assemble_*_panel: gives every rank the whole matrix.  With the library's communicator
    (init_comm): bella_hip_allgather_panels, one grouped point-to-point exchange straight between the device arrays.  Otherwise
    torch.distributed's all_gather (RCCL too with the "nccl" backend; the only path with "gloo").  Returns the path's name."""
    import torch
    if have_comm:
        eng.allgather_panels()
        return "bella_hip_allgather_panels (RCCL send/recv group)"
    pc, pr, pv = eng.panel_tensors(device_index)
    if backend != "nccl":
        pc, pr, pv = pc.cpu(), pr.cpu(), pv.cpu()
    colptr_t, ids_t, val_t = allgather_panels(pc, pr, pv)
    dev = torch.device("cuda", device_index)
    eng.set_B_device(k, eng.nkmers_counted, colptr_t.to(dev), ids_t.to(dev), val_t.to(dev))
    return "torch.distributed all_gather (%s)" % backend
