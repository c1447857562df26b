# One process per GPU over real RCCL (launched by torch.distributed.run with N >= 2 ranks): the library's own communicator
# (bella_hip_comm_init), the dictionary counted across the ranks, row-block panels, ONE grouped exchange, the layout for the rank's
# columns (BELLA_DIST_LAYOUT = 0 replicated / 1 shared formation of A'), the pass; the ranks' records gathered on rank 0 and compared, record
# for record, with a single-context run of the same reads.  Prints "RCCL-N OK ..." on rank 0; any mismatch or error is a non-zero exit.
# usage: python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/rccl_two_ranks.py
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bella_amd import BellaPars, Engine  # noqa: E402
from bella_amd import dist as bd  # noqa: E402
from bella_testkit import synth  # noqa: E402


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    shared = int(os.environ.get("BELLA_DIST_LAYOUT", "0"))
    nreads = int(os.environ.get("BELLA_RCCL_READS", "3000"))
    rs = synth.make_reads(nreads, read_len=4000, err=0.15, seed=11)
    eng = Engine(local)
    eng.set_reads(rs)
    assert bd.init_comm(eng, local, "nccl"), "the library's RCCL communicator could not be created"
    lo, n = bd.block_range(rank, world, rs.nreads)
    nk, nt, _ = eng.count_kmers_dist(lo, n, 17, 2, 8)
    eng.assemble_counted_panel(lo, n)
    eng.set_partition(rank, world)
    eng.set_tuning("dist_layout", shared)
    eng.allgather_panels()
    mem = eng.memory()
    assert int(mem.layout_shared) == shared, (int(mem.layout_shared), shared)
    pars = BellaPars(skipAlignment=True)
    eng.overlap(pars)
    own = eng.get_pairs()[0]
    merged = bd.gather_pairs(own, device="cuda:%d" % local)
    ok = 1
    if rank == 0:
        one = Engine(local)
        one.set_reads(rs)
        nk1, _, _ = one.count_kmers(17, 2, 8)
        one.assemble_counted()
        one.overlap(pars)
        ref = one.get_pairs()[0]
        B1, Bn = one.get_B(), eng.get_B()
        ok = int(nk1 == nk and all(np.array_equal(a, b) for a, b in zip(B1, Bn)) and len(ref) == len(merged) and np.array_equal(ref, merged))
        print("RCCL-%d %s formation=%s reads=%d pairs=%d layout_ms=%.2f" % (world, "OK" if ok else "MISMATCH", "shared" if shared else "replicated", rs.nreads, len(ref),
                                                                            eng.timings().layout_ms), flush=True)
        one.close()
    t = torch.tensor([ok], dtype=torch.int32, device="cuda:%d" % local)
    dist.broadcast(t, src=0)
    eng.comm_destroy()
    eng.close()
    dist.destroy_process_group()
    sys.exit(0 if int(t.item()) == 1 else 1)


if __name__ == "__main__":
    main()
