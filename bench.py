#!/usr/bin/env python3
"""bench.py -- candidate overlap pairs/s of the MI355X overlap engine on BASELINE.json's configuration.

N=1 workload = configs[1]: 10k synthetic PacBio reads (10 kb templates, 15 % error, 30x), k=17, SpGEMM-only
(--skip-alignment).  A "step" is one pass of the hot path (bella_hip_overlap: estimateFLOP + symbolic + numeric
SpGEMM under the binning semiring + pair compaction) over operands already resident in HBM, i.e. exactly what the
reference times around HashSpGEMM/LocalSpGEMM (overlap.hpp:650-789).  For N>1 (one rank per GPU, launched by
torch.distributed.run) the read set grows to 10k*N reads (weak scaling), every rank holds the operands and computes
the output columns i with i % N == rank: columns are independent, so the timed step has no data-path collective.

Prints ONE JSON line (rank 0).  Extra keys: roofline (HBM, algorithmic bytes 14*nnzA + 6*F + 16*P per step over the
row-kernel time measured with HIP events on the library's stream) and cpu_baseline (the reference's own HashSpGEMM,
compiled into oracle/_ref/libbella_ref.so, timed on this box's host cores; falls back to the C port)."""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline_child(npz, threads):
    """runs in a subprocess: the reference's HashSpGEMM (or the oracle port) on the saved workload"""
    os.environ["OMP_NUM_THREADS"] = str(threads)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import re
    import _oracle as O
    from bella_amd import synth
    z = np.load(npz, allow_pickle=False)
    rs = synth.ReadSet(z["codes"], z["offsets"], ["r%d" % i for i in range(len(z["offsets"]) - 1)])
    seqs = rs.seqs()
    tk, tr, tp, nk = z["tk"], z["tr"], z["tp"], int(z["nkmers"])
    if O.have_ref():
        t0 = time.time()
        data, so, se = O.ref_hashspgemm(seqs, rs.names, nk, tk, tr, tp, npz + ".out", skip=True, mem_mb=400000.0)
        wall = time.time() - t0
        m = re.search(r"OverlapTime = ([0-9.]+) seconds", se)
        nn = [ln for ln in so.split() if ln.strip().isdigit()]
        pairs = int(nn[2])
        sec = float(m.group(1)) if m else wall
        kind = "reference"
    else:
        Bc, Br, Bv = O.build_B(rs.nreads, tk, tr, tp)
        t0 = time.time()
        _, _, p = O.spgemm(seqs, nk, Bc, Br, Bv, 17)
        sec = time.time() - t0
        wall = sec
        pairs = len(p)
        kind = "port"
        threads = 1
    print(json.dumps({"pairs": pairs, "seconds": sec, "wall": wall, "kind": kind, "cores": threads}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=10000, help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=10000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--align", action="store_true", help="configs[2]: also run the X-drop stage (not the headline line)")
    ap.add_argument("--debug-flags", type=int, default=0, help="extra bella_hip_set_debug bits (development A/B, e.g. 8 = workgroup row kernel)")
    ap.add_argument("--cpu-baseline-child", default=None)
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args()
    if a.cpu_baseline_child:
        return cpu_baseline_child(a.cpu_baseline_child, a.threads)

    import torch
    import torch.distributed as dist
    from bella_amd import BellaPars, Engine, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("BELLA_BENCH_BACKEND", "nccl")       # "gloo" only to exercise the N>1 path on a 1-GPU box
    local = local % max(1, torch.cuda.device_count())
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)
    n_gpus = world
    assert a.gpus == n_gpus, "--gpus must equal WORLD_SIZE (launch with torch.distributed.run)"

    nreads = a.reads * n_gpus
    t0 = time.time()
    rs = synth.make_reads(nreads, read_len=a.read_len, coverage=30.0, err=0.15, seed=1)
    t1 = time.time()
    # reliable k-mer dictionary + tuples on the device (bella_hip_count_kmers; the reference's SplitCount + tuple loop)
    eng = Engine(local)
    eng.set_reads(rs)                                   # reads are replicated (2 bit/base)
    nk, nt, ndistinct = eng.count_kmers(17, 2, 8)
    kcount_ms = eng.timings().kcount_ms
    # host copy of the tuples only where the CPU baseline needs it (N = 1)
    tup = synth.Tuples(*eng.get_tuples(), nk) if (world == 1 and not a.no_cpu_baseline) else None
    t2 = time.time()
    if rank == 0:
        log("[bench] reads %d (%.1f s), distinct k-mers %d, reliable %d, tuples %d (device: %.1f ms)"
            % (nreads, t1 - t0, ndistinct, nk, nt, kcount_ms))

    xchg_ms = None
    if world == 1:
        eng.assemble_counted()
        asm_ms = eng.timings().assemble_ms
    else:
        # row-block panels: each rank assembles the rows of B of ITS reads, one all-gather (RCCL over xGMI) gives every rank
        # the whole matrix, which goes back into the library through device pointers
        from bella_amd import dist as bd
        lo, npanel = bd.block_range(rank, world, nreads)
        eng.assemble_counted_panel(lo, npanel)                 # from the device-resident tuples of this rank's read block
        asm_ms = eng.timings().assemble_ms
        pc, pr, pv = eng.panel_tensors(local)
        if backend != "nccl":
            pc, pr, pv = pc.cpu(), pr.cpu(), pv.cpu()
        torch.cuda.synchronize()
        dist.barrier()
        tx = time.perf_counter()
        colptr_t, ids_t, val_t = bd.allgather_panels(pc, pr, pv)
        torch.cuda.synchronize()
        xchg_ms = (time.perf_counter() - tx) * 1e3
        dev = torch.device("cuda", local)
        eng.set_B_device(17, nk, colptr_t.to(dev), ids_t.to(dev), val_t.to(dev))
        asm_ms += eng.timings().assemble_ms
        del colptr_t, ids_t, val_t
    eng.set_partition(rank, n_gpus)
    eng.set_debug(2 | a.debug_flags)   # diagnostics array (pair_ext) off in the timed path
    pars = BellaPars(skipAlignment=True)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # measured device-copy ceiling: 1 GiB device-to-device, read + write bytes per second
    copy_gbps = None
    if rank == 0:
        try:
            src = torch.empty(1 << 30, dtype=torch.uint8, device="cuda:%d" % local)
            dst = torch.empty_like(src)
            dst.copy_(src)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                dst.copy_(src)
            e1.record()
            torch.cuda.synchronize()
            copy_gbps = 5 * 2.0 * src.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9
            del src, dst
        except Exception:
            copy_gbps = None

    for _ in range(a.warmup):
        eng.overlap(pars)
    sync()
    kern_ms = 0.0
    rows_ms = 0.0
    fold_ms = 0.0
    sym_ms = 0.0
    comp_ms = 0.0
    launches = 0
    ts = time.perf_counter()
    for _ in range(a.steps):
        npairs, flops = eng.overlap(pars)
        tm = eng.timings()
        kern_ms += tm.spgemm_ms + tm.fold_ms
        rows_ms += tm.spgemm_ms
        fold_ms += tm.fold_ms
        sym_ms += tm.symbolic_ms
        comp_ms += tm.compact_ms
        launches += tm.spgemm_launches
    sync()
    elapsed = time.perf_counter() - ts
    tt = torch.tensor([elapsed, float(npairs), float(flops), kern_ms], dtype=torch.float64,
                      device=("cuda:%d" % local) if backend == "nccl" else "cpu")
    if world > 1:
        mx = tt.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tt.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0])
        tot_pairs, tot_flops = float(sm[1]), float(sm[2])
        kern_ms_max = float(mx[3])
    else:
        tot_pairs, tot_flops, kern_ms_max = float(npairs), float(flops), kern_ms
    if rank != 0:
        dist.destroy_process_group()
        return
    if a.align:
        apars = BellaPars()
        eng.overlap(apars)
        t0 = time.perf_counter()
        npass = eng.align_pairs(apars)
        t_al = time.perf_counter() - t0
        al = eng.get_alignments()
        steps_tot = float(al["steps"].astype(np.float64).sum())
        xms = eng.timings().xdrop_ms
        log("[bench] X-drop: %d pairs, %d pass, kernel %.1f ms (wall %.1f ms) -> %.3g pairs/s, %.3g anti-diagonal steps/s, %.1f GCUPS (31 cells/step), flagged %d"
            % (len(al), npass, xms, t_al * 1e3, len(al) / (xms * 1e-3), steps_tot / (xms * 1e-3), 31 * steps_tot / (xms * 1e-3) / 1e9,
               int(al["flagged"].sum())))

    colptr, _, _ = eng.get_B()
    nnz = int(colptr[-1])
    ms_per_step = elapsed * 1e3 / a.steps
    value = tot_pairs / (elapsed / a.steps)
    # roofline of the dominant kernel (the row kernels; all tiers are the same kernel): this rank's share
    alg_bytes = 14.0 * nnz / n_gpus + 6.0 * float(flops) + 16.0 * float(npairs)
    k_ms = kern_ms / a.steps
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    out = {
        "metric": "candidate overlap pairs/sec", "value": value, "unit": "pairs/s", "n_gpus": n_gpus, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u16/u32 integer", "data": "synthetic",
        "config": {"workload": "configs[1]: %d synthetic PacBio reads (%d b templates, 15%% err, 30x) k=17 SpGEMM-only "
                               "(--skip-alignment)" % (nreads, a.read_len),
                   "reads": nreads, "nkmers": nk, "nnzA": nnz, "flops": int(tot_flops), "pairs": int(tot_pairs),
                   "partition": "columns i %% %d == rank" % n_gpus},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                     "traffic": None, "kernel": "SpGEMM = k_spgemm_rows_* (one launch set = the concurrent tier launches of a pass) + k_fold_overflow", "kernel_ms_per_step": k_ms,
                     "launches_per_step": launches / a.steps, "algorithmic_bytes_per_step": alg_bytes,
                     "measured_copy_ceiling_GBps": copy_gbps},
        "phases_ms_per_step": {"symbolic+tiering": sym_ms / a.steps, "row_kernels": rows_ms / a.steps, "overflow_fold": fold_ms / a.steps,
                               "compaction": comp_ms / a.steps},
        "kcount_ms": kcount_ms, "assemble_ms": asm_ms, "panel_allgather_ms": xchg_ms,
    }
    # HBM traffic of the same kernels from the PMC counters (tools/collect_traffic.sh, separate FETCH_SIZE / WRITE_SIZE passes,
    # gfx950 FETCH_SIZE x2 correction calibrated on our own stream): a profiling run, so read from the committed summary
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")))
        if n_gpus == 1 and nreads == 10000 and abs(tr["algorithmic_bytes"] - alg_bytes) < 1e-3 * alg_bytes:
            out["roofline"]["traffic"] = tr["hbm_bytes_corrected"]
            out["roofline"]["traffic_source"] = "profiles/r01_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, bytes per step)"
    except Exception:
        pass
    if n_gpus == 1 and not a.no_cpu_baseline:
        try:
            import tempfile
            cores = os.cpu_count() or 1
            with tempfile.TemporaryDirectory() as tmp:
                npz = os.path.join(tmp, "w.npz")
                np.savez(npz, codes=rs.codes, offsets=rs.offsets, tk=tup.kmer, tr=tup.read, tp=tup.pos, nkmers=np.int64(tup.nkmers))
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", npz, "--threads", str(cores)],
                                   stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
                line = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1]
                cb = json.loads(line)
            out["cpu_baseline"] = {"value": cb["pairs"] / cb["seconds"], "unit": "pairs/s", "cores": cb["cores"], "kind": cb["kind"],
                                   "sample": "the whole workload (%d reads): the reference's HashSpGEMM OverlapTime bracket "
                                             "(overlap.hpp:714-727), %.2f s; pairs %d" % (nreads, cb["seconds"], cb["pairs"]),
                                   "pairs_match_gpu": cb["pairs"] == int(tot_pairs)}
        except Exception as e:  # the baseline is reported, never required
            out["cpu_baseline"] = {"value": None, "unit": "pairs/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
