#!/usr/bin/env python3
"""bench.py -- candidate overlap pairs/s of the MI355X overlap engine on BASELINE.json's configurations.

A "step" is one pass of the hot path (bella_hip_overlap: estimateFLOP + tiering + numeric SpGEMM under the binning semiring +
pair compaction) over operands already resident in HBM, i.e. what the reference times around HashSpGEMM/LocalSpGEMM
(overlap.hpp:650-789).  The step ends with the pair records in HBM (no D2H of the records, no writer: the reference's
OverlapTime bracket ends with host-resident vectors; the PCIe-inclusive numbers are in DESIGN.md).

N = 1 (default): the headline line is configs[1] -- 10k synthetic PacBio reads (10 kb templates, 15 % error, 30x), k=17,
SpGEMM-only.  The same line carries
  * "xdrop":       configs[2], the X-drop stage on the 10k set's candidate pairs (outside the timed SpGEMM loop),
  * "config_100k": configs[3]'s read set (100k reads) on ONE GPU: ms/step, pairs/s, roofline, and a CPU baseline on a bounded
                   sample (the reference cannot finish the whole set within the bench's time budget).
N > 1 (torch.distributed.run, one rank per GPU): STRONG scaling on the fixed 100k-read set of configs[3]: reads replicated, every
rank assembles the rows of B of its read block, one all-gather of the panels (RCCL) gives every rank the matrix, rank r computes
the output columns i % N == r; the timed step has no collective.  Rank 0 also times the whole set on its own GPU, so the line
carries the same-workload single-GPU time next to the N-GPU one.

Prints ONE JSON line (rank 0).  roofline: HBM, algorithmic bytes 14*nnzA + 6*F + 16*P per step over the row-kernel time measured
with HIP events on the library's stream.  cpu_baseline: the reference's own HashSpGEMM (oracle/_ref/libbella_ref.so) on this
box's host cores; falls back to the C port."""
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
BIG_READS = 100000       # configs[3]
SAMPLE_READS = 30000     # CPU baseline sample of the 100k set


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline_child(npz, threads):
    """runs in a subprocess: the reference's HashSpGEMM (or the oracle port) on the saved workload"""
    os.environ["OMP_NUM_THREADS"] = str(threads)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import re
    import _oracle as O
    from bella_testkit import synth
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


def run_cpu_baseline(codes, offsets, tk, tr, tp, nkmers, what):
    """the reference's HashSpGEMM on the given reads/tuples in a child process; returns the cpu_baseline object"""
    import tempfile
    try:
        cores = os.cpu_count() or 1
        with tempfile.TemporaryDirectory() as tmp:
            npz = os.path.join(tmp, "w.npz")
            np.savez(npz, codes=codes, offsets=offsets, tk=tk, tr=tr, tp=tp, nkmers=np.int64(nkmers))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", npz, "--threads", str(cores)],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
            line = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1]
            cb = json.loads(line)
        return {"value": cb["pairs"] / cb["seconds"], "unit": "pairs/s", "cores": cb["cores"], "kind": cb["kind"],
                "sample": "%s: the reference's HashSpGEMM OverlapTime bracket (overlap.hpp:714-727), %.2f s; pairs %d"
                          % (what, cb["seconds"], cb["pairs"]), "pairs": cb["pairs"]}
    except Exception as e:  # the baseline is reported, never required
        return {"value": None, "unit": "pairs/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,), "pairs": None}


def timed_passes(eng, pars, steps, warmup, sync):
    for _ in range(warmup):
        eng.overlap(pars)
    sync()
    acc = {"rows": 0.0, "fold": 0.0, "sym": 0.0, "comp": 0.0, "launches": 0}
    ts = time.perf_counter()
    for _ in range(steps):
        npairs, flops = eng.overlap(pars)
        tm = eng.timings()
        acc["rows"] += tm.spgemm_ms
        acc["fold"] += tm.fold_ms
        acc["sym"] += tm.symbolic_ms
        acc["comp"] += tm.compact_ms                       # k_order_*: slot order + move to the final place
        acc["launches"] += tm.spgemm_launches
        acc["retry"] = max(acc.get("retry", 0), int(tm.retry_columns))
    sync()
    acc["elapsed"] = time.perf_counter() - ts
    acc["npairs"], acc["flops"], acc["steps"] = npairs, flops, steps
    return acc


def roofline_of(acc, nnz_share, copy_gbps, traffic_key, expand_ms=0.0, layout="default"):
    """HBM roofline of the SpGEMM numeric phase.  algorithmic bytes = 14*nnzA + 6*F + 16*P; the 14*nnzA term is the operand
    traffic of the product expansion (B' entries + the gather of A's lists).  In the DEFAULT layout every pass does that expansion itself,
    so frac == frac_incl_expansion by construction.  With BELLA_TUNE_ROW_LISTS the expansion runs once at assembly time (expand_ms,
    bella_timings.expand_ms) and the pass streams ready-made products: frac_incl_expansion = bytes / (expansion + numeric phase) is the
    like-for-like figure, frac alone over-credits the pass."""
    alg_bytes = 14.0 * nnz_share + 6.0 * float(acc["flops"]) + 16.0 * float(acc["npairs"])
    k_ms = (acc["rows"] + acc["fold"] + acc["comp"]) / acc["steps"]
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    incl = alg_bytes / ((k_ms + expand_ms) * 1e-3) / 1e9 if k_ms > 0 else 0.0
    r = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
         "frac_incl_expansion": incl / HBM_PEAK_GBPS, "expansion_ms_outside_the_step": expand_ms, "layout": layout,
         "kernel": "SpGEMM numeric phase = k_spgemm_rows_lds (one launch set = the concurrent LDS-class launches of a pass; in the default layout "
                   "it expands B' x A' itself) + k_fold_overflow + k_order_wave/_block (the reference's slot order inside every column and the "
                   "move of the records to their final place)",
         "kernel_ms_per_step": k_ms, "launches_per_step": acc["launches"] / acc["steps"], "algorithmic_bytes_per_step": alg_bytes,
         "columns_redone_on_global_path": acc.get("retry", 0),
         "measured_copy_ceiling_GBps": copy_gbps}
    # HBM traffic of the same kernels from the PMC counters (tools/collect_traffic.sh: separate FETCH_SIZE / WRITE_SIZE passes,
    # gfx950 FETCH_SIZE x2 correction calibrated on our own stream): a separate profiling run, read from the committed summary and
    # accepted only if it was taken on the same workload AND layout
    for tag in ("r06", "r05"):
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "%s_hbm_traffic.json" % tag)))[traffic_key]
            if abs(tr["algorithmic_bytes"] - alg_bytes) < 1e-3 * alg_bytes and tr.get("layout", "default") == layout:
                r["traffic"] = tr["hbm_bytes_corrected"]
                r["traffic_uncorrected"] = tr["FETCH_SIZE_raw_bytes"] + tr["WRITE_SIZE_raw_bytes"]
                r["traffic_source"] = ("profiles/%s_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, bytes per step; a separate "
                                       "profiled run).  traffic = 2 x FETCH_SIZE + WRITE_SIZE: the gfx950 correction of the guide, calibrated on our own "
                                       "coalesced stream (0.500).  profiles/r04_mem_counters_100k.txt shows every L2 miss of these kernels leaving as a 128-byte "
                                       "fabric request (TCC_EA0_RDREQ_128B = TCC_EA0_RDREQ), so the correction holds for the random gathers of A' too: "
                                       "the traffic above the algorithmic bytes is whole lines fetched for 8 ... 24-byte list tails" % tag)
                # what the pass could reach at this traffic if it moved it at the copy ceiling measured in this run
                if copy_gbps and r["traffic"]:
                    r["ceiling_frac"] = copy_gbps / (r["traffic"] / alg_bytes) / HBM_PEAK_GBPS
                break
        except Exception:
            pass
    return r


def layout_ab_record(eng, pars, info, copy_gbps, sync, steps, traffic_key):
    """The two device layouts side by side on the operands `eng` holds (counted tuples resident): for each, the layout time, the COLD
    first pass (device time), their sum -- what a one-shot call pays -- and the warm step.  Justifies the default."""
    out = {}
    for name, rl in (("default", 0), ("row_lists", 1)):
        eng.set_tuning("row_lists", rl)
        eng.assemble_counted()
        tm = eng.timings()
        lay, exp_ms, rows_ms = tm.layout_ms, tm.expand_ms, tm.rows_ms
        eng.overlap(pars)
        cold = eng.timings().overlap_total_ms
        acc = timed_passes(eng, pars, steps, 1, sync)
        nnz = int(eng.get_B()[0][-1])
        r = roofline_of(acc, nnz, copy_gbps, traffic_key, expand_ms=exp_ms, layout=name)
        if name == "row_lists":                               # (the committed PMC summary of the same workload with the row lists, if there is one)
            try:
                tr = json.load(open(os.path.join(ROOT, "profiles", "r06_hbm_traffic.json"))).get(traffic_key + "_rl")
                if tr and abs(tr["algorithmic_bytes"] - r["algorithmic_bytes_per_step"]) < 1e-3 * tr["algorithmic_bytes"]:
                    r["traffic"] = tr["hbm_bytes_corrected"]
            except Exception:
                pass
        out[name] = {"layout_ms": lay, "expansion_ms": exp_ms, "rows_ms": rows_ms, "cold_pass_device_ms": cold, "cold_total_ms": lay + cold, "traffic": r.get("traffic"),
                     "warm_ms_per_step": acc["elapsed"] * 1e3 / steps, "numeric_kernel_ms": r["kernel_ms_per_step"], "frac": r["frac"],
                     "frac_incl_expansion": r["frac_incl_expansion"], "rowlist_bytes": int(eng.memory().rowlist_bytes)}
    eng.set_tuning("row_lists")
    out["default_chosen_by"] = "cold_total_ms (layout + first pass): HashSpGEMM is called once per run"
    return out


def phases_of(acc):
    s = acc["steps"]
    return {"symbolic+tiering": acc["sym"] / s, "row_kernels": acc["rows"] / s, "overflow_fold": acc["fold"] / s,
            "slot_order+placement": acc["comp"] / s}


def assemble_record(info, nnz):
    """assembly (tuples -> rows of B in MergeDuplicates slot order -> device layout B' / A'): SURVEY 8d bytes 8*T + 6*nnz(A)"""
    alg = 8.0 * info["ntuples"] + 6.0 * nnz
    ms = info["asm_ms"]
    ach = alg / (ms * 1e-3) / 1e9 if ms else 0.0
    return {"ms": ms, "rows_ms": info.get("rows_ms"), "layout_ms": info.get("layout_ms"), "tuples": int(info["ntuples"]),
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
                         "algorithmic_bytes": alg, "kernel": "k_asm_rows_* + radix sort of the entries by k-mer + k_layout_* (one call)"}}


def kcount_record(info, eng):
    """k-mer counting + dictionary + tuples (bella_hip_count_kmers): HBM roofline on 8 B per k-mer position in and out of the sort key
    array (emit + one read: 16 B / position) + 10 B per tuple written -- the bytes any counting-by-sorting of u64 words must move once"""
    npos = float(info.get("npositions") or 0)
    alg = 16.0 * npos + 10.0 * float(info["ntuples"])
    ms = info["kcount_ms"]
    ach = alg / (ms * 1e-3) / 1e9 if ms else 0.0
    runs = info.get("kcount_runs_ms")
    return {"ms": ms, "what": "the FIRST call on a fresh context (cold); warm_ms = a second call with the buffers in place", "warm_ms": runs[1] if runs else None,
            "runs_ms": runs, "positions": int(npos), "tuples": int(info["ntuples"]),
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "algorithmic_bytes": alg,
                         "kernel": "k_emit_codes + radix sort of the canonical words + run/dictionary/tuple passes (one call)"}}


# k_xdrop_slice, per anti-diagonal step of one wavefront (64 extensions), from the device assembly (tools/xdrop_isa.py ->
# profiles/r06_xdrop_isa.txt; tests/test_abi_cpu.py checks these constants against that listing): the instructions EVERY step
# executes, split by issue rate (profiles/r05_valu_rates.txt: ~4.15 cycles per SIMD for packed-16 / perm / min-max / shifts-left /
# compares, ~2.4 for add / logic / shift-right / mov), plus the sequence windows' checkpoint once per sixteen steps.  Rebase, clamp, the
# Phase-4 bookkeeping and the result stores are rare events and not counted: the figure is a LOWER bound of the issue time.
XDROP_VALU_HALF_RATE = 190
XDROP_VALU_FULL_RATE = 40
XDROP_VALU_CHECKPOINT = 5
XDROP_CYC_HALF, XDROP_CYC_FULL = 4.15, 2.4
GPU_SIMDS, GPU_CLOCK_HZ = 256 * 4, 2.4e9


def xdrop_record(eng, workload):
    """RunPairWiseAlignments on the candidate pairs the engine holds (one pass, outside the SpGEMM timing)"""
    from bella_amd import BellaPars
    apars = BellaPars()
    eng.overlap(apars)
    npass = eng.align_pairs(apars)
    xms = eng.timings().xdrop_ms
    al = eng.get_alignments()
    steps_tot = float(al["steps"].astype(np.float64).sum())
    rec = {"workload": workload % len(al), "pairs": int(len(al)), "passed": int(npass), "ms": xms,
           "pairs_per_s": len(al) / (xms * 1e-3) if xms else None, "antidiagonal_steps": steps_tot,
           "gcups": 31 * steps_tot / (xms * 1e-3) / 1e9 if xms else None, "flagged": int(al["flagged"].sum()),
           "bound": "VALU issue on a serial chain (one lane per extension, the band in VGPRs as packed i16)"}
    # in-run utilisation: the SIMD cycles the stage's anti-diagonal steps need at the measured issue rates if every wavefront ran 64 live
    # lanes, over the SIMD cycles the stage had.  1 - frac = idle lanes (an extension that ends idles its lane for the rest of its
    # 512-step slice), the rare regions, slice prologues / epilogues, launch gaps.
    if xms:
        insts = XDROP_VALU_HALF_RATE + XDROP_VALU_FULL_RATE + XDROP_VALU_CHECKPOINT
        cyc = (XDROP_VALU_HALF_RATE + XDROP_VALU_CHECKPOINT) * XDROP_CYC_HALF + XDROP_VALU_FULL_RATE * XDROP_CYC_FULL
        rec["valu_insts_per_wave_step"] = insts
        rec["valu_issue_cycles_per_wave_step"] = cyc
        rec["valu_issue_frac"] = (steps_tot / 64.0) * cyc / (GPU_SIMDS * GPU_CLOCK_HZ * xms * 1e-3)
        rec["valu_issue_source"] = ("profiles/r06_xdrop_isa.txt (instructions every step executes, from the ISA) x profiles/r05_valu_rates.txt (issue rates); "
                                    "steps and time measured in this run; SQ_INSTS_VALU of a profiled run: profiles/r06_xdrop_sq.txt")
    del al
    return rec


def dropin_call_record(rs, Bhost, nk, device):
    """The call the drop-in shim makes (bella_amd/host/bella_hip_shim.hpp: HashSpGEMM --skip-alignment), step by step through the C
    ABI on a FRESH context, wall-clock: host reads in, the reference's CSC of B in, ONE cold pass, records out, output file."""
    import tempfile
    from bella_amd import BellaPars, Engine
    from bella_amd.api import write_output, _ACGT
    pars = BellaPars(skipAlignment=True)
    asc = np.ascontiguousarray(_ACGT[rs.codes])
    offs = np.ascontiguousarray(rs.offsets, dtype=np.uint64)
    rec = {}
    t_all = time.perf_counter()
    t0 = time.perf_counter(); eng = Engine(device); rec["init_ms"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter(); eng.set_reads_raw(asc, offs, names=rs.names); rec["set_reads_ms"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter(); eng.set_B(17, nk, *Bhost); rec["set_B_ms"] = (time.perf_counter() - t0) * 1e3
    rec["set_B_device_layout_ms"] = eng.timings().layout_ms
    t0 = time.perf_counter(); npairs, flops = eng.overlap(pars); rec["cold_overlap_ms"] = (time.perf_counter() - t0) * 1e3
    rec["cold_overlap_device_ms"] = eng.timings().overlap_total_ms
    rec["spgemm_cold_device_ms"] = rec["set_B_device_layout_ms"] + rec["cold_overlap_device_ms"]     # layout + first pass: what the ONE call of a run pays on the device
    t0 = time.perf_counter(); pairs, _, colptr = eng.get_pairs(ext=False); rec["get_pairs_ms"] = (time.perf_counter() - t0) * 1e3
    with tempfile.TemporaryDirectory() as tmp:
        f = os.path.join(tmp, "out.out")
        open(f, "wb").close()
        st = write_output(f, pars, rs.names, rs.lengths, pairs)
        rec["writer_ms"] = st.seconds * 1e3
        rec["writer_format_ms"] = st.format_seconds * 1e3
        rec["writer_threads"] = int(st.threads)
        rec["output_bytes"] = int(st.bytes)
        rec["output_lines"] = int(st.lines)
        # one host thread on a bounded sample (<= 2M lines): what the multi-threaded writer is measured against
        ns = min(len(pairs), 2000000)
        open(f, "wb").close()
        s1 = write_output(f, pars, rs.names, rs.lengths, pairs[:ns], nthreads=1)
        rec["writer_1thread_ms_scaled"] = s1.seconds * 1e3 * (len(pairs) / max(ns, 1))
        rec["writer_speedup_vs_1thread"] = rec["writer_1thread_ms_scaled"] / rec["writer_ms"] if rec["writer_ms"] else None
    rec["total_ms"] = (time.perf_counter() - t_all) * 1e3 - s1.seconds * 1e3
    rec["pairs"] = int(npairs)
    rec["what"] = ("fresh context -> bella_hip_set_reads (ASCII bases from pageable host memory) -> bella_hip_set_B (the reference's CSC "
                   "arrays) -> bella_hip_overlap (first pass) -> bella_hip_get_pairs -> bella_hip_write_output; wall clock on the host")
    eng.close()
    return rec


def ingest_record(rs, device):
    """FASTQ file -> packed reads on the device (bella_hip_load_fastq) on a fresh context, the file in the page cache"""
    import tempfile
    from bella_amd import Engine
    from bella_testkit import synth
    with tempfile.TemporaryDirectory() as tmp:
        f = os.path.join(tmp, "reads.fastq")
        synth.write_fastq(f, rs)
        eng = Engine(device)
        eng.load_fastq(f)                                      # first call: pinned buffers are allocated, pages are mapped
        best = None
        for _ in range(2):
            t0 = time.perf_counter(); n, nb = eng.load_fastq(f); wall = (time.perf_counter() - t0) * 1e3
            st = eng.ingest_stats()
            if best is None or wall < best["ms"]:
                best = {"ms": wall, "index_ms": st["index_ms"], "upload_and_pack_ms": st["upload_ms"], "host_threads": int(st["threads"]),
                        "file_bytes": int(st["file_bytes"]), "reads": int(n), "bases": int(nb),
                        "file_gb_per_s": st["file_bytes"] / (wall * 1e-3) / 1e9}
        eng.close()
        best["what"] = ("bella_hip_load_fastq, warm: mmap + threaded line index, bases gathered into pinned 64 MB chunks under the previous "
                        "chunk's transfer, 2-bit pack on the device, names and lengths kept; wall clock on the host")
    return best


def e2e_on_set(rs, with_alignment):
    """the native command line on this read set's FASTQ (written to a temporary directory first; run when nothing else of the bench is
    using the host cores: the reference's OpenMP baseline takes all of them)"""
    import tempfile
    from bella_testkit import synth
    with tempfile.TemporaryDirectory() as tmp:
        f = os.path.join(tmp, "reads.fastq")
        synth.write_fastq(f, rs)
        return e2e_record(f, tmp, with_alignment)


def e2e_record(fastq, tmp, with_alignment):
    """bella_amd/bin/bella-hip (bella_amd/host/bella_hip_main.cpp) as a user runs it: FASTQ file in, output file out, a fresh process
    per run -- process start, context, reservation, ingest, k-mer counting, assembly, overlap, (alignment,) records to the host, the
    writer.  Wall clock of the process; the FASTQ is in the page cache (it was just written), the output goes to the same tmpfs / disk."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "bella_amd", "bin", "bella-hip")
    if not os.path.exists(exe):
        return {"error": "bella_amd/bin/bella-hip is not built"}
    with open(os.path.join(tmp, "in.txt"), "w") as fl:
        fl.write(fastq + "\n")
    rec = {"what": "wall clock of one `bella-hip -f in.txt -o out [--skip-alignment]` process on the FASTQ of this read set (k=17, defaults); "
                   "stages = the program's own log lines"}
    for key, flags in (("skip_alignment", ["--skip-alignment"]),) + ((("aligned", []),) if with_alignment else ()):
        t0 = time.perf_counter()
        p = subprocess.run([exe, "-f", "in.txt", "-o", "e2e_" + key] + flags, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        wall = time.perf_counter() - t0
        err = p.stderr.decode(errors="replace")
        nums = [ln.strip() for ln in p.stdout.decode(errors="replace").splitlines() if re.fullmatch(r"[0-9.eE+-]+", ln.strip())]
        if p.returncode != 0 or len(nums) < 4:
            rec[key] = {"error": "rc %d: %s" % (p.returncode, err[-300:])}
            continue
        grab = lambda name: (lambda m: float(m.group(1)) if m else None)(re.search(r"%s = ([0-9.]+) seconds" % name, err))
        out = os.path.join(tmp, "e2e_" + key + ".out")
        rec[key] = {"wall_s": wall, "nkmers": int(nums[0]), "nnzA": int(nums[1]), "pairs": int(nums[2]), "lines": int(nums[3]) if key == "aligned" else int(nums[2]),
                    "output_bytes": os.path.getsize(out) if os.path.exists(out) else 0,
                    "ingest_s": grab("fastqParsingTime"), "kcount_s": grab("KmerCountingTime"), "assemble_s": grab("SparseMatrixCreationTime"),
                    "output_s": grab("OutputtingTime"), "program_total_s": grab("TotalRuntime")}
        if os.path.exists(out):
            os.remove(out)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=0, help="override the read count of the headline workload (development)")
    ap.add_argument("--read-len", type=int, default=10000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-10k", "--no-100k", dest="no_10k", action="store_true", help="N=1: skip the config_10k sub-record (configs[1] / configs[2])")
    ap.add_argument("--no-xdrop", action="store_true", help="N=1: skip the X-drop records (configs[2]; configs[3]'s alignment stage)")
    ap.add_argument("--no-hifi", action="store_true", help="N=1: skip the config_hifi sub-record (configs[4]'s regime, 10k HiFi reads)")
    ap.add_argument("--no-dropin", action="store_true", help="N=1: skip the dropin_call records (the shim's call sequence, cold, wall clock)")
    ap.add_argument("--no-e2e", action="store_true", help="N=1: skip the e2e records (bella_amd/bin/bella-hip on the read set's FASTQ, wall clock)")
    ap.add_argument("--no-layout-ab", action="store_true", help="N=1: skip the layout A/B records (default layout vs row lists: layout + cold pass + warm step)")
    ap.add_argument("--debug-flags", type=int, default=0, help="extra bella_hip_set_debug bits (development A/B)")
    ap.add_argument("--layout-debug", type=int, default=0, help="bella_hip_set_debug bits in force while the operands are laid out (development A/B)")
    ap.add_argument("--tune", action="append", default=[], help="name=value for bella_hip_set_tuning on every context (development A/B), e.g. compact_b=1")
    ap.add_argument("--cpu-baseline-child", default=None)
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args()
    if a.cpu_baseline_child:
        return cpu_baseline_child(a.cpu_baseline_child, a.threads)

    import torch
    import torch.distributed as dist
    from bella_amd import BellaPars, Engine

    from bella_testkit import synth
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
    dev = "cuda:%d" % local

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def prepare(nreads, want_host_tuples, use_lib=False, rs=None):
        """reads -> reliable k-mer dictionary + tuples on the device (bella_hip_count_kmers: the reference's SplitCount + tuple
        loop) -> B (N = 1: whole; N > 1: this rank's row-block panel, then the all-gather) -> device layout.
        use_lib (N > 1): the library's own RCCL communicator does the counting exchange and the panel all-gather
        (bella_hip_count_kmers_dist, bella_hip_allgather_panels); otherwise every rank counts all reads and the panels travel
        through torch.distributed's all_gather."""
        t0 = time.time()
        if rs is None:
            rs = synth.make_reads_fast(nreads, read_len=a.read_len, coverage=30.0, err=0.15, seed=1)
        t1 = time.time()
        eng = Engine(local)
        # one slab from the driver up front (bella_hip_reserve): the first hipMalloc of the stages' multi-GB buffers costs tens of ms per
        # GB on this stack; a process that runs the pipeline pays that once, here, and reports it
        total_bases = int(rs.offsets[-1])
        want = int(min(0.55 * torch.cuda.mem_get_info(local)[0], max(6 << 30, 44 * total_bases / max(1, world if use_lib else 1))))
        reserve_ms = None
        if not os.environ.get("BELLA_BENCH_NO_RESERVE"):
            try:
                reserve_ms = eng.reserve(want)
            except Exception as e:
                log("[bench] rank %d: bella_hip_reserve(%d) failed (%r): the stages allocate for themselves" % (rank, want, e))
        if a.layout_debug:
            eng.set_debug(a.layout_debug)
        for tv in a.tune:
            tn, _, tval = tv.partition("=")
            eng.set_tuning(tn, int(tval))
        eng.set_reads(rs)                                   # reads are replicated (2 bit/base)
        t_setup = time.time()
        have_comm = False
        if world > 1:
            from bella_amd import dist as bd
            have_comm = bd.init_comm(eng, local, backend) if use_lib else False   # the library's own RCCL communicator (include/bella_hip.h)
            lo, npanel = bd.block_range(rank, world, nreads)
        def all_ok(flag):                                   # every rank takes the same path
            if world == 1:
                return bool(flag)
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(int(t.item()))

        counted = False
        info_fail = {}
        if have_comm:                                       # the dictionary is counted ACROSS the ranks, tuples for the own read block
            try:
                nk, nt, ndistinct = eng.count_kmers_dist(lo, npanel, 17, 2, 8)
                counted = True
            except Exception as e:                          # reported, then every rank counts all reads itself
                log("[bench] rank %d: bella_hip_count_kmers_dist failed (%r)" % (rank, e))
            counted = all_ok(counted)
            if not counted:                                 # one rank failed (or timed out): every rank leaves the library's communicator
                try:
                    eng.comm_destroy()
                except Exception:
                    pass
                have_comm = False
                info_fail["count"] = "bella_hip_count_kmers_dist failed on some rank"
        kc_runs = None
        if not counted:
            have_dist_count = False
            nk, nt, ndistinct = eng.count_kmers(17, 2, 8)
            kc_ms = eng.timings().kcount_ms
            if world == 1:
                # kcount_ms is the FIRST call on this fresh context (what a run pays); a second call (buffers in place) is recorded as warm
                eng.count_kmers(17, 2, 8)
                kc_runs = [kc_ms, eng.timings().kcount_ms]
        else:
            have_dist_count = True
            kc_ms = eng.timings().kcount_ms
        info = {"rs": rs, "nk": nk, "ntuples": nt, "npositions": int(np.maximum(rs.lengths.astype(np.int64) - 16, 0).sum()), "kcount_ms": kc_ms, "kcount_runs_ms": kc_runs, "reserve_ms": reserve_ms, "reserve_bytes": want if reserve_ms is not None else 0, "xchg_ms": None, "xchg_path": None, "have_comm": have_comm,
                "kcount_path": "bella_hip_count_kmers_dist (code space split over the ranks)" if have_dist_count else "bella_hip_count_kmers (every rank, all reads)"}
        info["tup"] = synth.Tuples(*eng.get_tuples(), nk) if want_host_tuples else None
        if rank == 0:
            log("[bench] reads %d (%.1f s), distinct k-mers %d, reliable %d, tuples %d (device: %.1f ms)"
                % (nreads, t1 - t0, ndistinct, nk, nt, info["kcount_ms"]))
        if world == 1:
            eng.assemble_counted()
            info["asm_ms"] = eng.timings().assemble_ms
            info["rows_ms"] = eng.timings().rows_ms
            info["layout_ms"] = eng.timings().layout_ms
        else:
            eng.assemble_counted_panel(lo, npanel)             # from the device-resident tuples of this rank's read block
            info["asm_ms"] = eng.timings().assemble_ms
            eng.set_partition(rank, world)                     # BEFORE the exchange: the layout it ends with holds B' for this rank's columns only
            sync()
            tx = time.perf_counter()
            xok = False
            if have_comm:
                try:
                    info["xchg_path"] = bd.exchange_panels(eng, local, backend, have_comm=True)
                    xok = True
                except Exception as e:
                    log("[bench] rank %d: bella_hip_allgather_panels failed (%r)" % (rank, e))
                xok = all_ok(xok)
                if not xok:
                    info_fail["exchange"] = "bella_hip_allgather_panels failed on some rank"
            if not xok:
                info["xchg_path"] = bd.exchange_panels(eng, local, backend, have_comm=False)
            sync()
            info["xchg_ms"] = (time.perf_counter() - tx) * 1e3               # exchange + device layout (wall, barrier to barrier)
            info["layout_ms"] = eng.timings().layout_ms
            info["asm_ms"] += eng.timings().assemble_ms
            mem = eng.memory()
            info["mem"] = {"layout_B_bytes": int(mem.layout_B_bytes), "layout_A_bytes": int(mem.layout_A_bytes), "matrix_bytes": int(mem.matrix_bytes),
                           "owned_nnz": int(mem.owned_nnz), "layout_shared": int(mem.layout_shared)}
        info["lib_failures"] = info_fail
        info["setup_wall_ms"] = (time.time() - t_setup) * 1e3                     # reads on the device -> operands laid out (count + assemble + exchange + layout)
        return eng, info

    # measured device-copy ceiling: 1 GiB device-to-device, read + write bytes per second
    copy_gbps = None
    if rank == 0:
        try:
            src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
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

    pars = BellaPars(skipAlignment=True)
    if world == 1:
        import threading

        def one_set(nreads, key, steps, warmup, want_cpu, full_cpu_in_background):
            """everything the line reports about one PacBio-shaped read set on one GPU: the timed SpGEMM step, roofline, the cold
            front end, the layout A/B, X-drop on all candidate pairs, the drop-in call, ingest, the CPU baseline"""
            eng, info = prepare(nreads, False)
            eng.set_debug(2 | a.debug_flags)                 # diagnostics array (pair_ext) off in the timed path
            acc = timed_passes(eng, pars, steps, warmup, sync)
            colptr, _, _ = eng.get_B()
            nnz = int(colptr[-1])
            elapsed = acc["elapsed"]
            rec = {"workload": "%d synthetic PacBio reads (%d b templates, 15%% err, 30x) k=17 SpGEMM-only (--skip-alignment); the step leaves the "
                               "pair records in HBM" % (nreads, a.read_len),
                   "steps": steps, "warmup": warmup, "ms_per_step": elapsed * 1e3 / steps, "value": acc["npairs"] / (elapsed / steps), "unit": "pairs/s",
                   "reads": nreads, "nkmers": info["nk"], "nnzA": nnz, "flops": int(acc["flops"]), "pairs": int(acc["npairs"]),
                   "roofline": roofline_of(acc, nnz, copy_gbps, key), "phases_ms_per_step": phases_of(acc),
                   "reserve_ms": info.get("reserve_ms"), "reserve_bytes": info.get("reserve_bytes"),
                   "kcount_ms": info["kcount_ms"], "assemble_ms": info["asm_ms"], "assemble": assemble_record(info, nnz), "kcount": kcount_record(info, eng)}
            cpu_thread, cpu_box = None, {}
            if want_cpu:
                # the reference's own HashSpGEMM on this box's host cores, on the WHOLE set: started now, in a child process, and
                # collected at the end of the record -- it runs under the GPU work below (X-drop, drop-in call, ingest)
                tk, tr, tp = eng.get_tuples()
                rs = info["rs"]

                def run_cb():
                    cpu_box["cb"] = run_cpu_baseline(rs.codes, rs.offsets, tk, tr, tp, info["nk"], "the whole workload (%d reads)" % nreads)
                if full_cpu_in_background:
                    cpu_thread = threading.Thread(target=run_cb)
                    cpu_thread.start()
                else:
                    run_cb()
            if not a.no_layout_ab:
                rec["layout_ab"] = layout_ab_record(eng, pars, info, copy_gbps, sync, 3 if nreads > 20000 else 5, key)
            if not a.no_xdrop:
                rec["xdrop"] = xdrop_record(eng, "X-drop (xdrop=7) on ALL %%d candidate pairs of the %d-read set" % nreads)
            Bhost = eng.get_B() if not a.no_dropin else None
            eng.close()
            if not a.no_dropin:
                rec["dropin_call"] = dropin_call_record(info["rs"], Bhost, info["nk"], local)
                rec["dropin_call"]["pairs_match_step"] = rec["dropin_call"]["pairs"] == int(acc["npairs"])
                rec["ingest"] = ingest_record(info["rs"], local)
            if want_cpu:
                if cpu_thread is not None:
                    cpu_thread.join()
                cb = cpu_box.get("cb") or {"value": None, "unit": "pairs/s", "cores": 0, "kind": "port", "sample": "failed", "pairs": None}
                cb["pairs_match_gpu"] = cb.pop("pairs", None) == int(acc["npairs"])
                rec["cpu_baseline"] = cb
            if not a.no_e2e:                                    # (after the CPU baseline: it runs on all host cores)
                rec["e2e"] = e2e_on_set(info["rs"], not a.no_xdrop)
            del eng, info, Bhost
            return rec

        nreads = a.reads or BIG_READS
        # HEADLINE: configs[3]'s read set (100k reads) on ONE GPU, SpGEMM-only -- the set BASELINE.json quotes the 40 % HBM-roofline
        # target on; configs[1] (10k reads) and configs[2] (its X-drop stage) travel as config_10k
        big = one_set(nreads, "100k" if nreads == BIG_READS else "10k" if nreads == 10000 else "none", a.steps, a.warmup,
                      not a.no_cpu_baseline, True)
        out = {
            "metric": "candidate overlap pairs/sec", "value": big["value"], "unit": "pairs/s", "n_gpus": 1,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": big["ms_per_step"], "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u16/u32 integer", "data": "synthetic",
            "headline_set": "configs[3]'s 100k-read set on one GPU since round 5 (rounds 1-4 quoted the 10k set: it travels as config_10k)",
            "config": {"workload": ("configs[3]'s read set on one GPU (the set the roofline target is quoted on): " if nreads == BIG_READS else "") + big["workload"],
                       "reads": nreads, "nkmers": big["nkmers"], "nnzA": big["nnzA"], "flops": big["flops"], "pairs": big["pairs"],
                       "partition": "all columns on one GPU"},
            "roofline": big["roofline"], "phases_ms_per_step": big["phases_ms_per_step"],
            "kcount_ms": big["kcount_ms"], "assemble_ms": big["assemble_ms"], "panel_allgather_ms": None,
        }
        for k2 in ("assemble", "kcount", "reserve_ms", "reserve_bytes", "layout_ab", "xdrop", "dropin_call", "ingest", "e2e", "cpu_baseline"):
            if k2 in big:
                out[k2] = big[k2]
        if not a.no_10k and not a.reads:
            out["config_10k"] = one_set(10000, "10k", 20, 3, not a.no_cpu_baseline, False)
            out["config_10k"]["workload"] = "configs[1] (+ configs[2] in xdrop): " + out["config_10k"]["workload"]
        if not a.no_hifi and not a.reads:
            # configs[4]'s regime at single-GPU scale: 10k HiFi reads (15 kb, 0.5 % error, 30x), syncmer selection (-s), the reference's
            # default bound -u 8 and the raised -u 40: hundreds of products per pair, columns above the LDS tiers
            rs = synth.make_reads_fast(10000, read_len=15000, coverage=30.0, err=0.005, seed=2, mix=(1 / 3, 1 / 3, 1 / 3))
            hp = BellaPars(skipAlignment=True, errorRate=0.005)
            hifi = {"workload": "configs[4]'s regime on one GPU: 10000 synthetic HiFi reads (15000 b, 0.5% err, 30x) k=17 syncmer mode (-s), SpGEMM-only"}
            for upper in (8, 40):
                eng = Engine(local)
                eng.set_reads(rs)
                nk, nt, _ = eng.count_kmers(17, 2, upper, syncmer=True)
                kc = eng.timings().kcount_ms
                eng.assemble_counted()
                asm_ms = eng.timings().assemble_ms
                exp_ms = eng.timings().expand_ms                     # long-list inputs get the row lists at layout time: their expansion counts
                eng.set_debug(2 | a.debug_flags)
                acc = timed_passes(eng, hp, 5, 2, sync)
                colptr, _, _ = eng.get_B()
                nnz = int(colptr[-1])
                hifi["u%d" % upper] = {"upper": upper, "nkmers": nk, "nnzA": nnz, "flops": int(acc["flops"]), "pairs": int(acc["npairs"]),
                                       "ms_per_step": acc["elapsed"] * 1e3 / 5, "products_per_s": acc["flops"] / (acc["elapsed"] / 5),
                                       "pairs_per_s": acc["npairs"] / (acc["elapsed"] / 5), "roofline": roofline_of(acc, nnz, copy_gbps, "hifi_u%d" % upper, expand_ms=exp_ms, layout="row_lists" if exp_ms else "default"),
                                       "expansion_ms_at_layout": exp_ms,
                                       "phases_ms_per_step": phases_of(acc), "kcount_ms": kc, "assemble_ms": asm_ms}
                eng.close()
            out["config_hifi"] = hifi
        print(json.dumps(out))
        return

    # ---- N > 1: strong scaling on the fixed 100k-read set (configs[3]: SpGEMM + alignment) ----
    nreads = a.reads or BIG_READS
    tdev = dev if backend == "nccl" else "cpu"

    def measure(eng, info):
        """the timed SpGEMM step on this rank's columns; max over ranks of the times, sums of the counts"""
        eng.set_partition(rank, n_gpus)
        eng.set_debug(2 | a.debug_flags)
        acc = timed_passes(eng, pars, a.steps, a.warmup, sync)
        mem = info.get("mem") or {}
        tt = torch.tensor([acc["elapsed"], acc["rows"] + acc["fold"] + acc["comp"], info["kcount_ms"], info["asm_ms"], info["xchg_ms"] or 0.0,
                           float(acc["npairs"]), float(acc["flops"]), info["setup_wall_ms"], info.get("layout_ms") or 0.0,
                           float(mem.get("layout_B_bytes", 0)), float(mem.get("owned_nnz", 0))], dtype=torch.float64, device=tdev)
        mx = tt.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tt.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        return {"acc": acc, "elapsed": float(mx[0]), "kernel_ms": float(mx[1]) / a.steps, "kcount_ms_max": float(mx[2]), "assemble_ms_max": float(mx[3]),
                "xchg_ms_max": float(mx[4]), "pairs": float(sm[5]), "flops": float(sm[6]), "setup_ms_max": float(mx[7]), "layout_ms_max": float(mx[8]),
                "layout_B_bytes_max": float(mx[9]), "layout_B_bytes_sum": float(sm[9]), "owned_nnz_max": float(mx[10]), "owned_nnz_sum": float(sm[10])}

    # The set-up runs ONCE, through the library's own RCCL communicator when there is one (C ABI: bella_hip_comm_init with its self-test,
    # bella_hip_count_kmers_dist, bella_hip_allgather_panels -- every wait has a deadline); any step that fails on any rank sends ALL
    # ranks to the torch.distributed path for that step (prepare: all_ok), and the line says which path ran.
    use_lib = (backend == "nccl" or bool(os.environ.get("BELLA_BENCH_FORCE_LIB_PROBE"))) and not os.environ.get("BELLA_BENCH_NO_LIBCOMM")
    # Process-level watchdog over set-up + measurement: the library's waits have deadlines, ncclCommInitRank / a torch
    # collective whose peer never arrives do not.  A rank that is still here after BELLA_BENCH_WATCHDOG_S (default 900 s) says so on
    # stderr, rank 0 prints a line the driver can parse ("value": null, "error"), and the process leaves.
    import threading
    wd_done = threading.Event()

    def watchdog():
        limit = float(os.environ.get("BELLA_BENCH_WATCHDOG_S", "900"))
        if wd_done.wait(limit):
            return
        log("[bench] rank %d: no result after %.0f s (a peer did not arrive?); giving up" % (rank, limit))
        if rank == 0:
            print(json.dumps({"metric": "candidate overlap pairs/sec", "value": None, "unit": "pairs/s", "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup,
                              "ms_per_step": None, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u16/u32 integer",
                              "data": "synthetic", "config": {"workload": "configs[3]'s read set over %d GPUs" % n_gpus},
                              "error": "watchdog: the ranks did not finish within %.0f s" % limit}), flush=True)
        os._exit(3)
    threading.Thread(target=watchdog, daemon=True).start()
    eng, info = prepare(nreads, False, use_lib=use_lib)
    mA = measure(eng, info)
    xd = None
    if not a.no_xdrop:
        # configs[3]'s alignment stage: every rank aligns the candidate pairs of ITS columns (no exchange: reads are replicated)
        apars = BellaPars()
        eng.set_debug(a.debug_flags)
        np_loc, _ = eng.overlap(apars)
        npass = eng.align_pairs(apars)
        xt = torch.tensor([eng.timings().xdrop_ms, float(np_loc), float(npass)], dtype=torch.float64, device=tdev)
        xm = xt.clone(); dist.all_reduce(xm, op=dist.ReduceOp.MAX)
        xs = xt.clone(); dist.all_reduce(xs, op=dist.ReduceOp.SUM)
        xd = {"workload": "configs[3]'s alignment stage: X-drop (xdrop=7), every rank on the pairs of its own columns", "ms_max_over_ranks": float(xm[0]),
              "pairs": int(xs[1]), "passed": int(xs[2]), "pairs_per_s": float(xs[1]) / (float(xm[0]) * 1e-3) if float(xm[0]) else None,
              "largest_rank_share": float(xm[1]) / max(float(xs[1]), 1.0)}
        eng.set_debug(2 | a.debug_flags)
    single = None
    colptr, _, _ = eng.get_B()
    nnz = int(colptr[-1])
    if rank == 0:
        # the same workload on ONE GPU, set-up included (a fresh context on rank 0's device): the denominator of the speed-ups
        e1 = Engine(local)
        if info.get("reserve_ms") is not None:               # like the ranks: the slab is reserved outside the timed set-up
            try:
                e1.reserve(int(min(0.55 * torch.cuda.mem_get_info(local)[0], max(6 << 30, 44 * int(info["rs"].offsets[-1])))))
            except Exception:
                pass
        t0 = time.perf_counter()
        e1.set_reads(info["rs"])
        t1 = time.perf_counter()
        e1.count_kmers(17, 2, 8)
        e1.assemble_counted()
        setup1 = (time.perf_counter() - t1) * 1e3
        e1.set_debug(2 | a.debug_flags)
        acc1 = timed_passes(e1, pars, 3, 1, torch.cuda.synchronize)
        single = {"ms_per_step": acc1["elapsed"] * 1e3 / 3, "value": acc1["npairs"] / (acc1["elapsed"] / 3), "pairs": int(acc1["npairs"]),
                  "setup_ms": setup1, "kcount_ms": e1.timings().kcount_ms, "assemble_ms": e1.timings().assemble_ms, "xdrop_ms": None}
        if not a.no_xdrop:
            e1.set_debug(a.debug_flags)
            e1.overlap(BellaPars())
            e1.align_pairs(BellaPars())
            single["xdrop_ms"] = e1.timings().xdrop_ms
        e1.close()
    dist.barrier()

    def line(m, setup):
        elapsed = m["elapsed"]
        return {
            "metric": "candidate overlap pairs/sec", "value": m["pairs"] / (elapsed / a.steps), "unit": "pairs/s", "n_gpus": n_gpus, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": elapsed * 1e3 / a.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u16/u32 integer", "data": "synthetic",
            "config": {"workload": "configs[3]: %d synthetic PacBio reads (%d b templates, 15%% err, 30x) k=17, row-block panels + one all-gather, "
                                   "SpGEMM step on %d GPUs (fixed set: strong scaling); alignment stage reported in xdrop" % (nreads, a.read_len, n_gpus),
                       "reads": nreads, "nkmers": info["nk"], "nnzA": nnz, "flops": int(m["flops"]), "pairs": int(m["pairs"]),
                       "partition": "columns i %% %d == rank" % n_gpus, "setup": setup},
            "roofline": roofline_of(m["acc"], nnz / n_gpus, copy_gbps, "none"),
            "phases_ms_per_step": phases_of(m["acc"]),
            "kcount_ms_max": m["kcount_ms_max"], "assemble_ms_max": m["assemble_ms_max"], "panel_allgather_ms": m["xchg_ms_max"],
            # the set-up of the step (reads resident -> operands laid out: count + assemble + exchange + device layout), wall clock, max over ranks
            "setup_ms_max_over_ranks": m["setup_ms_max"], "layout_ms_max_over_ranks": m["layout_ms_max"],
            "per_rank_layout": {"B_bytes_max": m["layout_B_bytes_max"], "B_bytes_sum": m["layout_B_bytes_sum"], "owned_nnz_max": m["owned_nnz_max"],
                                "owned_nnz_sum": m["owned_nnz_sum"], "nnz": nnz,
                                "formation": ("shared: every rank sorts its N-th of the k-mer id space, slices of A' and B' entries exchanged (BELLA_TUNE_DIST_LAYOUT)"
                                              if (info.get("mem") or {}).get("layout_shared") else "replicated: every rank sorts all entries"),
                                "note": "B' (10 B per entry) exists for the rank's own columns only; A' (8 B per entry) and the exchanged matrix (6 B) are whole on every rank"},
            "xdrop": xd,
            "single_gpu_same_workload": single,
            "speedup_vs_single_gpu": (single["ms_per_step"] / (elapsed * 1e3 / a.steps)) if single else None,
            # the figure that counts for a run: set-up + one step + the alignment stage, N GPUs against one
            "speedup_setup_step_xdrop": ((single["setup_ms"] + single["ms_per_step"] + (single["xdrop_ms"] or 0.0)) /
                                         (m["setup_ms_max"] + elapsed * 1e3 / a.steps + ((xd or {}).get("ms_max_over_ranks") or 0.0))) if single else None,
            "pairs_match_single_gpu": (single["pairs"] == int(m["pairs"])) if single else None,
        }

    out = line(mA, {"kcount": info["kcount_path"], "panel_allgather": info["xchg_path"]})
    if not use_lib:
        out["library_rccl_path"] = {"status": "not run (%s backend)" % backend}
    elif not info["have_comm"]:
        out["library_rccl_path"] = {"status": "communicator unavailable or failed its self-test: torch.distributed carried the set-up", "failures": info.get("lib_failures")}
    elif info.get("lib_failures"):
        out["library_rccl_path"] = {"status": "partly: " + "; ".join(info["lib_failures"].values()) + " -> torch.distributed for that step", "failures": info["lib_failures"]}
    else:
        out["library_rccl_path"] = {"status": "ok: this line"}
    if rank == 0:
        wd_done.set()
        print(json.dumps(out), flush=True)
    eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
