// bella_hip_shim.hpp -- the reference-side binding: BELLA's own call site, MI355X underneath.
//
// Include this header in the reference's src/main.cpp right AFTER its `#include "../include/align.hpp"` (main.cpp:55) and
// link libbella_hip.so.  It declares an overload of HashSpGEMM that is more specialised than the reference's template
// (include/overlap.hpp:650-652: IT=uint32_t [KMERINDEX, main.cpp:60], NT=unsigned short, FT=spmatPtr_), so the unchanged
// call at main.cpp:498-525 resolves to it.  Argument meaning, ownership and observable behaviour are the reference's:
//   A (spmat), B (transpmat): caller-owned CSC, read only.  Only B is read: A = B.Transpose() is rebuilt on the device with
//       ascending read ids per k-mer, i.e. the reference's own layout at OMP_NUM_THREADS=1 (transpose.h:26-50).
//   multop / addop: ignored -- the position-binning semiring (chain.hpp:74-150) runs inside the HIP kernels.
//   reads: caller-owned readVector_ (common.h:98-109), must be upper-case ACGT (align.hpp:40-55 asserts the same).
//   filename: the output file (already remove()d by main.cpp:130); BELLA or PAF lines, or 6-column --skip-alignment lines.
//   stdout: nnz(C) (overlap.hpp:686) and, when aligning, the number of lines written (overlap.hpp:771).
//   errors: the reference returns void and prints; so does this (message on stderr, then abort(), as CSC.cpp:269 does).
// The Kmer length is bpars.kmerSize (Kmer::set_k(bpars.kmerSize), main.cpp:183).
//   bpars.numGPU (-g): as the reference's GPU build fans batches over the devices inside the call (loganGPU/functions.cuh:441-443,
//       498-637; align.hpp:226-229), the call drives min(numGPU, devices) contexts from one host thread each: the host already
//       holds all of B, so every context receives it, takes the output columns i % N == g and the results are merged back into
//       the reference's column order (the multi-process path exchanges row-block panels with RCCL instead: bella_hip.h).
//   bpars.totalMemory / userDefMem (-m): stage count and stage boundaries by the reference's own formula (estimateMemory,
//       overlap.hpp:365-404; stages :682-710): ceil(1.5 * nnz(C) * (sizeof(spmatPtr_) + sizeof(uint32_t)) / free_memory), boundaries by
//       upper_bound on colptrC.  With more than one stage the output is formed stage by stage (each pass holds its own columns
//       only) and APPENDED -- the reference overwrites the file from offset 0 in every stage (overlap.hpp:613-636, a defect);
//       the file written here is the single-stage one.  totalMemory shapes the OUTPUT stages only: the planning pass runs the SpGEMM
//       over all columns on the device (device memory, not -m, bounds it: 24 bytes per product in flight) to obtain colptrC, and
//       with more than one stage every stage is then computed again -- about twice the device work of a single-stage call.
// Also here, in namespace bella_hip (the reference defines functions of the same names and signatures, so these cannot be
// overloads): bella_hip::xavierAlign -- include/align.hpp:152, same arguments, same xavierResult -- and bella_hip::alignXavier,
// the batched form shaped like alignLogan (include/align.hpp:210-211), both forwarding to bella_hip_xdrop_batch.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <iostream>
#include <string>
#include <vector>

#include "bella_hip.h"

namespace bella_hip_detail {
inline void check(bella_ctx* c, int rc, const char* what) {
    if (rc == 0) return;
    std::cerr << "bella_hip: " << what << " failed: " << (c ? bella_hip_last_error(c) : bella_hip_strerror(rc)) << " (" << rc << ")"
              << std::endl;
    std::abort();
}
}  // namespace bella_hip_detail

namespace bella_hip_detail {
// one context = one GPU: B in, then per stage [lo, hi): overlap (+ alignment) and the records of ITS columns
struct Worker {
    bella_ctx* ctx = nullptr;
    std::vector<uint64_t> colptr;            // colptrC of the last pass (nreads + 1)
    std::vector<bella_pair> pairs;
    std::vector<bella_aln> alns;
    uint64_t nnzc = 0;
};
// the reference's printLog (include/common/common.h:40-44): "INFO:\tfile(line)\tname = value" on stderr
#define BELLA_HIP_LOG(var) do { std::cerr << "INFO:\t" << "bella_hip_shim.hpp" << "(" << __LINE__ << ")\t" << #var << " = " << (var) << std::endl; } while (0)
}  // namespace bella_hip_detail

template <typename MultiplyOperation, typename AddOperation>
void HashSpGEMM(const CSC<uint32_t, unsigned short>& A, const CSC<uint32_t, unsigned short>& B, MultiplyOperation, AddOperation,
                const readVector_& reads, spmatPtr_& getvaluetype, char* filename, const BELLApars& bpars, const double& ratiophi) {
    using namespace bella_hip_detail;
    (void)A; (void)getvaluetype; (void)ratiophi;   // ratiophi = slope(bpars.errorRate) is recomputed from bpars
    const uint32_t nreads = (uint32_t)reads.size();
    const int ndev = bella_hip_device_count();
    if (ndev <= 0) check(nullptr, BELLA_ERR_NO_DEVICE, "bella_hip_device_count");
    // -g: one context per requested GPU.  BELLA_HIP_SHIM_OVERSUBSCRIBE=1 (tests on a one-GPU box) lets contexts share a device.
    int N = bpars.numGPU > 1 ? (int)bpars.numGPU : 1;
    if (N > ndev && !std::getenv("BELLA_HIP_SHIM_OVERSUBSCRIBE")) N = ndev;
    std::vector<uint64_t> offs(nreads + 1, 0);
    for (uint32_t r = 0; r < nreads; ++r) offs[r + 1] = offs[r] + reads[r].seq.size();
    std::string flat;
    flat.reserve(offs[nreads]);
    for (uint32_t r = 0; r < nreads; ++r) flat += reads[r].seq;
    bella_params p;
    p.kmer_size = bpars.kmerSize;
    p.bin_size = bpars.binSize;
    p.xdrop = bpars.xDrop;
    p.skip_alignment = bpars.skipAlignment;
    p.error_rate = bpars.errorRate;
    p.delta_chernoff = bpars.deltaChernoff;

    std::vector<Worker> W((size_t)N);
    auto on_all = [&](const std::function<void(int)>& fn) {                  // one host thread per context
        if (N == 1) { fn(0); return; }
        std::vector<std::thread> th;
        for (int g = 0; g < N; ++g) th.emplace_back(fn, g);
        for (auto& t : th) t.join();
    };
    on_all([&](int g) {
        Worker& w = W[(size_t)g];
        check(nullptr, bella_hip_init(g % ndev, &w.ctx), "bella_hip_init");
        check(w.ctx, bella_hip_set_reads(w.ctx, (const uint8_t*)flat.data(), offs.data(), nreads), "bella_hip_set_reads");
        check(w.ctx, bella_hip_set_B(w.ctx, bpars.kmerSize, (uint32_t)B.rows, B.colptr, B.rowids, B.values), "bella_hip_set_B");
        check(w.ctx, bella_hip_set_partition(w.ctx, (uint32_t)g, (uint32_t)N), "bella_hip_set_partition");
    });
    std::string().swap(flat);
    auto do_overlap = [&](uint32_t lo, uint32_t hi) {                        // columns [lo, hi) on every context; colptrC only
        on_all([&](int g) {
            Worker& w = W[(size_t)g];
            check(w.ctx, bella_hip_set_column_range(w.ctx, lo, hi - lo), "bella_hip_set_column_range");
            uint64_t flops = 0;
            check(w.ctx, bella_hip_overlap(w.ctx, &p, &w.nnzc, &flops), "bella_hip_overlap");
            w.colptr.assign((size_t)nreads + 1, 0);
            check(w.ctx, bella_hip_get_pairs(w.ctx, nullptr, nullptr, w.colptr.data()), "bella_hip_get_pairs");
        });
    };
    auto fetch = [&]() {                                                     // the records of the last pass (+ their alignments)
        on_all([&](int g) {
            Worker& w = W[(size_t)g];
            w.pairs.resize(w.nnzc);
            check(w.ctx, bella_hip_get_pairs(w.ctx, w.pairs.data(), nullptr, nullptr), "bella_hip_get_pairs");
            if (!bpars.skipAlignment) {
                uint64_t npass = 0;
                check(w.ctx, bella_hip_align_pairs(w.ctx, &p, &npass), "bella_hip_align_pairs");
                w.alns.resize(w.nnzc);
                if (w.nnzc) check(w.ctx, bella_hip_get_alignments(w.ctx, w.alns.data()), "bella_hip_get_alignments");
            }
        });
    };
    // first pass over all columns: nnz(C) and colptrC (what estimateNNZ_Hash + prefixsum give the reference, overlap.hpp:674-679)
    const double free_memory = bpars.totalMemory * 1024 * 1024;               // estimateMemory, overlap.hpp:365-404 (no LINUX/OSX define)
    do_overlap(0, nreads);
    uint64_t nnzc = 0;
    std::vector<uint64_t> colptrC((size_t)nreads + 1, 0);                    // merged over the contexts: column i lives on context i % N
    for (uint32_t i = 0; i < nreads; ++i) {
        const Worker& w = W[(size_t)(i % (uint32_t)N)];
        colptrC[i + 1] = colptrC[i] + (w.colptr[i + 1] - w.colptr[i]);
    }
    nnzc = colptrC[nreads];
    std::cout << nnzc << std::endl;                                           // overlap.hpp:686
    const double safety_net = 1.5;                                            // overlap.hpp:92
    const uint64_t required_memory = (uint64_t)(safety_net * nnzc * (sizeof(spmatPtr_) + sizeof(uint32_t)));
    int stages = (int)std::ceil((double)required_memory / free_memory);       // overlap.hpp:683
    if (stages < 1) stages = 1;
    const uint64_t nnzcperstage = (uint64_t)(free_memory / (safety_net * (sizeof(spmatPtr_) + sizeof(uint32_t))));
    std::vector<uint32_t> colStart((size_t)stages + 1, 0);
    for (int i = 1; i < stages; ++i) {                                        // overlap.hpp:704-710
        auto upper = std::upper_bound(colptrC.begin(), colptrC.end(), (uint64_t)i * nnzcperstage);
        colStart[(size_t)i] = (uint32_t)(upper - colptrC.begin() - 1);
    }
    colStart[(size_t)stages] = nreads;

    // names and lengths once; the writer (bella_hip_write_output: per-thread buffers, offset writes, overlap.hpp:603-642) appends
    std::vector<const char*> names(nreads);
    std::vector<uint32_t> lens(nreads);
    for (uint32_t r = 0; r < nreads; ++r) { names[r] = reads[r].nametag.c_str(); lens[r] = (uint32_t)reads[r].seq.length(); }
    for (int b = 0; b < stages; ++b) {
        const uint32_t lo = colStart[(size_t)b], hi = colStart[(size_t)b + 1];
        const auto t_stage = std::chrono::steady_clock::now();
        if (stages > 1) do_overlap(lo, hi);                                   // a single stage reuses the first pass
        fetch();
        const double aligntime = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_stage).count();
        // the stage's records in the reference's column order (N contexts: column i lives on context i % N)
        const bella_pair* pp = W[0].pairs.data();
        const bella_aln* aa = W[0].alns.data();
        uint64_t np = W[0].nnzc;
        std::vector<bella_pair> mp;
        std::vector<bella_aln> ma;
        if (N > 1) {
            for (uint32_t i = lo; i < hi; ++i) {
                const Worker& w = W[(size_t)(i % (uint32_t)N)];
                mp.insert(mp.end(), w.pairs.begin() + (std::ptrdiff_t)w.colptr[i], w.pairs.begin() + (std::ptrdiff_t)w.colptr[i + 1]);
                if (!bpars.skipAlignment) ma.insert(ma.end(), w.alns.begin() + (std::ptrdiff_t)w.colptr[i], w.alns.begin() + (std::ptrdiff_t)w.colptr[i + 1]);
            }
            pp = mp.data(); aa = ma.data(); np = mp.size();
        }
        bella_write_stats ws;
        const int wrc = bella_hip_write_output(filename, &p, bpars.outputPaf ? 1 : 0, nreads, names.data(), lens.data(), pp, bpars.skipAlignment ? nullptr : aa, np, 0, &ws);
        if (wrc) check(nullptr, wrc, "bella_hip_write_output");
        const std::string ColumnsRange = "[" + std::to_string(lo) + " - " + std::to_string(hi) + "]";
        BELLA_HIP_LOG(ColumnsRange);
        if (!bpars.skipAlignment) {                                           // the per-stage statistics of overlap.hpp:750-777
            const std::string AlignmentTime = std::to_string(aligntime) + " seconds";
            BELLA_HIP_LOG(AlignmentTime);
            const std::string AlignmentRate = std::to_string((long long)((double)ws.aligned_bases / aligntime)) + " bases/second";
            BELLA_HIP_LOG(AlignmentRate);
            const std::string AverageReadLength = std::to_string(ws.aligned_pairs ? (long long)((double)ws.total_read_len / (2.0 * (double)ws.aligned_pairs)) : 0LL);
            BELLA_HIP_LOG(AverageReadLength);
            const std::string PairsAligned = std::to_string(ws.aligned_pairs);
            BELLA_HIP_LOG(PairsAligned);
            std::cout << ws.lines << std::endl;                               // overlap.hpp:771 (per stage)
            const std::string AverageLengthSuccessfulAlignment = std::to_string(ws.lines ? (long long)((double)ws.bases_passed / (double)ws.lines) : 0LL) + " bps";
            BELLA_HIP_LOG(AverageLengthSuccessfulAlignment);
            const uint64_t nfail = ws.aligned_pairs - ws.lines;
            const std::string AverageLengthFailedAlignment = std::to_string(nfail ? (long long)((double)ws.bases_failed / (double)nfail) : 0LL) + " bps";
            BELLA_HIP_LOG(AverageLengthFailedAlignment);
        }
        const uint64_t LinesOutputted = ws.lines;
        BELLA_HIP_LOG(LinesOutputted);
        const std::string OutputtingTime = std::to_string(ws.seconds) + " seconds";
        BELLA_HIP_LOG(OutputtingTime);
    }
    for (auto& w : W) bella_hip_destroy(w.ctx);
}

// ---- the align.hpp call surface ---------------------------------------------------------------------------------------------
namespace bella_hip {

namespace detail {
struct ThreadCtx {                            // xavierAlign is called concurrently from the reference's OpenMP pair loop (overlap.hpp:565)
    bella_ctx* ctx = nullptr;
    ~ThreadCtx() { if (ctx) bella_hip_destroy(ctx); }
};
inline bella_ctx* thread_context() {
    static thread_local ThreadCtx t;
    if (!t.ctx) bella_hip_detail::check(nullptr, bella_hip_init(0, &t.ctx), "bella_hip_init");
    return t.ctx;
}
}  // namespace detail

// alignLogan-shaped batch (include/align.hpp:210-211): target[n] ("row", read H), query[n] ("col", read V) and seeds[n] in
// (beginPositionH/V = the seed k-mer's start on target / query, as SeedX(i, j, kmerSize) sets them, align.hpp:166);
// out[n] = what xavierAlign(target[n], query[n], target[n].size(), i, j, bpars.xDrop, bpars.kmerSize) returns, index-aligned.
inline void alignXavier(const std::vector<std::string>& target, const std::vector<std::string>& query, const std::vector<SeedX>& seeds,
                        const BELLApars& bpars, std::vector<xavierResult>& out) {
    using bella_hip_detail::check;
    const size_t n = seeds.size();
    out.resize(n);
    if (!n) return;
    bella_ctx* ctx = detail::thread_context();
    std::vector<uint64_t> offs(2 * n + 1, 0);
    std::string flat;
    for (size_t t = 0; t < n; ++t) {
        offs[2 * t + 1] = offs[2 * t] + target[t].size();
        offs[2 * t + 2] = offs[2 * t + 1] + query[t].size();
        flat += target[t];
        flat += query[t];
    }
    check(ctx, bella_hip_set_reads(ctx, (const uint8_t*)flat.data(), offs.data(), (uint32_t)(2 * n)), "bella_hip_set_reads");
    std::vector<bella_seed> sd(n);
    for (size_t t = 0; t < n; ++t) {
        sd[t].rid = (uint32_t)(2 * t);
        sd[t].cid = (uint32_t)(2 * t + 1);
        sd[t].seedH = (uint16_t)seeds[t].beginPositionH;
        sd[t].seedV = (uint16_t)seeds[t].beginPositionV;
    }
    bella_params p;
    p.kmer_size = bpars.kmerSize;
    p.bin_size = bpars.binSize;
    p.xdrop = bpars.xDrop;
    p.skip_alignment = 0;
    p.error_rate = bpars.errorRate;
    p.delta_chernoff = bpars.deltaChernoff;
    std::vector<bella_aln> al(n);
    check(ctx, bella_hip_xdrop_batch(ctx, sd.data(), n, &p, al.data()), "bella_hip_xdrop_batch");
    for (size_t t = 0; t < n; ++t) {
        out[t].score = al[t].score;
        out[t].strand = al[t].strand ? "c" : "n";
        out[t].seed.beginPositionH = al[t].begH;
        out[t].seed.beginPositionV = al[t].begV;
        out[t].seed.endPositionH = al[t].endH;
        out[t].seed.endPositionV = al[t].endV;
    }
}

// include/align.hpp:152.  One pair per call: both reads are uploaded and one extension pair is launched every time (a thread-local
// context keeps it safe inside the reference's OpenMP pair loop, overlap.hpp:565, not fast): a caller with many pairs uses alignXavier
// above, or the HashSpGEMM overload, which aligns all candidate pairs of a stage in one batch.
inline xavierResult xavierAlign(const std::string& row, const std::string& col, int rowLen, int i, int j, int xDrop, int kmerSize) {
    (void)rowLen;                                                              // == row.size() at the reference's call site (overlap.hpp:565)
    BELLApars bp;
    bp.kmerSize = (unsigned short)kmerSize;
    bp.xDrop = (unsigned short)xDrop;
    bp.errorRate = 0.15;
    std::vector<xavierResult> out;
    alignXavier({row}, {col}, {SeedX(i, j, kmerSize)}, bp, out);
    return out[0];
}

}  // namespace bella_hip
