// bella_hip_driver.hpp -- what a host program does AROUND the C ABI once the operands are on the device: the reference's stage plan,
// the passes, the alignment, the output file and the stdout protocol of HashSpGEMM (include/overlap.hpp:650-789).  Shared by the two
// host programs of this repository: the reference-side shim (bella_hip_shim.hpp: BELLA's own main.cpp, its k-mer counter and CSC
// constructor on the host) and the native command line (bella_hip_main.cpp: FASTQ to output file on the device).  No reference types,
// no reference code: plain C++ over include/bella_hip.h.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include "bella_hip.h"

namespace bella_hip_detail {

inline void check(bella_ctx* c, int rc, const char* what) {
    if (rc == 0) return;
    std::cerr << "bella_hip: " << what << " failed: " << (c ? bella_hip_last_error(c) : bella_hip_strerror(rc)) << " (" << rc << ")"
              << std::endl;
    std::abort();                                    // (the reference returns void and prints; CSC.cpp:269 aborts the same way)
}

// a host array the library fills: no value-initialisation (std::vector::resize writes 2.4 GB of zeros in front of the copy that
// overwrites them: 0.25 s of a 100k-read run with alignment)
template <class T>
struct RawBuf {
    T* p = nullptr;
    size_t n = 0, cap = 0;
    RawBuf() = default;
    RawBuf(const RawBuf&) = delete;
    RawBuf& operator=(const RawBuf&) = delete;
    RawBuf(RawBuf&& o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = o.cap = 0; }
    ~RawBuf() { std::free(p); }
    void resize(size_t m) {
        if (m > cap) { std::free(p); p = (T*)std::malloc(m * sizeof(T) + 64); if (!p) { std::cerr << "bella_hip: out of host memory" << std::endl; std::abort(); } cap = m; }
        n = m;
    }
    T* data() { return p; }
    const T* data() const { return p; }
    const T* begin() const { return p; }
    size_t size() const { return n; }
};

// one context = one GPU: operands in, then per stage [lo, hi): overlap (+ alignment) and the records of ITS columns
struct Worker {
    bella_ctx* ctx = nullptr;
    std::vector<uint64_t> colptr;            // colptrC of the last pass (nreads + 1)
    RawBuf<bella_pair> pairs;
    RawBuf<bella_aln> alns;
    uint64_t nnzc = 0;
};
// what the last call of this process did (tests, logs): every column must be computed by the numeric phase exactly once
struct CallStats {
    uint64_t numeric_columns = 0;    // sum over the contexts of the columns their numeric passes computed
    uint64_t numeric_passes = 0, symbolic_passes = 0;
    uint64_t nreads = 0;
    int stages = 0, contexts = 0;
    uint64_t layout_B_bytes_max = 0, layout_B_bytes_sum = 0;   // B' per context: follows the partition
    uint64_t host_upload_bytes = 0;  // matrix bytes that went host -> device, all contexts together
    uint64_t nnzc = 0, lines = 0;    // nnz(C); lines written
    double overlap_seconds = 0, align_seconds = 0, write_seconds = 0;   // wall clock, summed over the stages
};
inline CallStats& last_call_stats() { static CallStats s; return s; }   // single caller, like HashSpGEMM itself (not re-entrant: overlap.hpp:92)

// the reference's printLog (include/common/common.h:40-44): "INFO:\tfile(line)\tname = value" on stderr
#define BELLA_HIP_LOGT(tag, var) do { std::cerr << "INFO:\t" << (tag) << "(" << __LINE__ << ")\t" << #var << " = " << (var) << std::endl; } while (0)

inline void on_all(int N, const std::function<void(int)>& fn) {             // one host thread per context
    if (N == 1) { fn(0); return; }
    std::vector<std::thread> th;
    for (int g = 0; g < N; ++g) th.emplace_back(fn, g);
    for (auto& t : th) t.join();
}

// One slab of device memory up front (bella_hip_reserve): every stage then cuts its buffers from it instead of going to the driver --
// the first hipMalloc of a multi-GB buffer costs tens of ms per GB on ROCm 7.2.  Sized like bench.py sizes it: 44 bytes per base of the
// read set (what counting + assembly + one pass hold at their peak), at least 1 GB; a failed reservation is not an error (the stages
// allocate for themselves).  BELLA_HIP_NO_RESERVE=1 leaves it out.
inline void reserve_for(bella_ctx* ctx, uint64_t total_bases, int contexts_per_device) {
    if (std::getenv("BELLA_HIP_NO_RESERVE")) return;
    uint64_t per_base = 44;
    if (const char* e = std::getenv("BELLA_HIP_RESERVE_BYTES_PER_BASE")) per_base = (uint64_t)std::strtoull(e, nullptr, 10);
    uint64_t want = per_base * total_bases;
    if (want < (1ull << 30)) want = 1ull << 30;
    if (contexts_per_device > 1) want /= (uint64_t)contexts_per_device;
    double ms = 0.0;
    while (want >= (1ull << 29) && bella_hip_reserve(ctx, want, &ms) != 0) want /= 2;   // (a slab the device cannot give: try half)
}

struct StageOpts {
    int N = 1;                       // contexts
    uint32_t nreads = 0;
    bella_params p{};
    int paf = 0;
    double total_memory_mb = 0;      // BELLApars::totalMemory (-m); <= 0: one stage
    double per_nnz = 20.0;           // sizeof(spmatPtr_) + sizeof(uint32_t) = 16 + 4 (overlap.hpp:365-404)
    const char* filename = nullptr;  // the output file: lines are APPENDED
    const char* tag = "bella_hip_driver.hpp";   // the log lines' file name
    int exact = 0;                   // alignments by the exact (growing band) X-drop of the reference's CUDA build
};

// Stage plan (overlap.hpp:365-404,682-710), passes, alignment, output.  W[g].ctx holds the operands, laid out for the output columns
// i % N == g.  The products (estimateFLOP, a sum over a count stream) bound nnz(C) from above: if even they fit one stage the numeric phase
// runs at once over all columns; otherwise the symbolic phase (bella_hip_count_pairs = the reference's estimateNNZ_Hash + prefixsum,
// :674-679) gives the exact colptrC the boundaries are taken from -- every column is computed by the numeric phase exactly once.  With
// more than one stage the output is formed stage by stage (each pass holds its own columns only) and APPENDED -- the reference overwrites
// the file from offset 0 in every stage (overlap.hpp:613-636, a defect); the file written here is the single-stage one.
// stdout: nnz(C) (overlap.hpp:686) and, when aligning, the number of lines written per stage (:771).
inline void run_stages(std::vector<Worker>& W, const StageOpts& o, const char* const* names, const uint32_t* lens) {
    const int N = o.N;
    const uint32_t nreads = o.nreads;
    const bella_params& p = o.p;
    const char* const tag = o.tag;
    auto do_overlap = [&](uint32_t lo, uint32_t hi) {                        // the numeric phase on the columns [lo, hi) of every context
        on_all(N, [&](int g) {
            Worker& w = W[(size_t)g];
            check(w.ctx, bella_hip_set_column_range(w.ctx, lo, hi - lo), "bella_hip_set_column_range");
            uint64_t flops = 0;
            check(w.ctx, bella_hip_overlap(w.ctx, &p, &w.nnzc, &flops), "bella_hip_overlap");
            w.colptr.assign((size_t)nreads + 1, 0);
            check(w.ctx, bella_hip_get_pairs(w.ctx, nullptr, nullptr, w.colptr.data()), "bella_hip_get_pairs");
        });
    };
    auto fetch = [&]() {                                                     // the records of the last pass (+ their alignments)
        on_all(N, [&](int g) {
            Worker& w = W[(size_t)g];
            w.pairs.resize(w.nnzc);
            check(w.ctx, bella_hip_get_pairs(w.ctx, w.pairs.data(), nullptr, nullptr), "bella_hip_get_pairs");
            if (!p.skip_alignment) {
                uint64_t npass = 0;
                if (o.exact) check(w.ctx, bella_hip_align_pairs_exact(w.ctx, &p, &npass), "bella_hip_align_pairs_exact");
                else check(w.ctx, bella_hip_align_pairs(w.ctx, &p, &npass), "bella_hip_align_pairs");
                w.alns.resize(w.nnzc);
                if (w.nnzc) check(w.ctx, bella_hip_get_alignments(w.ctx, w.alns.data()), "bella_hip_get_alignments");
            }
        });
    };
    auto merged_colptr = [&](std::vector<uint64_t>& colptrC) {               // column i lives on context i % N
        colptrC.assign((size_t)nreads + 1, 0);
        for (uint32_t i = 0; i < nreads; ++i) {
            const Worker& w = W[(size_t)(i % (uint32_t)N)];
            colptrC[i + 1] = colptrC[i] + (w.colptr[i + 1] - w.colptr[i]);
        }
    };
    CallStats& cs = last_call_stats();
    cs = CallStats();
    const double free_memory = o.total_memory_mb * 1024 * 1024;               // estimateMemory, overlap.hpp:365-404 (no LINUX/OSX define)
    const double safety_net = 1.5;                                            // overlap.hpp:92
    const double per_nnz = o.per_nnz;
    std::vector<uint64_t> colptrC;
    uint64_t nnzc = 0;
    std::vector<uint64_t> wflops((size_t)N, 0);
    on_all(N, [&](int g) { check(W[(size_t)g].ctx, bella_hip_count_pairs(W[(size_t)g].ctx, &p, nullptr, nullptr, &wflops[(size_t)g]), "bella_hip_count_pairs (flops)"); });
    uint64_t flops_all = 0;
    for (uint64_t f : wflops) flops_all += f;
    bool computed = false;                                                    // the numeric phase already ran over all columns
    const bool no_budget = !(free_memory > 0.0);                              // -m 0 or unset: one stage (the formula would divide by it)
    const auto t_first = std::chrono::steady_clock::now();
    if (no_budget || safety_net * (double)flops_all * per_nnz <= free_memory) {
        do_overlap(0, nreads);
        computed = true;
    } else {
        on_all(N, [&](int g) {
            Worker& w = W[(size_t)g];
            w.colptr.assign((size_t)nreads + 1, 0);
            uint64_t fl = 0;
            check(w.ctx, bella_hip_count_pairs(w.ctx, &p, w.colptr.data(), &w.nnzc, &fl), "bella_hip_count_pairs");
        });
    }
    cs.overlap_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_first).count();
    merged_colptr(colptrC);
    nnzc = colptrC[nreads];
    std::cout << nnzc << std::endl;                                           // overlap.hpp:686
    const uint64_t required_memory = (uint64_t)(safety_net * nnzc * per_nnz);
    int stages = no_budget ? 1 : (int)std::ceil((double)required_memory / free_memory);       // overlap.hpp:683
    if (stages < 1) stages = 1;
    const uint64_t nnzcperstage = no_budget ? nnzc + 1 : (uint64_t)(free_memory / (safety_net * per_nnz));
    std::vector<uint32_t> colStart((size_t)stages + 1, 0);
    for (int i = 1; i < stages; ++i) {                                        // overlap.hpp:704-710
        auto upper = std::upper_bound(colptrC.begin(), colptrC.end(), (uint64_t)i * nnzcperstage);
        colStart[(size_t)i] = (uint32_t)(upper - colptrC.begin() - 1);
    }
    colStart[(size_t)stages] = nreads;

    for (int b = 0; b < stages; ++b) {
        const uint32_t lo = colStart[(size_t)b], hi = colStart[(size_t)b + 1];
        const auto t_stage = std::chrono::steady_clock::now();
        if (!computed) {
            do_overlap(lo, hi);                                               // (computed: one stage was certain, the pass over all columns ran above)
            cs.overlap_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_stage).count();
        }
        const auto t_align = std::chrono::steady_clock::now();
        fetch();
        const double aligntime = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_stage).count();
        cs.align_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_align).count();
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
                if (!p.skip_alignment) ma.insert(ma.end(), w.alns.begin() + (std::ptrdiff_t)w.colptr[i], w.alns.begin() + (std::ptrdiff_t)w.colptr[i + 1]);
            }
            pp = mp.data(); aa = ma.data(); np = mp.size();
        }
        bella_write_stats ws;
        const int wrc = bella_hip_write_output(o.filename, &p, o.paf ? 1 : 0, nreads, names, lens, pp, p.skip_alignment ? nullptr : aa, np, 0, &ws);
        if (wrc) check(nullptr, wrc, "bella_hip_write_output");
        cs.write_seconds += ws.seconds;
        cs.lines += ws.lines;
        const std::string ColumnsRange = "[" + std::to_string(lo) + " - " + std::to_string(hi) + "]";
        BELLA_HIP_LOGT(tag, ColumnsRange);
        if (!p.skip_alignment) {                                              // the per-stage statistics of overlap.hpp:750-777
            const std::string AlignmentTime = std::to_string(aligntime) + " seconds";
            BELLA_HIP_LOGT(tag, AlignmentTime);
            const std::string AlignmentRate = std::to_string((long long)((double)ws.aligned_bases / aligntime)) + " bases/second";
            BELLA_HIP_LOGT(tag, AlignmentRate);
            const std::string AverageReadLength = std::to_string(ws.aligned_pairs ? (long long)((double)ws.total_read_len / (2.0 * (double)ws.aligned_pairs)) : 0LL);
            BELLA_HIP_LOGT(tag, AverageReadLength);
            const std::string PairsAligned = std::to_string(ws.aligned_pairs);
            BELLA_HIP_LOGT(tag, PairsAligned);
            std::cout << ws.lines << std::endl;                               // overlap.hpp:771 (per stage)
            const std::string AverageLengthSuccessfulAlignment = std::to_string(ws.lines ? (long long)((double)ws.bases_passed / (double)ws.lines) : 0LL) + " bps";
            BELLA_HIP_LOGT(tag, AverageLengthSuccessfulAlignment);
            const uint64_t nfail = ws.aligned_pairs - ws.lines;
            const std::string AverageLengthFailedAlignment = std::to_string(nfail ? (long long)((double)ws.bases_failed / (double)nfail) : 0LL) + " bps";
            BELLA_HIP_LOGT(tag, AverageLengthFailedAlignment);
        }
        const uint64_t LinesOutputted = ws.lines;
        BELLA_HIP_LOGT(tag, LinesOutputted);
        const std::string OutputtingTime = std::to_string(ws.seconds) + " seconds";
        BELLA_HIP_LOGT(tag, OutputtingTime);
    }
    cs.nreads = nreads; cs.stages = stages; cs.contexts = N; cs.nnzc = nnzc;
    for (auto& w : W) {
        bella_timings tm;
        bella_memory mm;
        if (bella_hip_get_timings(w.ctx, &tm) == 0) { cs.numeric_columns += tm.numeric_columns; cs.numeric_passes += tm.numeric_passes; cs.symbolic_passes += tm.symbolic_passes; }
        if (bella_hip_get_memory(w.ctx, &mm) == 0) { cs.layout_B_bytes_sum += mm.layout_B_bytes; cs.layout_B_bytes_max = std::max<uint64_t>(cs.layout_B_bytes_max, mm.layout_B_bytes); }
    }
    const uint64_t NumericColumns = cs.numeric_columns, NumericPasses = cs.numeric_passes, SymbolicPasses = cs.symbolic_passes;
    BELLA_HIP_LOGT(tag, NumericColumns);
    BELLA_HIP_LOGT(tag, NumericPasses);
    BELLA_HIP_LOGT(tag, SymbolicPasses);
}

}  // namespace bella_hip_detail
