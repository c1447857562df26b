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
//       498-637; align.hpp:226-229), the call drives min(numGPU, devices) contexts from one host thread each.  Reads are replicated;
//       the matrix is 1D row-block partitioned: context g uploads ONLY the rows of B of its read block (bella_hip_set_B_panel), the
//       blocks are exchanged device to device (bella_hip_allgather_panels on the in-process transport: the contexts share this
//       process; across processes the same call runs on RCCL), and every context lays out B' for its own output columns i % N == g
//       only (bella_hip_set_partition before the exchange).  The results are merged back into the reference's column order.
//   bpars.totalMemory / userDefMem (-m): stage count and stage boundaries by the reference's own formula (estimateMemory,
//       overlap.hpp:365-404; stages :682-710): ceil(1.5 * nnz(C) * (sizeof(spmatPtr_) + sizeof(uint32_t)) / free_memory), boundaries by
//       upper_bound on colptrC.  nnz(C) and colptrC come from the SYMBOLIC phase alone (bella_hip_count_pairs = estimateFLOP +
//       estimateNNZ_Hash + prefixsum, as in the reference, overlap.hpp:667-679) -- and not even that when the product count, an upper
//       bound of nnz(C), already fits one stage: every column is computed by the numeric phase exactly once.  With more than one
//       stage the output is formed stage by stage (each pass holds its own columns only) and APPENDED -- the reference overwrites
//       the file from offset 0 in every stage (overlap.hpp:613-636, a defect); the file written here is the single-stage one.
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
#include <mutex>
#include <string>
#include <vector>

#include "bella_hip.h"
#include "bella_hip_driver.hpp"      // stage plan, passes, alignment, output file and stdout protocol (shared with the native command line)

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
    StageOpts o;
    o.N = N;
    o.nreads = nreads;
    o.p.kmer_size = bpars.kmerSize;
    o.p.bin_size = bpars.binSize;
    o.p.xdrop = bpars.xDrop;
    o.p.skip_alignment = bpars.skipAlignment;
    o.p.error_rate = bpars.errorRate;
    o.p.delta_chernoff = bpars.deltaChernoff;
    o.paf = bpars.outputPaf ? 1 : 0;
    o.total_memory_mb = bpars.totalMemory;
    o.per_nnz = (double)(sizeof(spmatPtr_) + sizeof(uint32_t));
    o.filename = filename;
    o.tag = "bella_hip_shim.hpp";

    std::vector<Worker> W((size_t)N);
    uint8_t comm_id[BELLA_HIP_COMM_ID_BYTES];
    if (N > 1) check(nullptr, bella_hip_comm_id_local(comm_id), "bella_hip_comm_id_local");
    on_all(N, [&](int g) {
        Worker& w = W[(size_t)g];
        check(nullptr, bella_hip_init(g % ndev, &w.ctx), "bella_hip_init");
        // one slab up front: the matrix, its layout and a pass cut their buffers from it (6 + 26 bytes per nonzero and about 70 per
        // product: about as much as the reads' 44 bytes per base on PacBio-like input)
        reserve_for(w.ctx, offs[nreads], (N + ndev - 1) / ndev);
        check(w.ctx, bella_hip_set_reads(w.ctx, (const uint8_t*)flat.data(), offs.data(), nreads), "bella_hip_set_reads");
        if (N == 1) {
            check(w.ctx, bella_hip_set_B(w.ctx, bpars.kmerSize, (uint32_t)B.rows, B.colptr, B.rowids, B.values), "bella_hip_set_B");
            return;
        }
        // row block g of B up, the other blocks from the peers' devices; the layout for the output columns i % N == g
        const uint32_t lo = (uint32_t)((uint64_t)nreads * (uint64_t)g / (uint64_t)N), hi = (uint32_t)((uint64_t)nreads * (uint64_t)(g + 1) / (uint64_t)N);
        check(w.ctx, bella_hip_set_partition(w.ctx, (uint32_t)g, (uint32_t)N), "bella_hip_set_partition");
        check(w.ctx, bella_hip_comm_init_local(w.ctx, N, g, comm_id), "bella_hip_comm_init_local");
        check(w.ctx, bella_hip_set_B_panel(w.ctx, bpars.kmerSize, (uint32_t)B.rows, lo, hi - lo, B.colptr, B.rowids, B.values), "bella_hip_set_B_panel");
        check(w.ctx, bella_hip_allgather_panels(w.ctx), "bella_hip_allgather_panels");
    });
    std::string().swap(flat);
    // names and lengths once; the writer (bella_hip_write_output: per-thread buffers, offset writes, overlap.hpp:603-642) appends
    std::vector<const char*> names(nreads);
    std::vector<uint32_t> lens(nreads);
    for (uint32_t r = 0; r < nreads; ++r) { names[r] = reads[r].nametag.c_str(); lens[r] = (uint32_t)reads[r].seq.length(); }
    run_stages(W, o, names.data(), lens.data());
    last_call_stats().host_upload_bytes = (uint64_t)6 * (uint64_t)B.colptr[nreads] + (uint64_t)4 * ((uint64_t)nreads + 1) * (uint64_t)(N == 1 ? 1 : 0);
    for (auto& w : W) bella_hip_destroy(w.ctx);
}

// ---- the align.hpp call surface ---------------------------------------------------------------------------------------------
namespace bella_hip {

namespace detail {
struct ThreadCtx {                            // xavierAlign is called concurrently from the reference's OpenMP pair loop (overlap.hpp:565)
    bella_ctx* ctx = nullptr;
    ~ThreadCtx() { if (ctx) bella_hip_destroy(ctx); }
};
inline void thread_reads_invalidate();
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
    detail::thread_reads_invalidate();                                         // (the per-pair form re-uploads the registered read set)
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

// ---- include/align.hpp:152, per pair ------------------------------------------------------------------------------------------
// The reference calls xavierAlign(seq1, seq2, seq1len, i, j, xDrop, kmerSize) from its OpenMP pair loop (overlap.hpp:565) with the
// strings of two reads of the read set.  Uploading two strings per call is what makes a per-pair GPU call useless, so the per-pair
// form here works on RESIDENT reads: use_reads(reads) once per run registers the read set (a pointer, nothing is copied), every
// calling thread's context uploads it at its first call (one upload per thread and run), and
//     xavierAlign(rid, cid, i, j, xDrop, kmerSize)
// only sends the 12-byte seed and launches one extension pair.  Same xavierResult as the reference for reads[rid].seq / reads[cid].seq.
// A maintainer who keeps RunPairWiseAlignments changes its call to pass the two read ids (they are at hand there: overlap.hpp:531-565).
// Still ONE launch and one synchronisation per pair: a lane of this kernel runs one extension, so a lone pair takes milliseconds
// where the batch takes microseconds per pair -- for throughput use alignXavier above, or the HashSpGEMM overload, which aligns all
// candidate pairs of a stage in one batch.
namespace detail {
struct ResidentReads {
    const readVector_* reads = nullptr;
    uint64_t generation = 0;
};
inline ResidentReads& resident() { static ResidentReads r; return r; }
struct ThreadReads { uint64_t generation = 0; };
inline ThreadReads& thread_reads() { static thread_local ThreadReads t; return t; }
inline void thread_reads_invalidate() { thread_reads().generation = 0; }
}  // namespace detail

// registers the read set the per-pair calls refer to (call once, before the pair loop; `reads` must outlive the loop)
inline void use_reads(const readVector_& reads) {
    detail::ResidentReads& r = detail::resident();
    r.reads = &reads;
    r.generation++;
}

inline xavierResult xavierAlign(uint32_t rid, uint32_t cid, int i, int j, int xDrop, int kmerSize) {
    using bella_hip_detail::check;
    detail::ResidentReads& rr = detail::resident();
    if (!rr.reads) { std::cerr << "bella_hip::xavierAlign(rid, cid, ...): call bella_hip::use_reads(reads) first" << std::endl; std::abort(); }
    bella_ctx* ctx = detail::thread_context();
    detail::ThreadReads& tr = detail::thread_reads();
    if (tr.generation != rr.generation) {                                      // this thread's context does not hold the read set yet
        const readVector_& reads = *rr.reads;
        std::vector<uint64_t> offs(reads.size() + 1, 0);
        for (size_t r = 0; r < reads.size(); ++r) offs[r + 1] = offs[r] + reads[r].seq.size();
        std::string flat;
        flat.reserve(offs[reads.size()]);
        for (size_t r = 0; r < reads.size(); ++r) flat += reads[r].seq;
        check(ctx, bella_hip_set_reads(ctx, (const uint8_t*)flat.data(), offs.data(), (uint32_t)reads.size()), "bella_hip_set_reads");
        tr.generation = rr.generation;
    }
    bella_seed sd;
    sd.rid = rid; sd.cid = cid; sd.seedH = (uint16_t)i; sd.seedV = (uint16_t)j;
    bella_params p;
    p.kmer_size = (uint16_t)kmerSize; p.bin_size = 500; p.xdrop = (uint16_t)xDrop; p.skip_alignment = 0; p.error_rate = 0.15; p.delta_chernoff = 0.1;
    bella_aln al;
    check(ctx, bella_hip_xdrop_batch(ctx, &sd, 1, &p, &al), "bella_hip_xdrop_batch");
    xavierResult out;
    out.score = al.score;
    out.strand = al.strand ? "c" : "n";
    out.seed.beginPositionH = al.begH; out.seed.beginPositionV = al.begV;
    out.seed.endPositionH = al.endH; out.seed.endPositionV = al.endV;
    return out;
}

// The reference's own signature (two strings).  Kept so that an unchanged call site compiles and gives the reference's answers, but it is
// the SLOW path -- both reads are uploaded at every call -- and says so once per process on stderr.
[[deprecated("uploads both reads per call: register the reads with bella_hip::use_reads and call xavierAlign(rid, cid, i, j, xDrop, kmerSize), or batch with alignXavier")]]
inline xavierResult xavierAlign(const std::string& row, const std::string& col, int rowLen, int i, int j, int xDrop, int kmerSize) {
    (void)rowLen;                                                              // == row.size() at the reference's call site (overlap.hpp:565)
    static std::once_flag once;
    std::call_once(once, [] {
        std::cerr << "bella_hip::xavierAlign(row, col, ...): slow path -- both reads are uploaded at every call; use bella_hip::use_reads + "
                     "xavierAlign(rid, cid, i, j, xDrop, kmerSize) for reads of the read set, alignXavier for batches" << std::endl;
    });
    BELLApars bp;
    bp.kmerSize = (unsigned short)kmerSize;
    bp.xDrop = (unsigned short)xDrop;
    bp.errorRate = 0.15;
    std::vector<xavierResult> out;
    alignXavier({row}, {col}, {SeedX(i, j, kmerSize)}, bp, out);
    return out[0];
}

}  // namespace bella_hip
