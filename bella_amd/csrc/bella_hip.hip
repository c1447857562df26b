// bella_hip.hip -- libbella_hip.so: context, memory, launch sequencing and the C ABI (include/bella_hip.h).
// gfx950 only.  No CPU execution path: every entry point either runs the HIP kernels or returns an error.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <initializer_list>
#include <iterator>
#include <map>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bella_hip.h"
#include "assemble.hpp"
#include "comm.hpp"
#include "core.hpp"
#include "fastq.hpp"
#include "kcount.hpp"
#include "logan.hpp"
#include "order.hpp"
#include "spgemm.hpp"
#include "util.hpp"
#include "wide.hpp"
#include "writer.hpp"
#include "xdrop.hpp"
#include "xdrop_packed.hpp"

using namespace bella;

namespace {

// Large device buffers a context gives up are kept on a free list and handed to the next request they fit, instead of going back to the
// driver: hipMalloc hands out freshly freed VRAM only after the driver has wiped it, and a phase that frees tens of GB right before
// the next one asks for as much waits about a second for that (200k reads: assembly 1.05 s instead of 60 ms, first pass 0.75 s
// instead of 15 ms).  Same synchronisation as hipFree (the device is idle when a buffer changes hands); what is pooled counts as free
// memory wherever the library sizes something by it, and goes back to the driver when an allocation fails or the context is destroyed.
// An ARENA (bella_hip_reserve): one slab taken from the driver up front, from which the buffers below are cut.  The first hipMalloc
// of a multi-GB buffer costs ~30 ms per GB on this stack (the driver maps and wipes the pages): at 100k reads the first k-mer count paid
// 850 ms for its 30 GB of sort buffers and the first assembly 210 ms, against 47 and 21 ms of kernels.  A process that will run the
// pipeline reserves once; every stage then finds its memory in place.  First fit over an offset-ordered free list with coalescing.
struct Arena {
    char* base = nullptr;
    size_t size = 0, in_use = 0, peak = 0;
    std::map<size_t, size_t> holes;                             // offset -> length of the free ranges
    static size_t round(size_t n) { return (n + 4095) & ~(size_t)4095; }
    bool owns(const void* p) const { return base && (const char*)p >= base && (const char*)p < base + size; }
    size_t free_bytes() const { return size - in_use; }
    // the largest single buffer the slab can still give (first fit over a fragmented free list: the sum of the holes is not it)
    size_t largest_hole() const { size_t m = 0; for (const auto& h : holes) m = std::max(m, h.second); return m; }
    void* take(size_t n) {
        n = round(n);
        for (auto it = holes.begin(); it != holes.end(); ++it)
            if (it->second >= n) {
                const size_t off = it->first, len = it->second;
                holes.erase(it);
                if (len > n) holes.emplace(off + n, len - n);
                in_use += n;
                peak = std::max(peak, in_use);
                return base + off;
            }
        return nullptr;
    }
    void give(void* p, size_t n) {
        n = round(n);
        in_use -= n;
        size_t off = (size_t)((char*)p - base);
        auto nx = holes.lower_bound(off);
        if (nx != holes.begin()) { auto pv = std::prev(nx); if (pv->first + pv->second == off) { off = pv->first; n += pv->second; holes.erase(pv); } }
        if (nx != holes.end() && off + n == nx->first) { n += nx->second; holes.erase(nx); }
        holes.emplace(off, n);
    }
};
struct BufPool {
    std::vector<std::pair<void*, size_t>> free;
    size_t bytes = 0;
    Arena arena;
};
constexpr size_t kPoolMinBytes = 128ull << 20, kPoolMaxBytes = 160ull << 30;
struct Buf {
    void* p = nullptr;
    size_t cap = 0;
    BufPool* pool = nullptr;     // where the buffer goes when it is released (set by ensure_bytes)
};

// Column tiers by product count.  A column runs in one 512-thread workgroup with 14.5 B of LDS per product (19 B with the
// half-size key tables of pair-rich inputs); tiers whose workgroups need the same share of a CU's 160 KB (1/4, 1/3, 1/2, all of
// it) form one LDS class = ONE launch that walks its tiers' lists from the largest columns down.  Last: the global-workspace
// tier.  BELLA_HIP_TIERS=a,b,c (read once per context; tuning aid / tests) overrides the LDS caps.
constexpr uint32_t kNumTiers = 15;  // at most
constexpr uint32_t kDefaultTiers = 14;
const uint32_t kTierCapsDefault[kNumTiers] = {384, 689, 1000, 1394, 2048, 2752, 3712, 4600, 5568, 8192, 11008, 65535, 65535, 65535};
const uint32_t kTierCapsHalf[kNumTiers] = {300, 526, 800, 1064, 1600, 2105, 2789, 3500, 4096, 6144, 8192, 11008, 65535, 65535};
constexpr uint32_t kFullTableMaxCap = 300;               // pair-rich inputs: the smallest tier holds as many pairs as products
constexpr uint32_t kHalfTableMaxCap = 4096;              // above: quarter-size key tables on any input
// LDS classes: the share of a CU's 160 KB a column's workgroup needs, and the workgroup size that goes with it.  A column costs
// a fixed ~18 us of latency (three dependent memory round trips, ~16 barriers) plus time per product: small columns get small
// workgroups so that 16 or 8 of them share a CU; where only one or two columns fit a CU they get 1024 threads -- every class
// keeps the CU's 32 wavefront slots filled.
constexpr uint32_t kNumClasses = 6;
const size_t kClassLds[kNumClasses] = {10240, 20480, 40160, 54608, 81920, 163840};   // 16, 8, 4, 3, 2, 1 workgroups per CU
#ifndef BELLA_CLASS_BLOCKS
#define BELLA_CLASS_BLOCKS {128, 256, 512, 512, 1024, 1024}
#endif
const int kClassBlock[kNumClasses] = BELLA_CLASS_BLOCKS;   // measured (tools/ab_blocks.sh): DESIGN.md 4.1
#ifndef BELLA_LONG_LIST_MAX_CAP
#define BELLA_LONG_LIST_MAX_CAP 2752
#endif
constexpr uint32_t kLongListMaxCap = BELLA_LONG_LIST_MAX_CAP;   // long-list inputs: LDS tiers above this many products are closed
constexpr uint32_t kMidToWideMin = 16;      // columns above the LDS tiers in one pass from which on they take the sort-based path
constexpr uint32_t kRerunGrid = 256;        // persistent workgroups of the rerun launch (columns an LDS tier handed over)
constexpr uint32_t kGlobalGrid = 1024;      // persistent workgroups of the global path (1024 threads each: latency-bound, two resident per CU)
constexpr uint32_t kAsmGrid = 128;

struct CastU64 {
    __host__ __device__ uint64_t operator()(const uint32_t& v) const { return (uint64_t)v; }
};
struct CastU8 {
    __host__ __device__ uint32_t operator()(const uint8_t& v) const { return (uint32_t)v; }
};
struct CountU64 {                                        // pair counts: without the "already in slot order" mark
    __host__ __device__ uint64_t operator()(const uint32_t& v) const { return (uint64_t)(v & ~kOrderedBit); }
};

}  // namespace

struct bella_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    uint32_t debug = 0;

    // reads
    uint32_t nreads = 0;
    uint64_t total_bases = 0;
    Buf packed, roff;
    // B (reference layout) and device layout
    uint32_t nkmers = 0, kmer_size = 0;
    uint64_t nnz = 0;
    bool have_reads = false, have_matrix = false, have_pairs = false, have_alns = false;
    std::vector<std::string> names;      // read names when the reads came through bella_hip_load_fastq
    std::vector<uint32_t> host_lens;
    bool have_tuples = false;            // device-resident tuples + dictionary of bella_hip_count_kmers
    uint64_t kc_ntuples = 0;
    uint32_t kc_nkmers = 0, kc_k = 0;
    uint32_t kc_bfirst = 0, kc_brows = 0; // the read block the device-resident tuples cover (all reads unless counted distributed)
    Buf kc_nk, kc_koff, kc_hist, kc_keys, kc_alt, kc_runlen, kc_flag, kc_slot, kc_nruns, kc_dcode, kc_dcount, kc_hkey, kc_hval, kc_found,
        kc_tstart, kc_cursor, kc_sel, kc_ringo, kc_ringp, kc_opos, kc_oid, kc_opos2, kc_oid2;
    Buf Bptr, Bk, Bpos, Bent, rowF, Aent, Aent2, Aov, Arow, Bloc;
    bool have_rowlists = false;          // Aent2 / Arow hold the row lists of the current layout
    bool want_rowlists = false;          // BELLA_TUNE_ROW_LISTS: build them with the next layouts (when they fit)
    uint32_t part_first = 0, part_stride = 1;
    uint32_t layout_first = 0, layout_stride = 1;   // the partition the current layout was built for (B' / row lists: owned columns only)
    uint64_t owned_nnz = 0;              // nonzeros of the owned columns (what the current layout was built from)
    uint64_t live_nnz = 0;               // B' entries of the current layout: those with a later read (all of them with BELLA_TUNE_COMPACT_B 1)
    uint64_t sym_sig[6] = {};            // what flops / nnzC were last cleared for
    uint64_t layout_gen = 0;             // bumped by every build of the device layout
    uint32_t range_lo = 0, range_hi = 0xFFFFFFFFu;   // stage: the contiguous column range computed by the next passes
    bool have_panel = false;
    uint32_t panel_first = 0, panel_rows = 0;
    uint64_t panel_nnz = 0;
    // assembly temporaries
    Buf t_kmer, t_read, t_pos, tstart, Bk_tmp, Bpos_tmp, rowcnt, asm_ws, asm_cls;
    Buf lk_key, lk_key2, lk_val, lk_val2, w, wscan, lk_rinfo;
    // overlap
    uint64_t flops = 0, npairs = 0;
    uint32_t pair_ratio1024 = 1024;   // max over sampled columns of 1024 * pairs/products
    bool long_list_sample = false;    // the sampled columns TOGETHER hold >= 64 products per pair
#ifdef BELLA_DEV_PROF
    Buf prof;
#endif
    Buf flopsr, flopptr, nnzC, colptrC, rowlists, tiercaps, tmp_pairs, tmp_ext, pairs, ext, sortscr, ws,
        status, cubtmp, plist_hv, overflow, ctl, retry, orderlist, order_ws;
    bool order_attr = false;
    Buf w_f, w_off, w_key, w_key2, w_idx, w_idx2, w_hv, w_ovfl, w_plist, w_scr, w_rlen, w_rstart, w_rrank, w_redo, w_segfirst,
        w_toff, w_table, w_nruns, w_desc, w_gtab, w_gcount, w_gbase, w_rfirst, w_aent2, w_aov;
    uint32_t n_wide = 0;
    uint32_t n_retry = 0;
    uint32_t n_overflow = 0;
    // alignment
    uint64_t nalns = 0;
    Buf alns, seeds, xest, xest2, xids, xorder, xres, xstate, xlive, lg_res, lg_redo, lg_scratch;
    bella_timings tm{};
    hipEvent_t ev[12]{};
    uint32_t* pinned = nullptr;          // 128 host words the per-pass read backs land in
    Stager stager;                       // pinned bounce buffers of the large host <-> device copies
    bella_ingest_stats ingest{};         // of the last bella_hip_load_fastq
    int caps_state = 0;
    ncclComm_t comm = nullptr;           // communicator of bella_hip_comm_init (one rank per context)
    const Rccl* api = nullptr;           // its transport: RCCL (comm.hpp: rccl()) or the in-process one (loopback())
    int comm_ranks = 0, comm_rank = 0;
    Buf comm_meta;
    bool pass_known = false;             // tier lengths and product total of the last pass (valid for pass_sig)
    uint64_t pass_sig[6] = {};
    uint32_t pass_tcnt[20] = {};
    uint64_t pass_products = 0;
    uint64_t pass_wide_products[2] = {};  // ... of the last tier's columns and of the wide columns (what wide.hpp's path will hold)
    bool pass_rare_free = false;         // the last pass on pass_sig needed no rerun, no overflow fold, no wide column
    bool pass_big_free = false;          // ... and had no column with more than 1,024 pairs (k_order_block)
    uint32_t tier_caps[kNumTiers] = {};  // ascending LDS caps, last = 65535 (global-workspace tier); per context
    uint32_t ntiers = 0;
    bool tiers_from_env = false;         // custom tier table (bella_hip_set_tuning)
    uint64_t kcount_budget = 1ull << 30, wide_budget = 1ull << 30;   // items per pass of the counting sort / of the wide-column path
    bool lane_order_ok = true;           // k_lane_order_selftest at init
    uint64_t xdrop_class_min = 4ull * 4096 * 64;   // extensions of a batch from which on the slices run it as four classes (BELLA_TUNE_XDROP_CLASS_MIN)
    BufPool pool;                        // released device buffers waiting for the next request they fit
    uint32_t layout_inline = 0;          // the device layout holds B' entries in the INLINE form (util.hpp); 0 / 1
    // layout and path choices (bella_hip_set_tuning; the matching bella_hip_set_debug bits of earlier rounds are still read as aliases)
    uint32_t tune_layout_order = 0;      // BELLA_TUNE_LAYOUT_ORDER: 0 lists of A' in k-mer order, 1 in order of first appearance in B'
    uint32_t tune_inline = 0;            // BELLA_TUNE_INLINE_ENTRIES: 0 by the size of A' against the last-level cache, 1 never, 2 always
    uint32_t tune_row_path = 0;          // BELLA_TUNE_ROW_PATH: 0 LDS tiers, 1 every column on the global-workspace (repairing) path
    uint32_t tune_compact_b = 0;         // BELLA_TUNE_COMPACT_B: 0 the layout drops the B' entries that have no later read, 1 keeps them
    uint32_t tune_dist_layout = 2;       // BELLA_TUNE_DIST_LAYOUT: bella_hip_allgather_panels forms A' by k-mer id ranges over the communicator's ranks
                                         // (1: always, 0: never, 2 = default: on the in-process transport only -- over RCCL the path has not run on
                                         // more than one device yet, and its extra 20 B per nonzero of traffic want a measurement first)
    bool dist_agreed = false;            // ... and every rank of the communicator said it can (bella_hip_allgather_panels asks)
    bool layout_dist = false;            // ... and the current layout was built that way
    uint64_t tune_cache_bytes = 192ull << 20;   // BELLA_TUNE_CACHE_BYTES: A' above this size counts as "larger than the last-level cache"
    uint32_t xdrop_variant = 1;          // 0: one launch in length-sorted order; 1 (default): slices with compaction; 2: packed kernel in pair order; 3: scalar statement
    size_t lds_attr[24] = {};             // largest dynamic-LDS size already granted to each row-kernel instantiation
    hipStream_t side[kNumTiers + 1]{};   // independent tier launches / fold instances run concurrently
    hipEvent_t fork = nullptr, join[kNumTiers + 1]{};
};

namespace {

int fail(bella_ctx* c, int code, const char* fmt, ...) {
    char b[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(b, sizeof(b), fmt, ap);
    va_end(ap);
    if (c) c->err = b;
    return code;
}

#define HIPCHK(c, call)                                                                                           \
    do {                                                                                                          \
        hipError_t e_ = (call);                                                                                   \
        if (e_ != hipSuccess)                                                                                     \
            return fail(c, BELLA_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define KCHK(c) HIPCHK(c, hipGetLastError())
#define NCCLCHK(c, call)                                                                                          \
    do {                                                                                                          \
        ncclResult_t r_ = (call);                                                                                 \
        if (r_ != ncclSuccess)                                                                                    \
            return fail(c, BELLA_ERR_HIP, "%s failed: %s", #call, c->api && c->api->GetErrorString ? c->api->GetErrorString(r_) : "RCCL error"); \
    } while (0)

void release(Buf& b);
void trim_pool(BufPool& pl) {
    for (auto& q : pl.free) (void)hipFree(q.first);
    pl.free.clear();
    pl.bytes = 0;
}
int ensure_bytes(bella_ctx* c, Buf& b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return 0;
    release(b);
    BufPool& pl = c->pool;
    b.pool = &pl;
    const size_t want = bytes + bytes / 16 + 256;
    int best = -1;                                              // best fit on the free list: at least `want`, at most about twice that
    for (int i = 0; i < (int)pl.free.size(); ++i)
        if (pl.free[i].second >= want && pl.free[i].second <= 2 * want + (64ull << 20) && (best < 0 || pl.free[i].second < pl.free[best].second)) best = i;
    if (best >= 0) {
        b.p = pl.free[best].first; b.cap = pl.free[best].second;
        pl.bytes -= b.cap;
        pl.free.erase(pl.free.begin() + best);
        return 0;
    }
    if (pl.arena.base) {                                        // the reserved slab first (bella_hip_reserve): no trip to the driver
        void* q = pl.arena.take(want);
        if (q) { b.p = q; b.cap = Arena::round(want); return 0; }
    }
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess && !pl.free.empty()) {                  // what the free list holds back may be what is missing
        (void)hipGetLastError();
        trim_pool(pl);
        e = hipMalloc(&b.p, want);
    }
    if (e != hipSuccess) {
        b.p = nullptr;
        return fail(c, BELLA_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    b.cap = want;
    return 0;
}
#define ENSURE(c, buf, bytes)                  \
    do {                                       \
        int r_ = ensure_bytes(c, buf, bytes);  \
        if (r_) return r_;                     \
    } while (0)

void release(Buf& b) {
    if (b.p) {
        BufPool* pl = b.pool;
        if (pl && pl->arena.owns(b.p)) {
            (void)hipDeviceSynchronize();                       // (as below: nothing in flight touches the range when it changes hands)
            pl->arena.give(b.p, b.cap);
        } else if (pl && b.cap >= kPoolMinBytes && pl->free.size() < 32 && pl->bytes + b.cap <= kPoolMaxBytes) {
            (void)hipDeviceSynchronize();                       // (what hipFree does: nothing in flight touches the buffer when it changes hands)
            pl->free.emplace_back(b.p, b.cap);
            pl->bytes += b.cap;
        } else {
            (void)hipFree(b.p);
        }
    }
    b.p = nullptr;
    b.cap = 0;
}

template <class T>
T* ptr(const Buf& b) { return (T*)b.p; }

inline unsigned nblk(uint64_t n, unsigned per = 256) { return (unsigned)((n + per - 1) / per); }

int read_status(bella_ctx* c, uint32_t* out) {
    HIPCHK(c, hipMemcpyAsync(out, c->status.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int status_to_error(bella_ctx* c, uint32_t st) {
    if (st & 4u) return fail(c, BELLA_ERR_BAD_BASE, "reads must be upper-case ACGT only");
    if (st & 8u) return fail(c, BELLA_ERR_TUPLE_ORDER, "tuples must be grouped by non-decreasing read id < nreads");
    if (st & 16u) return fail(c, BELLA_ERR_READ_TOO_LONG, "a read has >= 65536 tuples");
    if (st & 32u) return fail(c, BELLA_ERR_BAD_ARG, "k-mer id >= nkmers");
    if (st & 64u) return fail(c, BELLA_ERR_BAD_ARG, "a k-mer occurs in more than 16383 reads");
    if (st & 128u) return fail(c, BELLA_ERR_BAD_ARG, "a tuple's position + k exceeds the length of its read");
    if (st & 2u) return fail(c, BELLA_ERR_ROW_TOO_LARGE, "an output column has >= 65536 products");
    return 0;
}

// exclusive scans (rocPRIM through hipCUB: plumbing, not a hot op)
int scan_u32(bella_ctx* c, const uint32_t* in, uint32_t* out, uint64_t n) {
    size_t tb = 0;
    HIPCHK(c, hipcub::DeviceScan::ExclusiveSum(nullptr, tb, in, out, (int)n, c->stream));
    ENSURE(c, c->cubtmp, tb);
    HIPCHK(c, hipcub::DeviceScan::ExclusiveSum(c->cubtmp.p, tb, in, out, (int)n, c->stream));
    return 0;
}
int scan_u32_to_u64(bella_ctx* c, const uint32_t* in, uint64_t* out, uint64_t n) {
    hipcub::TransformInputIterator<uint64_t, CastU64, const uint32_t*> it(in, CastU64());
    size_t tb = 0;
    HIPCHK(c, hipcub::DeviceScan::ExclusiveSum(nullptr, tb, it, out, (int)n, c->stream));
    ENSURE(c, c->cubtmp, tb);
    HIPCHK(c, hipcub::DeviceScan::ExclusiveSum(c->cubtmp.p, tb, it, out, (int)n, c->stream));
    return 0;
}

float ev_ms(hipEvent_t a, hipEvent_t b) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms;
}

// the row pointers the passes index B' with: the rows of the context's own columns (build_layout; the other rows are empty)
inline const uint32_t* layout_bptr(const bella_ctx* c) { return ptr<uint32_t>(c->Bloc); }

// B (Bptr/Bk/Bpos on device) -> Bent / Aent   (see assemble.hpp): one stable radix sort of the entries by k-mer id (the runs are
// the k-mer lists of A', ascending read id) and coalesced segmented passes; 32 B of temporaries per nonzero.
// The layout follows the context's partition (bella_hip_set_partition): A' is whole, B' entries -- and the optional row lists --
// exist for the owned columns only, so the per-column part of the work and of the memory is 1/stride of the whole.
// The large working buffers of k-mer counting (the sort's two key arrays and the tile arrays: 29 GB at 100k reads) stay with the context
// when the call returns and go when the next stage starts (assembly, device layout): a second count finds them in place, a pipeline's
// peak is what it was -- and no call frees tens of GB only for the next one to ask for them again at once (hipMalloc right after a
// hipFree of that size was measured at ~1 s on ROCm 7.2, against microseconds otherwise).
void release_count_scratch(bella_ctx* c) {
    release(c->kc_keys); release(c->kc_alt); release(c->kc_runlen); release(c->kc_flag); release(c->kc_slot);
}

// (bella_hip_allgather_panels with BELLA_TUNE_DIST_LAYOUT: the formation of A' shared over the ranks; defined behind the collectives)
int layout_dist(bella_ctx* c, uint64_t nown_nnz, uint32_t rmask, uint32_t inl, int kbits, uint32_t** ekey_out, uint64_t** eval_out,
                uint32_t** fkey_out, uint64_t** fval_out);
void layout_dist_fail(bella_ctx* c);
bool layout_dist_eligible(const bella_ctx* c);

// collective: every rank of the context's communicator is inside this call (bella_hip_allgather_panels) -- the only caller that may
// take the shared formation of A'
int build_layout(bella_ctx* c, bool collective = false) {
    release_count_scratch(c);
    const uint64_t nnz = c->nnz;
    const uint32_t nk = c->nkmers;
    const uint32_t pf = c->part_first, ps = c->part_stride;
    if (nnz >= 0xFFFFFFF0ull) return fail(c, BELLA_ERR_BAD_ARG, "nnz(A) must be < 2^32 (KMERINDEX uint32, main.cpp:60)");
    c->have_matrix = false;
    c->have_rowlists = false;
    c->layout_inline = 0;
    c->tm.expand_ms = 0.f;
    // shared formation of A' (layout_dist): every rank agreed to it in bella_hip_allgather_panels, so a rank that fails BEFORE it gets
    // there still says so in the formation's metadata exchange -- the others leave instead of waiting for it
    const bool dist = collective && c->dist_agreed;
    struct Bail { bella_ctx* c; bool armed; ~Bail() { if (armed) layout_dist_fail(c); } } bail{c, dist};
    if (dist && (c->debug & 262144u)) return fail(c, BELLA_ERR_STATE, "debug bit 18: this rank fails before its share of the formation");
    if (!dist) {                                                   // (shared: the sort buffers are sized by the rank's share)
        ENSURE(c, c->lk_key, 4 * nnz);
        ENSURE(c, c->lk_key2, 4 * nnz);
        ENSURE(c, c->lk_val, 8 * nnz);
        ENSURE(c, c->lk_val2, 8 * nnz);
    }
    ENSURE(c, c->lk_rinfo, 8 * ((size_t)c->nreads + 1));
    // (w also holds one word per READ: the owned rows' lengths of a partitioned layout, the rows' product counts of the row lists --
    // a matrix with fewer nonzeros than reads must not shrink it below that)
    ENSURE(c, c->w, 4 * std::max<uint64_t>(nnz + 1, (uint64_t)c->nreads + 2));
    ENSURE(c, c->wscan, 4 * std::max<uint64_t>(nnz + 1, (uint64_t)c->nreads + 2));
    ENSURE(c, c->Aent, 8 * nnz + 64);
    HIPCHK(c, hipMemsetAsync(c->status.p, 0, 4, c->stream));
    HIPCHK(c, hipEventRecord(c->ev[4], c->stream));
    // the context's own row pointers: all rows, or the owned ones (the others keep no entries)
    uint64_t nown_nnz = nnz;
    c->layout_first = pf; c->layout_stride = ps;
    ENSURE(c, c->Bloc, 4 * ((size_t)c->nreads + 2));
    if (ps > 1) {
        uint32_t* len = ptr<uint32_t>(c->w);
        k_layout_own_lengths<<<nblk((uint64_t)c->nreads + 1), 256, 0, c->stream>>>(ptr<uint32_t>(c->Bptr), c->nreads, pf, ps, len);
        KCHK(c);
        int rc = scan_u32(c, len, ptr<uint32_t>(c->Bloc), (uint64_t)c->nreads + 1);
        if (rc) return rc;
        uint32_t n32 = 0;
        HIPCHK(c, hipMemcpyAsync(&n32, ptr<uint32_t>(c->Bloc) + c->nreads, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        nown_nnz = n32;
    } else {
        HIPCHK(c, hipMemcpyAsync(c->Bloc.p, c->Bptr.p, 4 * ((size_t)c->nreads + 1), hipMemcpyDeviceToDevice, c->stream));
    }
    c->owned_nnz = 0;
    uint32_t ratio1024 = 1024;
    c->long_list_sample = false;
    const uint32_t* Bloc = ptr<uint32_t>(c->Bloc);                 // the rows of the owned columns
    // (buffers only ever grow; a context that goes from a whole layout to a partition's gives the difference back)
    if (c->Bent.cap > 16 * nown_nnz + (1u << 20)) release(c->Bent);
    ENSURE(c, c->Bent, 8 * nown_nnz);
    const bool first_app = c->tune_layout_order == 1 || (c->debug & 1024u);
    c->layout_dist = dist;
    if (nnz || dist) {
        int kbits = 1;
        while (kbits < 32 && (1ull << kbits) < (uint64_t)nk) ++kbits;
        const uint32_t rmask = c->nreads <= (1u << 30) ? 0x3FFFFFFFu : 0x7FFFFFFFu;   // read id field of the sort value
        // Default: the lists of A' stay in k-mer order (= the sorted order itself; nothing to scatter, scan or look up) and every pass
        // expands B' x A' itself: the fastest ONE-SHOT call (layout + first pass).  BELLA_TUNE_LAYOUT_ORDER 1 (tests): the lists in
        // order of first appearance in B' (the owner row of a list streams it; three more random-access passes here).
        const uint32_t by_kmer = first_app ? 0u : 1u;
        // entries whose k-mer has exactly one later read carry that read (util.hpp: INLINE form) when bit 31 of the list index and bit 30
        // of a read id are free -- and when A' is larger than the last-level cache (256 MB): the form saves a line from HBM per such
        // entry and pass (row kernels 4.03 -> 3.69 ms at 100k reads, A' = 1.6 GB), but where the gather is a cache hit it only adds a
        // branch (10k reads, A' = 92 MB: 0.360 -> 0.367 ms per step).  BELLA_TUNE_INLINE_ENTRIES 1: the plain form everywhere, as for inputs
        // beyond those bounds; 2: the inline form on any size
        const bool inl_pays = 8ull * nnz > c->tune_cache_bytes || c->tune_inline == 2 || (c->debug & 65536u);
        const uint32_t inl = rmask == 0x3FFFFFFFu && nnz < 0x80000000ull && inl_pays && c->tune_inline != 1 && !(c->debug & 32768u) ? 1u : 0u;
        c->layout_inline = inl;
        uint32_t* ekey = nullptr;                                  // the nown_nnz B' entries of this context's rows {own index, entry}, any order ...
        uint64_t* eval = nullptr;
        const uint32_t* skey = nullptr;                            // ... and two free buffers of that size for the partition pass below
        const uint64_t* sval = nullptr;
        if (dist) {
            uint32_t *fk = nullptr; uint64_t* fv = nullptr;
            bail.armed = false;                                    // (from here on layout_dist speaks for this rank)
            int rcd = layout_dist(c, nown_nnz, rmask, inl, kbits, &ekey, &eval, &fk, &fv);
            if (rcd) return rcd;
            skey = fk; sval = fv;
        } else {
        k_layout_prep<<<nblk(c->nreads, kWaves), kBlock, 0, c->stream>>>(ptr<uint32_t>(c->Bptr), ptr<uint32_t>(c->Bk), ptr<uint16_t>(c->Bpos), c->nreads,
                                                                        ptr<uint32_t>(c->packed), ptr<uint64_t>(c->roff), c->kmer_size, nk,
                                                                        ptr<uint32_t>(c->lk_key), ptr<uint64_t>(c->lk_val),
                                                                        first_app ? ptr<uint32_t>(c->w) : nullptr, ptr<uint32_t>(c->status));
        KCHK(c);
        // bad input (k-mer id out of range, k-mer past the end of its read) stops here: the passes below trust the keys
        uint32_t st0 = 0;
        int rc0 = read_status(c, &st0);
        if (rc0) return rc0;
        rc0 = status_to_error(c, st0);
        if (rc0) return rc0;
        hipcub::DoubleBuffer<uint32_t> dk(ptr<uint32_t>(c->lk_key), ptr<uint32_t>(c->lk_key2));
        hipcub::DoubleBuffer<uint64_t> dv(ptr<uint64_t>(c->lk_val), ptr<uint64_t>(c->lk_val2));
        size_t tb = 0;
        HIPCHK(c, hipcub::DeviceRadixSort::SortPairs(nullptr, tb, dk, dv, (uint64_t)nnz, 0, kbits, c->stream));
        ENSURE(c, c->cubtmp, tb);
        HIPCHK(c, hipcub::DeviceRadixSort::SortPairs(c->cubtmp.p, tb, dk, dv, (uint64_t)nnz, 0, kbits, c->stream));
        skey = dk.Current();
        sval = dv.Current();
        if (!by_kmer) {
            k_layout_heads<<<nblk(nnz), 256, 0, c->stream>>>(skey, sval, nnz, ptr<uint32_t>(c->Bptr), rmask, ptr<uint32_t>(c->w), ptr<uint32_t>(c->status));
            KCHK(c);
            int rc = scan_u32(c, ptr<uint32_t>(c->w), ptr<uint32_t>(c->wscan), nnz);
            if (rc) return rc;
        }
        ekey = dk.Alternate();                                     // (the sort's other buffers are free now)
        eval = dv.Alternate();
        uint32_t* counter = ptr<uint32_t>(c->status) + 7;
        HIPCHK(c, hipMemsetAsync(counter, 0, 4, c->stream));
        uint2* rinfo = ptr<uint2>(c->lk_rinfo);
        k_layout_rinfo<<<nblk(c->nreads), 256, 0, c->stream>>>(Bloc, ptr<uint64_t>(c->roff), c->nreads, rinfo);
        KCHK(c);
        k_layout_emit<<<nblk(nnz, ps > 1 ? kLayoutEmitPartBlock : 256), ps > 1 ? kLayoutEmitPartBlock : 256, 0, c->stream>>>(skey, sval, nnz, ptr<uint32_t>(c->Bptr), Bloc, ptr<uint32_t>(c->wscan), ptr<uint32_t>(c->packed),
                                                        ptr<uint64_t>(c->roff), c->kmer_size, rmask, ptr<uint2>(c->Aent), ekey, eval, by_kmer, pf, ps, counter,
                                                        ptr<uint32_t>(c->status), inl, rinfo);
        KCHK(c);
        {   // a k-mer in more than 16,383 reads overflows the count field of the B' entries: stop before anything trusts them
            uint32_t st1 = 0;
            int rc1 = read_status(c, &st1);
            if (rc1) return rc1;
            rc1 = status_to_error(c, st1);
            if (rc1) return rc1;
        }
        }   // !dist
        if (nown_nnz) {   // one radix pass on the top 8 bits of the entry index: B' is then written region by region
            int ebits = 1;
            while (ebits < 32 && (1ull << ebits) < nown_nnz) ++ebits;
            hipcub::DoubleBuffer<uint32_t> ek(ekey, const_cast<uint32_t*>(skey));
            hipcub::DoubleBuffer<uint64_t> ev(eval, const_cast<uint64_t*>(sval));
            if (ebits > 12) {
                size_t tb2 = 0;
                HIPCHK(c, hipcub::DeviceRadixSort::SortPairs(nullptr, tb2, ek, ev, (uint64_t)nown_nnz, ebits - 8, ebits, c->stream));
                ENSURE(c, c->cubtmp, tb2);
                HIPCHK(c, hipcub::DeviceRadixSort::SortPairs(c->cubtmp.p, tb2, ek, ev, (uint64_t)nown_nnz, ebits - 8, ebits, c->stream));
            }
            k_layout_place<<<((nblk(nown_nnz) + 7u) / 8u) * 8u, 256, 0, c->stream>>>(ek.Current(), ev.Current(), nown_nnz, ptr<uint2>(c->Bent));
            KCHK(c);
        }
        c->owned_nnz = nown_nnz;
        c->live_nnz = nown_nnz;
        ENSURE(c, c->rowF, 4 * ((size_t)c->nreads + 2));
        if (!nown_nnz) HIPCHK(c, hipMemsetAsync(c->rowF.p, 0, 4 * ((size_t)c->nreads + 2), c->stream));
        if (nown_nnz) {
            // one pass over the rows of B': the rows' products (rowF: what estimateFLOP returns, kept like a row pointer) and their live
            // entries; then B' without the entries that have no later read (assemble.hpp: k_layout_live / k_layout_compact): new row
            // pointers by a scan of the live counts, the live entries moved in their order; from here on Bloc indexes the compacted B'
            const bool compact = !c->tune_compact_b;
            uint32_t* len = ptr<uint32_t>(c->w);
            uint32_t* bnew = ptr<uint32_t>(c->wscan);
            k_layout_live<<<nblk((uint64_t)c->nreads + 1, kWaves), kBlock, 0, c->stream>>>(Bloc, ptr<uint2>(c->Bent), c->nreads, inl, compact ? len : nullptr,
                                                                                         ptr<uint32_t>(c->rowF));
            KCHK(c);
            if (compact) {
                int rc = scan_u32(c, len, bnew, (uint64_t)c->nreads + 1);
                if (rc) return rc;
                uint32_t live = 0;
                HIPCHK(c, hipMemcpyAsync(&live, bnew + c->nreads, 4, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                Buf out;
                rc = ensure_bytes(c, out, 8 * (size_t)live);
                if (rc) return rc;
                k_layout_compact<<<nblk(c->nreads, kWaves), kBlock, 0, c->stream>>>(Bloc, bnew, ptr<uint2>(c->Bent), c->nreads, inl, ptr<uint2>(out));
                if (hipGetLastError() != hipSuccess) { release(out); return fail(c, BELLA_ERR_HIP, "k_layout_compact failed to launch"); }
                HIPCHK(c, hipMemcpyAsync(c->Bloc.p, bnew, 4 * ((size_t)c->nreads + 1), hipMemcpyDeviceToDevice, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                std::swap(c->Bent, out);
                release(out);
                c->live_nnz = live;
            }
        }
        // pairs/products on a sample of (owned) columns -> key-table budget of the LDS tiers (see k_sample_pair_ratio), and: is this a
        // long-list input (HiFi-like: at most one pair in 64 products)?
        const size_t bitmap_bytes = 4 * (((size_t)c->nreads + 31) / 32);
        uint32_t samp[2] = {0, 0};
        if (c->owned_nnz && bitmap_bytes <= 128 * 1024) {
            HIPCHK(c, hipMemsetAsync(ptr<uint32_t>(c->status) + 6, 0, 16, c->stream));   // [6] max ratio, [8] pairs, [9] products / 64 of the sample
            const uint32_t ncols = c->nreads / ps + 1;                  // owned columns: first + j * stride
            const uint32_t nsample = ncols < 512 ? ncols : 512;
            const uint32_t step = (ncols / nsample ? ncols / nsample : 1) * ps;
            HIPCHK(c, hipFuncSetAttribute((const void*)k_sample_pair_ratio, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bitmap_bytes));
            k_sample_pair_ratio<<<nsample, kBlock, bitmap_bytes, c->stream>>>(Bloc, ptr<uint2>(c->Bent), ptr<uint2>(c->Aent),
                                                                             c->nreads, pf, step, ptr<uint32_t>(c->status) + 6, inl);
            KCHK(c);
            HIPCHK(c, hipMemcpyAsync(&ratio1024, ptr<uint32_t>(c->status) + 6, 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipMemcpyAsync(samp, ptr<uint32_t>(c->status) + 8, 8, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        // Long-list inputs send most of their columns to the path above the LDS tiers, which groups a column's products in LDS from a
        // product list -- expanded per batch and per pass when the layout has none.  The same expansion once, here, costs a one-shot
        // call nothing and every further pass less: such inputs get the row lists whenever they fit.
        c->long_list_sample = samp[1] > 0 && samp[0] <= samp[1];     // >= 64 products per pair on the sample
        const bool auto_lists = c->long_list_sample && !(c->debug & 1u) && c->tune_row_path != 1 && c->lane_order_ok;
        if (by_kmer && (c->want_rowlists || auto_lists) && c->nreads <= (1u << 30)) {
            // row lists (BELLA_TUNE_ROW_LISTS): products per owned row -> row starts -> the tails of the lists copied in product order.
            // Optional in every respect: if they do not fit next to what a pass needs, or an allocation fails, the layout stands without them.
            int rc = ensure_bytes(c, c->Arow, 8 * ((size_t)c->nreads + 2));
            uint64_t F = 0;
            if (!rc) {
                rc = scan_u32_to_u64(c, ptr<uint32_t>(c->rowF), ptr<uint64_t>(c->Arow), (uint64_t)c->nreads + 1);   // (rowF[nreads] = 0)
                if (rc) return rc;
                HIPCHK(c, hipMemcpyAsync(&F, ptr<uint64_t>(c->Arow) + c->nreads, 8, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
            }
            size_t mfree = 0, mtotal = 0;
            HIPCHK(c, hipMemGetInfo(&mfree, &mtotal));
            // (released buffers and the reserved slab count as free for this purpose -- the slab by its largest HOLE: the lists are one
            // buffer, and a fragmented slab that cannot give it would send the request to the driver next to the slab's 55 % of VRAM)
            mfree += c->pool.bytes + c->pool.arena.largest_hole();
            // 10 bytes per product for the lists; a pass over all owned columns then needs about 60 more per product (records twice,
            // product lists, scratch, diagnostics) -- the lists are only built when both fit (debug bit 11: tests, "no room")
            const bool fits = (double)F * 70.0 + 1e8 <= (double)mfree && !(c->debug & 2048u);
            if (!rc && fits) {
                HIPCHK(c, hipEventRecord(c->ev[10], c->stream));
                rc = ensure_bytes(c, c->Aent2, 8 * F + 64);
                if (!rc) rc = ensure_bytes(c, c->Aov, 2 * F + 64);
                if (!rc) {
                    const uint32_t grid = c->nreads < 4096u ? c->nreads : 4096u;
                    if (grid) k_layout_rowlists<<<grid, kRowListBlock, 0, c->stream>>>(Bloc, ptr<uint2>(c->Bent), ptr<uint2>(c->Aent), ptr<uint64_t>(c->Arow),
                                                                                     ptr<uint64_t>(c->roff), nullptr, c->nreads, ptr<uint2>(c->Aent2), ptr<uint16_t>(c->Aov), inl);
                    KCHK(c);
                    HIPCHK(c, hipEventRecord(c->ev[11], c->stream));
                    HIPCHK(c, hipEventSynchronize(c->ev[11]));
                    c->tm.expand_ms = ev_ms(c->ev[10], c->ev[11]);
                    c->have_rowlists = true;
                }
            }
            if (!c->have_rowlists) { release(c->Aent2); release(c->Aov); release(c->Arow); c->err.clear(); }
        }
    }
    if (!c->have_rowlists) { release(c->Aent2); release(c->Aov); release(c->Arow); }
    if (!nnz) {
        HIPCHK(c, hipMemsetAsync(c->Bloc.p, 0, 4 * ((size_t)c->nreads + 2), c->stream));
        ENSURE(c, c->rowF, 4 * ((size_t)c->nreads + 2));
        HIPCHK(c, hipMemsetAsync(c->rowF.p, 0, 4 * ((size_t)c->nreads + 2), c->stream));
    }
    HIPCHK(c, hipEventRecord(c->ev[5], c->stream));
    uint32_t st = 0;
    int rc = read_status(c, &st);
    if (rc) return rc;
    rc = status_to_error(c, st);
    if (rc) return rc;
    c->pair_ratio1024 = ratio1024;
    // assembly temporaries are large (tens of bytes per nonzero): give them back
    release(c->lk_key); release(c->lk_key2); release(c->lk_val); release(c->lk_val2); release(c->w); release(c->wscan); release(c->lk_rinfo);
    c->tm.layout_ms = ev_ms(c->ev[4], c->ev[5]);
    c->have_matrix = true;
    c->layout_gen++;
    c->have_pairs = c->have_alns = false;
    return 0;
}

int check_params(bella_ctx* c, const bella_params* p) {
    if (!p) return fail(c, BELLA_ERR_BAD_ARG, "params is NULL");
    if (p->kmer_size != c->kmer_size)
        return fail(c, BELLA_ERR_BAD_ARG, "params.kmer_size (%u) differs from the matrix's (%u)", p->kmer_size, c->kmer_size);
    // (bin_size == 0 is legal: chain.hpp:114 compares |overlap difference| < binSize, so no two products ever share a bin)
    if (!p->skip_alignment) {
        if (p->xdrop > 127) return fail(c, BELLA_ERR_BAD_ARG, "params.xdrop must fit Xavier's int8 scores (<= 127)");
        if (!(p->error_rate >= 0.0 && p->error_rate <= 1.0) || !(p->delta_chernoff >= 0.0 && p->delta_chernoff <= 1.0))
            return fail(c, BELLA_ERR_BAD_ARG, "params.error_rate and params.delta_chernoff must lie in [0, 1]");
    }
    return 0;
}

}  // namespace

extern "C" {

int bella_hip_abi_version(void) { return BELLA_HIP_ABI_VERSION; }

int bella_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* bella_hip_strerror(int code) {
    switch (code) {
        case BELLA_OK: return "ok";
        case BELLA_ERR_NO_DEVICE: return "no gfx950 HIP device";
        case BELLA_ERR_HIP: return "HIP runtime error";
        case BELLA_ERR_BAD_ARG: return "bad argument";
        case BELLA_ERR_BAD_BASE: return "read contains a character other than ACGT";
        case BELLA_ERR_READ_TOO_LONG: return "read too long for 16-bit positions";
        case BELLA_ERR_TUPLE_ORDER: return "tuples not grouped by read";
        case BELLA_ERR_STATE: return "call order violated";
        case BELLA_ERR_ROW_TOO_LARGE: return "output column too large";
        case BELLA_ERR_BINS: return "more than 16 overlap bins";
        case BELLA_ERR_NOMEM: return "out of device memory";
    }
    return "unknown error";
}

const char* bella_hip_last_error(const bella_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int bella_hip_init(int device, bella_ctx** out) {
    if (!out) return BELLA_ERR_BAD_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return BELLA_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return BELLA_ERR_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return BELLA_ERR_NO_DEVICE;
    bella_ctx* c = new bella_ctx();
    c->device = device;
    c->ntiers = 0;
    for (uint32_t t = 0; t < kNumTiers; ++t) { c->tier_caps[t] = kTierCapsDefault[t]; c->ntiers = t + 1; if (kTierCapsDefault[t] == 65535) break; }
    hipError_t he = hipSetDevice(device);
    if (he == hipSuccess) he = hipStreamCreate(&c->stream);
    for (auto& e : c->ev) if (he == hipSuccess) he = hipEventCreate(&e);
    {   // side streams 0 / 1 carry the classes of the largest columns: their workgroups go first whenever a CU has room
        int least = 0, greatest = 0;
        if (he == hipSuccess) he = hipDeviceGetStreamPriorityRange(&least, &greatest);
        int si = 0;
        for (auto& st : c->side) {
            // (the last one carries the host-driven sort-based path of wide.hpp next to the class launches: its small kernels and
            // copies must not queue behind the classes' workgroups)
            if (he == hipSuccess) he = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, si < 2 || si == kNumTiers ? greatest : least);
            ++si;
        }
    }
    if (he == hipSuccess) he = hipEventCreateWithFlags(&c->fork, hipEventDisableTiming);
    for (auto& e : c->join) if (he == hipSuccess) he = hipEventCreateWithFlags(&e, hipEventDisableTiming);
    if (he != hipSuccess) { bella_hip_destroy(c); return BELLA_ERR_HIP; }
    if (hipHostMalloc((void**)&c->pinned, 512, hipHostMallocDefault) != hipSuccess) { delete c; return BELLA_ERR_NOMEM; }
    if (ensure_bytes(c, c->status, 256)) { delete c; return BELLA_ERR_NOMEM; }
    (void)hipMemset(c->status.p, 0, 256);
    {   // the LDS tiers of the row kernels rely on lane-ordered same-address LDS atomics (spgemm.hpp, phase S): checked once per context
        uint32_t bad = 1;
        k_lane_order_selftest<<<1, 64, 0, c->stream>>>(ptr<uint32_t>(c->status) + 12);
        if (hipMemcpyAsync(&bad, ptr<uint32_t>(c->status) + 12, 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess) bad = 1;
        c->lane_order_ok = bad == 0;
        c->tm.lane_order = c->lane_order_ok ? 1u : 2u;
    }
    *out = c;
    return 0;
}

void bella_hip_destroy(bella_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->comm && c->api) { (void)c->api->CommDestroy(c->comm); c->comm = nullptr; }
    release(c->comm_meta);
    Buf* all[] = {&c->packed, &c->roff, &c->Bptr, &c->Bk, &c->Bpos, &c->Bent, &c->rowF, &c->Aent, &c->Aent2, &c->Aov, &c->Arow, &c->Bloc, &c->t_kmer, &c->t_read, &c->t_pos,
                  &c->tstart, &c->Bk_tmp, &c->Bpos_tmp, &c->rowcnt, &c->asm_ws, &c->asm_cls, &c->lk_key, &c->lk_key2, &c->lk_val, &c->lk_val2, &c->lk_rinfo,
                  &c->w, &c->wscan, &c->flopsr, &c->flopptr, &c->nnzC, &c->colptrC,
                  &c->rowlists, &c->tiercaps, &c->tmp_pairs, &c->tmp_ext, &c->pairs, &c->ext, &c->sortscr, &c->ws,
                  &c->status, &c->cubtmp, &c->alns, &c->seeds, &c->plist_hv, &c->overflow, &c->ctl, &c->retry, &c->orderlist, &c->order_ws, &c->w_f, &c->w_off, &c->w_key, &c->w_key2, &c->w_idx, &c->w_idx2, &c->w_hv, &c->w_ovfl,
                  &c->w_plist, &c->w_scr, &c->w_rlen, &c->w_rstart, &c->w_rrank, &c->w_redo, &c->w_segfirst, &c->w_toff, &c->w_table, &c->w_nruns, &c->w_desc, &c->w_gtab, &c->w_gcount, &c->w_gbase, &c->w_rfirst, &c->w_aent2, &c->w_aov, &c->kc_nk, &c->kc_koff, &c->kc_hist, &c->kc_keys, &c->kc_alt, &c->kc_runlen,
                  &c->kc_flag, &c->kc_slot, &c->kc_nruns, &c->kc_dcode, &c->kc_dcount, &c->kc_hkey, &c->kc_hval, &c->kc_found, &c->kc_tstart, &c->kc_cursor, &c->kc_sel, &c->kc_ringo, &c->kc_ringp, &c->kc_opos, &c->kc_oid, &c->kc_opos2, &c->kc_oid2, &c->xest, &c->xest2, &c->xids, &c->xorder, &c->xres, &c->xstate, &c->xlive, &c->lg_res, &c->lg_redo, &c->lg_scratch};
    for (Buf* b : all) release(*b);
    trim_pool(c->pool);
    if (c->pool.arena.base) { (void)hipFree(c->pool.arena.base); c->pool.arena = Arena(); }
    for (auto& e : c->ev) (void)hipEventDestroy(e);
    for (auto& st : c->side) (void)hipStreamDestroy(st);
    (void)hipEventDestroy(c->fork);
    for (auto& e : c->join) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(c->stream);
    if (c->pinned) (void)hipHostFree(c->pinned);
    c->stager.destroy();
    delete c;
}

int bella_hip_set_tuning(bella_ctx* c, uint32_t what, const uint64_t* values, uint32_t n) {
    if (!c || (n && !values)) return BELLA_ERR_BAD_ARG;
    switch (what) {
        case BELLA_TUNE_LDS_TIERS: {                               // ascending product capacities of the LDS tiers, each in [64, 11008]; n = 0: defaults
            uint32_t caps[kNumTiers];
            uint32_t m = 0;
            for (uint32_t q = 0; q < n && m + 1 < kNumTiers; ++q) {
                const uint64_t v = values[q];
                if (v < 64 || v > 11008 || (m && v <= caps[m - 1])) return fail(c, BELLA_ERR_BAD_ARG, "tier caps must ascend within [64, 11008]");
                caps[m++] = (uint32_t)v;
            }
            if (m) {
                for (uint32_t t = 0; t < m; ++t) c->tier_caps[t] = caps[t];
                c->tier_caps[m] = 65535;
                c->ntiers = m + 1;
                c->tiers_from_env = true;
            } else {
                c->ntiers = 0;
                for (uint32_t t = 0; t < kNumTiers; ++t) { c->tier_caps[t] = kTierCapsDefault[t]; c->ntiers = t + 1; if (kTierCapsDefault[t] == 65535) break; }
                c->tiers_from_env = false;
            }
            c->caps_state = 0;
            c->pass_known = false;
            return 0;
        }
        case BELLA_TUNE_KCOUNT_BUDGET: c->kcount_budget = n && values[0] ? values[0] : (1ull << 30); return 0;
        case BELLA_TUNE_WIDE_BUDGET: c->wide_budget = n && values[0] ? values[0] : (1ull << 30); return 0;
        case BELLA_TUNE_XDROP_VARIANT:
            if (n && values[0] > 3) return fail(c, BELLA_ERR_BAD_ARG, "x-drop variant 0..3");   // (validated before it is stored)
            c->xdrop_variant = n ? (uint32_t)values[0] : 1u;
            return 0;
        case BELLA_TUNE_XDROP_CLASS_MIN: c->xdrop_class_min = n ? values[0] : 4ull * 4096 * 64; return 0;
        case BELLA_TUNE_ROW_LISTS:
            if (n && values[0] > 1) return fail(c, BELLA_ERR_BAD_ARG, "row lists: 0 or 1");
            c->want_rowlists = n && values[0] == 1;
            return 0;
        case BELLA_TUNE_LAYOUT_ORDER:
            if (n && values[0] > 1) return fail(c, BELLA_ERR_BAD_ARG, "layout order: 0 (k-mer order) or 1 (first appearance)");
            c->tune_layout_order = n ? (uint32_t)values[0] : 0u;
            return 0;
        case BELLA_TUNE_INLINE_ENTRIES:
            if (n && values[0] > 2) return fail(c, BELLA_ERR_BAD_ARG, "inline entries: 0 (by size), 1 (never) or 2 (always)");
            c->tune_inline = n ? (uint32_t)values[0] : 0u;
            return 0;
        case BELLA_TUNE_ROW_PATH:
            if (n && values[0] > 1) return fail(c, BELLA_ERR_BAD_ARG, "row path: 0 (LDS tiers) or 1 (global workspace)");
            c->tune_row_path = n ? (uint32_t)values[0] : 0u;
            c->pass_known = false;
            return 0;
        case BELLA_TUNE_DIST_LAYOUT:
            if (n && values[0] > 2) return fail(c, BELLA_ERR_BAD_ARG, "distributed layout: 0 (never), 1 (always) or 2 (on the in-process transport only)");
            c->tune_dist_layout = n ? (uint32_t)values[0] : 2u;
            return 0;
        case BELLA_TUNE_CACHE_BYTES:
            c->tune_cache_bytes = n && values[0] ? values[0] : (192ull << 20);
            return 0;
        case BELLA_TUNE_COMPACT_B:
            if (n && values[0] > 1) return fail(c, BELLA_ERR_BAD_ARG, "compact B': 0 (drop the entries without a later read) or 1 (keep them)");
            c->tune_compact_b = n ? (uint32_t)values[0] : 0u;
            return 0;
    }
    return fail(c, BELLA_ERR_BAD_ARG, "unknown tuning parameter %u", what);
}

// Released buffers the context keeps for reuse (BufPool) go back to the driver: for hosts that share the device with other allocators
// (another context, RCCL, PyTorch) and would rather have the memory than the faster next stage.  The reserved slab is not touched.
int bella_hip_trim(bella_ctx* c) {
    if (!c) return BELLA_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    trim_pool(c->pool);
    return 0;
}

// One slab from the driver, up front (see Arena): bytes == 0 gives an existing, unused slab back.  A larger request replaces an unused
// slab; while buffers live in the slab it stays as it is (BELLA_ERR_STATE).  What does not fit the slab later goes to hipMalloc as before.
int bella_hip_reserve(bella_ctx* c, uint64_t bytes, double* ms) {
    if (!c) return BELLA_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    Arena& a = c->pool.arena;
    if (ms) *ms = 0.0;
    if (a.base && a.in_use) return bytes <= a.size ? 0 : fail(c, BELLA_ERR_STATE, "the reserved slab is in use (%zu of %zu bytes)", a.in_use, a.size);
    if (a.base && bytes && bytes <= a.size) return 0;
    if (a.base) { HIPCHK(c, hipDeviceSynchronize()); (void)hipFree(a.base); a = Arena(); }
    if (!bytes) return 0;
    const auto t0 = std::chrono::steady_clock::now();
    void* q = nullptr;
    trim_pool(c->pool);
    const size_t sz = Arena::round((size_t)bytes);
    hipError_t e = hipMalloc(&q, sz);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(c, BELLA_ERR_NOMEM, "hipMalloc(%zu) for the reserved slab failed: %s", sz, hipGetErrorString(e)); }
    // touch it now: the driver maps (and wipes) pages on first use, which is the cost this call exists to take out of the stages
    e = hipMemsetAsync(q, 0, sz, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { (void)hipFree(q); return fail(c, BELLA_ERR_HIP, "reserve: %s", hipGetErrorString(e)); }
    a.base = (char*)q; a.size = sz; a.in_use = 0; a.peak = 0;
    a.holes.clear();
    a.holes.emplace(0, sz);
    if (ms) *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}

int bella_hip_set_debug(bella_ctx* c, uint32_t flags) {
    if (!c) return BELLA_ERR_BAD_ARG;
    c->debug = flags;
    return 0;
}

int bella_hip_set_column_range(bella_ctx* c, uint32_t first, uint32_t count) {
    if (!c) return BELLA_ERR_BAD_ARG;
    c->range_lo = first;
    c->range_hi = (uint64_t)first + count > 0xFFFFFFFFull ? 0xFFFFFFFFu : first + count;
    c->have_pairs = c->have_alns = false;
    return 0;
}

int bella_hip_set_partition(bella_ctx* c, uint32_t first, uint32_t stride) {
    if (!c || stride == 0 || first >= stride) return fail(c, BELLA_ERR_BAD_ARG, "partition needs 0 <= first < stride");
    c->part_first = first;
    c->part_stride = stride;
    c->have_pairs = c->have_alns = false;
    return 0;
}

// the reads onto the device: `fill(pinned, o, n)` produces bytes [o, o + n) of the concatenated bases (Stager::h2d_fill)
extern "C++" {
template <typename Fill>
static int set_reads_impl(bella_ctx* c, const uint64_t* offsets, uint32_t nreads, Fill&& fill) {
    HIPCHK(c, hipSetDevice(c->device));
    const uint64_t total = offsets[nreads];
    for (uint32_t r = 0; r < nreads; ++r) {
        if (offsets[r + 1] < offsets[r]) return fail(c, BELLA_ERR_BAD_ARG, "offsets must be non-decreasing");
        if (offsets[r + 1] - offsets[r] >= 65536ull)
            return fail(c, BELLA_ERR_READ_TOO_LONG, "read %u has %llu bases; positions are 16 bit (common.h:122-126)", r,
                        (unsigned long long)(offsets[r + 1] - offsets[r]));
    }
    const uint64_t nwords = (total + 15) / 16;
    ENSURE(c, c->packed, 4 * (nwords + 16));
    ENSURE(c, c->roff, 8 * ((size_t)nreads + 1));
    Buf raw;
    int rc = ensure_bytes(c, raw, total);
    if (rc) return rc;
    hipError_t e = total ? c->stager.h2d_fill(raw.p, total, c->stream, fill) : hipSuccess;
    if (e == hipSuccess) e = hipMemcpyAsync(c->roff.p, offsets, 8 * ((size_t)nreads + 1), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->packed.p, 0, 4 * (nwords + 16), c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->status.p, 0, 4, c->stream);
    if (e == hipSuccess && nwords) {
        k_pack_reads<<<nblk(nwords), 256, 0, c->stream>>>(ptr<uint8_t>(raw), total, ptr<uint32_t>(c->packed), nwords,
                                                          ptr<uint32_t>(c->status));
        e = hipGetLastError();
    }
    uint32_t st = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&st, c->status.p, 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    release(raw);
    if (e != hipSuccess) return fail(c, BELLA_ERR_HIP, "set_reads: %s", hipGetErrorString(e));
    rc = status_to_error(c, st);
    if (rc) return rc;
    c->nreads = nreads;
    c->total_bases = total;
    c->names.clear();
    c->host_lens.resize(nreads);
    for (uint32_t r = 0; r < nreads; ++r) c->host_lens[r] = (uint32_t)(offsets[r + 1] - offsets[r]);
    c->have_reads = true;
    c->have_matrix = c->have_pairs = c->have_alns = false;
    c->have_tuples = false;
    return 0;
}

}  // extern "C++"

int bella_hip_set_reads(bella_ctx* c, const uint8_t* bases, const uint64_t* offsets, uint32_t nreads) {
    if (!c || !offsets || (nreads && !bases)) return fail(c, BELLA_ERR_BAD_ARG, "null argument");
    return set_reads_impl(c, offsets, nreads, [bases](void* pinned, size_t o, size_t n) { Stager::par_memcpy(pinned, bases + o, n); });
}

// ---- FASTQ ingest (fastq.hpp) ---------------------------------------------------------------------------------------------
// The file is mapped and indexed on all cores; the bases go mapping -> pinned buffer -> device one 64 MB chunk at a time, the
// gather of chunk i+1 overlapping the transfer of chunk i.  The concatenated bases never exist on the host.
// One file or several: the reference's -f names a LIST of FASTQ files whose reads are numbered through in list order
// (kmercount.hpp:82-105 GetFiles, main.cpp:339-423).  Every file is mapped and indexed for itself; the base stream of the set is the
// files' streams back to back, and a chunk that spans a file boundary is gathered from both sides.
int bella_hip_load_fastq_list(bella_ctx* c, const char* const* paths, uint32_t nfiles, uint32_t* nreads, uint64_t* nbases) {
    if (!c || (nfiles && !paths)) return fail(c, BELLA_ERR_BAD_ARG, "null argument");
    std::vector<FastqIndex> ix(nfiles);
    std::vector<uint64_t> first(nfiles + 1, 0);                 // first[f]: where file f's bases start in the set's base stream
    std::vector<uint64_t> offsets(1, 0);
    std::string err;
    const auto t0 = std::chrono::steady_clock::now();
    uint64_t file_bytes = 0, total_reads = 0;
    unsigned threads = 1;
    for (uint32_t f = 0; f < nfiles; ++f) {
        if (!paths[f]) return fail(c, BELLA_ERR_BAD_ARG, "null path");
        if (ix[f].build(paths[f], err)) return fail(c, BELLA_ERR_BAD_ARG, "%s", err.c_str());
        total_reads += ix[f].names.size();
        if (total_reads >= 0xFFFFFFF0ull) return fail(c, BELLA_ERR_BAD_ARG, "more than 2^32 reads");
        first[f + 1] = first[f] + ix[f].nbases();
        for (uint32_t r = 0; r < ix[f].nreads(); ++r) offsets.push_back(first[f] + ix[f].offsets[r + 1]);
        file_bytes += (uint64_t)ix[f].file.n;
        threads = std::max(threads, ix[f].threads);
    }
    const auto t1 = std::chrono::steady_clock::now();
    int rc = set_reads_impl(c, offsets.data(), (uint32_t)total_reads, [&](void* pinned, size_t o, size_t n) {
        uint32_t f = (uint32_t)(std::upper_bound(first.begin(), first.end(), (uint64_t)o) - first.begin()) - 1;
        uint8_t* dst = (uint8_t*)pinned;
        while (n) {
            while (f + 1 < nfiles && first[f + 1] <= o) ++f;    // (files without bases)
            const size_t take = (size_t)std::min<uint64_t>(n, first[f + 1] - o);
            ix[f].gather_parallel(dst, o - first[f], take);
            dst += take; o += take; n -= take;
        }
    });
    if (rc) return rc;
    const auto t2 = std::chrono::steady_clock::now();
    c->ingest = {file_bytes, c->total_bases, c->nreads, threads, std::chrono::duration<double, std::milli>(t1 - t0).count(),
                 std::chrono::duration<double, std::milli>(t2 - t1).count()};
    c->names.clear();
    c->names.reserve(total_reads);
    for (uint32_t f = 0; f < nfiles; ++f)
        for (auto& nm : ix[f].names) c->names.push_back(std::move(nm));
    if (nreads) *nreads = c->nreads;
    if (nbases) *nbases = c->total_bases;
    return 0;
}

int bella_hip_load_fastq(bella_ctx* c, const char* path, uint32_t* nreads, uint64_t* nbases) {
    if (!c || !path) return fail(c, BELLA_ERR_BAD_ARG, "null argument");
    return bella_hip_load_fastq_list(c, &path, 1, nreads, nbases);
}

int bella_hip_get_ingest_stats(bella_ctx* c, bella_ingest_stats* out) {
    if (!c || !out) return BELLA_ERR_BAD_ARG;
    *out = c->ingest;
    return 0;
}

int bella_hip_get_read_names(bella_ctx* c, char* buf, uint64_t buflen, uint64_t* offsets, uint64_t* needed) {
    if (!c) return BELLA_ERR_BAD_ARG;
    if (!c->have_reads || c->names.size() != c->nreads) return fail(c, BELLA_ERR_STATE, "the reads were not loaded with bella_hip_load_fastq");
    uint64_t tot = 0;
    for (const auto& n : c->names) tot += n.size() + 1;
    if (needed) *needed = tot;
    if (!buf) return 0;
    if (buflen < tot) return fail(c, BELLA_ERR_BAD_ARG, "name buffer too small: %llu < %llu", (unsigned long long)buflen, (unsigned long long)tot);
    uint64_t o = 0;
    for (size_t r = 0; r < c->names.size(); ++r) {
        if (offsets) offsets[r] = o;
        std::memcpy(buf + o, c->names[r].c_str(), c->names[r].size() + 1);          // NUL-terminated, back to back
        o += c->names[r].size() + 1;
    }
    if (offsets) offsets[c->names.size()] = o;
    return 0;
}

int bella_hip_get_read_lengths(bella_ctx* c, uint32_t* lens) {
    if (!c || !lens) return BELLA_ERR_BAD_ARG;
    if (!c->have_reads) return fail(c, BELLA_ERR_STATE, "no reads");
    std::memcpy(lens, c->host_lens.data(), 4 * (size_t)c->nreads);
    return 0;
}

int bella_hip_set_B(bella_ctx* c, uint16_t kmer_size, uint32_t nkmers, const uint32_t* colptr, const uint32_t* rowids,
                    const uint16_t* values) {
    if (!c || !colptr) return fail(c, BELLA_ERR_BAD_ARG, "null argument");
    if (!c->have_reads) return fail(c, BELLA_ERR_STATE, "set_reads first");
    if (kmer_size < 1 || kmer_size > 32) return fail(c, BELLA_ERR_BAD_ARG, "k must be in [1,32] (MAX_KMER_SIZEK, kmercode/common.h:13)");
    HIPCHK(c, hipSetDevice(c->device));
    c->have_matrix = c->have_pairs = c->have_alns = false;
    c->have_panel = false;
    c->have_tuples = false;
    if (colptr[0] != 0) return fail(c, BELLA_ERR_BAD_ARG, "colptr[0] must be 0");
    for (uint32_t r = 0; r < c->nreads; ++r) {
        if (colptr[r + 1] < colptr[r]) return fail(c, BELLA_ERR_BAD_ARG, "colptr must be non-decreasing (column %u)", r);
        if (colptr[r + 1] - colptr[r] >= 65536u) return fail(c, BELLA_ERR_READ_TOO_LONG, "column %u has >= 65536 entries", r);
    }
    const uint64_t nnz = colptr[c->nreads];
    if (nnz && (!rowids || !values)) return fail(c, BELLA_ERR_BAD_ARG, "null argument");
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    ENSURE(c, c->Bptr, 4 * ((size_t)c->nreads + 1));
    ENSURE(c, c->Bk, 4 * nnz);
    ENSURE(c, c->Bpos, 2 * nnz);
    HIPCHK(c, hipMemcpyAsync(c->Bptr.p, colptr, 4 * ((size_t)c->nreads + 1), hipMemcpyHostToDevice, c->stream));
    if (nnz) {
        HIPCHK(c, c->stager.h2d(c->Bk.p, rowids, 4 * nnz, c->stream));
        HIPCHK(c, c->stager.h2d(c->Bpos.p, values, 2 * nnz, c->stream));
    }
    c->nkmers = nkmers;
    c->nnz = nnz;
    c->kmer_size = kmer_size;
    int rc = build_layout(c);
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev[1]));
    c->tm.assemble_ms = ev_ms(c->ev[0], c->ev[1]);
    return 0;
}

// tuples of the reads first .. first+nr-1 -> their rows of B (CSR: c->Bptr local offsets, c->Bk, c->Bpos), on device
// t_kmer == nullptr: the tuples are already on the device (bella_hip_count_kmers), `toff` tuples into the buffers
static int assemble_rows_device(bella_ctx* c, uint32_t first, uint32_t nr, uint64_t ntuples, const uint32_t* t_kmer,
                                const uint32_t* t_read, const uint16_t* t_pos, uint64_t* nnz_out, uint64_t toff = 0) {
    release_count_scratch(c);
    if (t_kmer || !ntuples) {
        ENSURE(c, c->t_kmer, 4 * ntuples);
        ENSURE(c, c->t_read, 4 * ntuples);
        ENSURE(c, c->t_pos, 2 * ntuples);
    }
    ENSURE(c, c->tstart, 8 * ((size_t)nr + 2));
    ENSURE(c, c->Bk_tmp, 4 * ntuples);
    ENSURE(c, c->Bpos_tmp, 2 * ntuples);
    ENSURE(c, c->rowcnt, 4 * ((size_t)nr + 2));
    ENSURE(c, c->Bptr, 4 * ((size_t)nr + 2));
    const uint64_t ws_stride = ((uint64_t)8 * (asm_dedup_slots(65536) + 1) + (uint64_t)6 * 65536 + 255) & ~(uint64_t)255;
    ENSURE(c, c->asm_cls, 4 * ((size_t)kAsmClasses * (nr + 1) + 16));
    if (ntuples && t_kmer) {                          // nullptr: the tuples are already there (bella_hip_count_kmers)
        HIPCHK(c, c->stager.h2d(c->t_kmer.p, t_kmer, 4 * ntuples, c->stream));
        HIPCHK(c, c->stager.h2d(c->t_read.p, t_read, 4 * ntuples, c->stream));
        HIPCHK(c, c->stager.h2d(c->t_pos.p, t_pos, 2 * ntuples, c->stream));
    }
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    HIPCHK(c, hipMemsetAsync(c->status.p, 0, 4, c->stream));
    HIPCHK(c, hipMemsetAsync(c->rowcnt.p, 0, 4 * ((size_t)nr + 2), c->stream));
    uint32_t* const cls_lists = ptr<uint32_t>(c->asm_cls);
    uint32_t* const cls_counts = cls_lists + (size_t)kAsmClasses * (nr + 1);
    HIPCHK(c, hipMemsetAsync(cls_counts, 0, 4 * 8, c->stream));
    k_tuple_bounds<<<nblk(ntuples + 1), 256, 0, c->stream>>>(ptr<uint32_t>(c->t_read) + toff, ntuples, first, nr, ptr<uint64_t>(c->tstart),
                                                             ptr<uint32_t>(c->status));
    KCHK(c);
    if (nr) {
        k_asm_classify<<<nblk(nr), 256, 0, c->stream>>>(ptr<uint64_t>(c->tstart), nr, cls_lists, cls_counts, ptr<uint32_t>(c->rowcnt), ptr<uint32_t>(c->status));
        KCHK(c);
    }
    uint32_t st = 0;
    uint32_t ccount[8] = {};
    HIPCHK(c, hipMemcpyAsync(ccount, cls_counts, sizeof(ccount), hipMemcpyDeviceToHost, c->stream));
    int rc = read_status(c, &st);
    if (rc) return rc;
    rc = status_to_error(c, st);
    if (rc) return rc;
    AsmArgs a;
    a.t_kmer = ptr<uint32_t>(c->t_kmer) + toff;
    a.t_pos = ptr<uint16_t>(c->t_pos) + toff;
    a.tstart = ptr<uint64_t>(c->tstart);
    a.nreads = nr;
    a.Bk_tmp = ptr<uint32_t>(c->Bk_tmp);
    a.Bpos_tmp = ptr<uint16_t>(c->Bpos_tmp);
    a.rowcnt = ptr<uint32_t>(c->rowcnt);
    a.ws_stride = ws_stride;
    a.status = ptr<uint32_t>(c->status);
#ifdef BELLA_ASM_CLOCK
    static unsigned long long* d_aclk = nullptr;
    if (!d_aclk) (void)hipMalloc(&d_aclk, 64 * 8);
    (void)hipMemsetAsync(d_aclk, 0, 64 * 8, c->stream);
    a.clk = d_aclk;
#endif
    {
        // one launch per LDS class (table sizes <= 1024, 2048, 4096, 8192 slots: 256, 512, 1024, 1024 threads), then the global tables
        struct Cls { uint32_t hi; int blk; void (*kern)(AsmArgs); };
        const Cls cls[4] = {{1024u, 256, k_asm_rows_lds<256>}, {2048u, 512, k_asm_rows_lds<512>}, {4096u, 1024, k_asm_rows_lds<1024>},
                            {8192u, 1024, k_asm_rows_lds<1024>}};
        for (int q = 0; q < 4; ++q) {
            if (!ccount[q]) continue;
            HIPCHK(c, hipFuncSetAttribute((const void*)cls[q].kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)asm_lds_bytes(kAsmLdsSlots)));
            a.list = cls_lists + (size_t)q * nr; a.nlist = ccount[q];
#ifdef BELLA_ASM_CLOCK
            a.clk = d_aclk + 8 * q;
#endif
            cls[q].kern<<<ccount[q], cls[q].blk, asm_lds_bytes(cls[q].hi), c->stream>>>(a);
            KCHK(c);
        }
        if (ccount[4]) {
            ENSURE(c, c->asm_ws, ws_stride * kAsmGrid);
            a.ws = ptr<uint8_t>(c->asm_ws);
            a.list = cls_lists + (size_t)4 * nr; a.nlist = ccount[4];
            k_asm_rows_global<<<ccount[4] < kAsmGrid ? ccount[4] : kAsmGrid, 1024, 0, c->stream>>>(a);
            KCHK(c);
        }
    }
#ifdef BELLA_ASM_CLOCK
    {
        unsigned long long hh[64];
        (void)hipMemcpyAsync(hh, d_aclk, 64 * 8, hipMemcpyDeviceToHost, c->stream);
        (void)hipStreamSynchronize(c->stream);
        for (int q = 0; q < 4; ++q) {
            const unsigned long long* h = hh + 8 * q;
            unsigned long long tot = 0;
            for (int i = 0; i < 6; ++i) tot += h[i];
            if (!tot) continue;
            std::fprintf(stderr, "[asm clock] class %d reads %u:", q, ccount[q]);
            const char* nm[6] = {"head+init", "insert", "dup", "rounds", "scan", "emit"};
            for (int i = 0; i < 6; ++i) std::fprintf(stderr, " %s %.1f%%", nm[i], 100.0 * (double)h[i] / (double)tot);
            std::fprintf(stderr, " | cycles per read %.0f\n", (double)tot / ccount[q]);
        }
    }
#endif
    rc = scan_u32(c, ptr<uint32_t>(c->rowcnt), ptr<uint32_t>(c->Bptr), (uint64_t)nr + 1);
    if (rc) return rc;
    uint32_t nnz32 = 0;
    HIPCHK(c, hipMemcpyAsync(&nnz32, ptr<uint32_t>(c->Bptr) + nr, 4, hipMemcpyDeviceToHost, c->stream));
    rc = read_status(c, &st);
    if (rc) return rc;
    rc = status_to_error(c, st);
    if (rc) return rc;
    const uint64_t nnz = nnz32;
    ENSURE(c, c->Bk, 4 * nnz);
    ENSURE(c, c->Bpos, 2 * nnz);
    if (nr) {
        k_compact_B<<<nblk(nr, kWaves), kBlock, 0, c->stream>>>(ptr<uint64_t>(c->tstart), ptr<uint32_t>(c->Bptr), nr,
                                                                ptr<uint32_t>(c->Bk_tmp), ptr<uint16_t>(c->Bpos_tmp),
                                                                ptr<uint32_t>(c->Bk), ptr<uint16_t>(c->Bpos));
        KCHK(c);
    }
    HIPCHK(c, hipEventRecord(c->ev[6], c->stream));            // rows_ms = ev[0] .. ev[6]
    *nnz_out = nnz;
    return 0;
}

int bella_hip_assemble_tuples(bella_ctx* c, uint16_t kmer_size, uint32_t nkmers, uint64_t ntuples, const uint32_t* t_kmer,
                              const uint32_t* t_read, const uint16_t* t_pos) {
    if (!c || (ntuples && (!t_kmer || !t_read || !t_pos))) return fail(c, BELLA_ERR_BAD_ARG, "null argument");
    if (!c->have_reads) return fail(c, BELLA_ERR_STATE, "set_reads first");
    if (kmer_size < 1 || kmer_size > 32) return fail(c, BELLA_ERR_BAD_ARG, "k must be in [1,32]");
    if (ntuples >= 0xFFFFFFF0ull) return fail(c, BELLA_ERR_BAD_ARG, "tuple count must be < 2^32");
    HIPCHK(c, hipSetDevice(c->device));
    c->have_matrix = c->have_pairs = c->have_alns = false;
    c->have_panel = false;
    c->have_tuples = false;
    uint64_t nnz = 0;
    int rc = assemble_rows_device(c, 0, c->nreads, ntuples, t_kmer, t_read, t_pos, &nnz);
    if (rc) return rc;
    c->nkmers = nkmers;
    c->nnz = nnz;
    c->kmer_size = kmer_size;
    rc = build_layout(c);
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev[1]));
    c->tm.assemble_ms = ev_ms(c->ev[0], c->ev[1]);
    c->tm.rows_ms = ev_ms(c->ev[0], c->ev[6]);
    release(c->t_kmer); release(c->t_read); release(c->t_pos); release(c->Bk_tmp); release(c->Bpos_tmp); release(c->asm_ws);
    return 0;
}

// ---- k-mer counting, dictionary, tuples (kcount.hpp) ----------------------------------------------------------------------
static int grow_keep(bella_ctx* c, Buf& b, size_t need, size_t used) {
    if (b.cap >= need) return 0;
    Buf nb;
    int rc = ensure_bytes(c, nb, need + need / 4);
    if (rc) return rc;
    if (used) HIPCHK(c, hipMemcpyAsync(nb.p, b.p, used, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    release(b);
    b = nb;
    return 0;
}

// dist: collective over the context's communicator -- rank r counts the canonical words of ITS range of histogram bins over all
// reads (1/N of the sort), the partial dictionaries (ascending, so their concatenation in rank order is the whole ascending
// dictionary) are exchanged with one grouped send/recv, and tuples are generated for the reads bfirst .. bfirst + brows - 1 only
static int comm_agree(bella_ctx* c, int local_rc);
static int comm_sync(bella_ctx* c, const char* what);

static int count_kmers_impl(bella_ctx* c, uint16_t kmer_size, uint32_t lower, uint32_t upper, uint32_t mode, uint32_t window,
                            uint32_t* nkmers_out, uint64_t* ntuples_out, uint64_t* ndistinct_out, bool dist = false, uint32_t bfirst = 0,
                            uint32_t brows = 0xFFFFFFFFu) {
    if (!c) return BELLA_ERR_BAD_ARG;
    if (!c->have_reads) return fail(c, BELLA_ERR_STATE, "set_reads first");
    if (kmer_size < 1 || kmer_size > 32) return fail(c, BELLA_ERR_BAD_ARG, "k must be in [1,32]");
    if (lower < 2 || upper < lower || upper > 65535) return fail(c, BELLA_ERR_BAD_ARG, "need 2 <= lower <= upper <= 65535");
    if (mode == 2 && (window < 1 || window > 65535)) return fail(c, BELLA_ERR_BAD_ARG, "minimizer window must be in [1,65535]");
    if (mode == 1 && kmer_size <= kSmerLen) return fail(c, BELLA_ERR_BAD_ARG, "syncmer selection needs k > 5 (smerlen, syncmer.hpp:45)");
    HIPCHK(c, hipSetDevice(c->device));
    c->have_tuples = false;
    const uint32_t nr = c->nreads, k = kmer_size;
    static const int dev_marks = getenv("BELLA_DEV_KCMARKS") ? atoi(getenv("BELLA_DEV_KCMARKS")) : 0;   // 1: stage times (synchronising), 2: host time only
    auto tmark0 = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!dev_marks) return;
        if (dev_marks == 1) (void)hipStreamSynchronize(c->stream);
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[kc] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - tmark0).count());
        tmark0 = t;
    };
    const int NR = dist ? c->comm_ranks : 1, me = dist ? c->comm_rank : 0;
    if (dist && !c->comm) return fail(c, BELLA_ERR_STATE, "bella_hip_comm_init first");
    if (brows == 0xFFFFFFFFu) { bfirst = 0; brows = nr; }
    if ((uint64_t)bfirst + brows > nr) return fail(c, BELLA_ERR_BAD_ARG, "read block exceeds the read set");
    // Local phase (histogram, sort passes, partial dictionary): a rank that fails here still takes part in the status exchange below,
    // so that all ranks leave the call together instead of waiting for it in the dictionary exchange.
    const uint8_t* d_sel = nullptr;
    uint64_t nk_total = 0, ndistinct = 0, nocc = 0;                // reliable k-mers, distinct k-mers, occurrences of reliable k-mers so far
    uint32_t *occ_pos = nullptr, *occ_id = nullptr, *occ_pos2 = nullptr, *occ_id2 = nullptr;   // the list of occurrences and what it is sorted against
    bool fast = false, one_pass_all = false;
    uint32_t pb = 0;
    int rc = [&]() -> int {
    const unsigned rgrid = nr < 16384u ? (nr ? nr : 1u) : 16384u;
    ENSURE(c, c->kc_nk, 4 * ((size_t)nr + 2));
    ENSURE(c, c->kc_koff, 8 * ((size_t)nr + 2));
    ENSURE(c, c->kc_hist, 8 * kCountBins);
    ENSURE(c, c->kc_found, 4 * ((size_t)nr + 2));
    ENSURE(c, c->kc_tstart, 8 * ((size_t)nr + 2));
    ENSURE(c, c->kc_nruns, 16);
    ENSURE(c, c->kc_cursor, 16);
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    k_kmers_per_read<<<nblk((uint64_t)nr + 1), 256, 0, c->stream>>>(ptr<uint64_t>(c->roff), nr, k, ptr<uint32_t>(c->kc_nk));
    KCHK(c);
    int rc = scan_u32_to_u64(c, ptr<uint32_t>(c->kc_nk), ptr<uint64_t>(c->kc_koff), (uint64_t)nr + 1);
    if (rc) return rc;
    // ntot = koff[nr] is needed before the histogram in minimizer mode (the selection flags are per position)
    uint64_t ntot = 0;
    HIPCHK(c, hipMemcpyAsync(&ntot, ptr<uint64_t>(c->kc_koff) + nr, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (mode == 2) {
        const uint32_t cap = window + 2;
        ENSURE(c, c->kc_sel, ntot + 16);
        ENSURE(c, c->kc_ringo, 8 * (size_t)cap * (nr ? nr : 1));
        ENSURE(c, c->kc_ringp, 4 * (size_t)cap * (nr ? nr : 1));
        HIPCHK(c, hipMemsetAsync(c->kc_sel.p, 0, ntot + 16, c->stream));
        k_minimizer_select<<<nblk(nr ? nr : 1, 64), 64, 0, c->stream>>>(ptr<uint32_t>(c->packed), ptr<uint64_t>(c->roff), ptr<uint32_t>(c->kc_nk),
                                                                    ptr<uint64_t>(c->kc_koff), nr, k, window, cap, ptr<uint64_t>(c->kc_ringo),
                                                                    ptr<uint32_t>(c->kc_ringp), ptr<uint8_t>(c->kc_sel));
        KCHK(c);
        d_sel = ptr<uint8_t>(c->kc_sel);
        release(c->kc_ringo); release(c->kc_ringp);
    }
    // Words with their positions (kcount.hpp): when the word's 2k bits and a global position index fit 64 bits together -- and the
    // positions that make tuples are the positions that are counted (not so in syncmer mode) -- every sorted word still knows where it
    // came from and the reliable ones leave (position, id) on a list that one more sort turns into the tuples: no hash table over the
    // dictionary, no look-up per position (which was a third of this call).  Not across ranks (a rank only sorts its share of the words).
    {
        int posbits = 1;
        while (posbits < 63 && (1ull << posbits) < ntot) ++posbits;
        fast = NR == 1 && mode != 1 && 2 * (int)k + posbits <= 64 && posbits <= 32 && !(c->debug & 16384u);   // debug bit 14: tests, the look-up path
        pb = fast ? (uint32_t)posbits : 0u;
    }
    uint64_t budget = c->kcount_budget;
    if (budget > 0x7FFF0000ull) budget = 0x7FFF0000ull;               // rocPRIM item counts are 32-bit
    uint64_t hist[kCountBins] = {};
    if (mode == 0 && NR == 1 && ntot <= budget) {
        hist[0] = ntot;                                                // one pass over every position: no histogram needed to size it
        one_pass_all = true;
    } else {
    HIPCHK(c, hipMemsetAsync(c->kc_hist.p, 0, 8 * kCountBins, c->stream));
    k_code_hist<<<rgrid, kBlock, 0, c->stream>>>(ptr<uint32_t>(c->packed), ptr<uint64_t>(c->roff), ptr<uint32_t>(c->kc_nk), nr, k, mode,
                                                 ptr<uint64_t>(c->kc_koff), d_sel, (unsigned long long*)c->kc_hist.p);
    KCHK(c);
    HIPCHK(c, hipMemcpyAsync(hist, c->kc_hist.p, sizeof(hist), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    // passes = runs of consecutive bins holding at most `budget` k-mers (28 bytes of HBM per k-mer in flight)
    // this rank's bins: consecutive, balanced by their word counts (every rank derives the same split from the same histogram)
    uint32_t bin_lo = 0, bin_hi = kCountBins;
    if (NR > 1) {
        uint64_t tot = 0;
        for (uint32_t b = 0; b < kCountBins; ++b) tot += hist[b];
        auto cut = [&](int r) -> uint32_t {                       // first bin of rank r
            if (r <= 0) return 0u;
            if (r >= NR) return kCountBins;
            const uint64_t want = tot / (uint64_t)NR * (uint64_t)r;
            uint64_t acc = 0;
            uint32_t b = 0;
            while (b < kCountBins && acc + hist[b] / 2 < want) acc += hist[b++];
            return b;
        };
        bin_lo = cut(me);
        bin_hi = cut(me + 1);
    }
    std::vector<uint32_t> pass_lo, pass_hi;
    std::vector<uint64_t> pass_n;
    if (one_pass_all) { pass_lo.push_back(0); pass_hi.push_back(kCountBins); pass_n.push_back(ntot); }
    else
    for (uint32_t b = bin_lo; b < bin_hi;) {
        uint64_t n = hist[b];
        if (n > 0x7FFF0000ull) return fail(c, BELLA_ERR_NOMEM, "k-mer counting: one bin of canonical k-mers holds %llu words", (unsigned long long)n);
        uint32_t e = b + 1;
        while (e < bin_hi && n + hist[e] <= budget) n += hist[e++];
        pass_lo.push_back(b); pass_hi.push_back(e); pass_n.push_back(n);
        b = e;
    }
    const bool single = pass_n.size() == 1 && mode == 0 && NR == 1;    // every position contributes: word j of read r has a fixed place
    for (size_t p = 0; p < pass_n.size(); ++p) {
        const uint64_t np = pass_n[p];
        if (!np) continue;
        ENSURE(c, c->kc_keys, 8 * np);
        ENSURE(c, c->kc_alt, 8 * np);
        ENSURE(c, c->kc_runlen, 4 * (np + 64));                    // (fast path: per-tile arrays, two of ntile + 8 words each -- np / 2048 tiles)
        ENSURE(c, c->kc_flag, 4 * (np + 64));
        ENSURE(c, c->kc_slot, 4 * (np + 64));
        if (!single) HIPCHK(c, hipMemsetAsync(c->kc_cursor.p, 0, 8, c->stream));
        k_emit_codes<<<rgrid, kBlock, 0, c->stream>>>(ptr<uint32_t>(c->packed), ptr<uint64_t>(c->roff), ptr<uint32_t>(c->kc_nk),
                                                      ptr<uint64_t>(c->kc_koff), nr, k, mode, pass_lo[p], pass_hi[p], d_sel, ptr<uint64_t>(c->kc_keys),
                                                      single ? nullptr : (unsigned long long*)c->kc_cursor.p, pb);
        KCHK(c);
        mark("alloc + emit codes");
        hipcub::DoubleBuffer<uint64_t> db(ptr<uint64_t>(c->kc_keys), ptr<uint64_t>(c->kc_alt));
        size_t tb = 0;
        // (with the positions in the keys the run kernels handle groups of four neighbouring words: the sort may leave out the word's
        // two lowest bits, and does when that saves it a pass of eight bits -- 32 instead of 34 bits for k = 17)
        const uint32_t gb = fast && 2 * k >= 10 && (2 * k + 7) / 8 > (2 * k - 2 + 7) / 8 ? 2u : 0u;
        HIPCHK(c, hipcub::DeviceRadixSort::SortKeys(nullptr, tb, db, (int)np, (int)(pb + gb), (int)pb + 2 * (int)k, c->stream));
        ENSURE(c, c->cubtmp, tb);
        HIPCHK(c, hipcub::DeviceRadixSort::SortKeys(c->cubtmp.p, tb, db, (int)np, (int)(pb + gb), (int)pb + 2 * (int)k, c->stream));
        uint64_t* sorted = db.Current();
        mark("sort");
        if (fast) {
            // runs of equal words, tile by tile: reliable runs per tile -> scan over the tiles = the ids of this pass -> every reliable
            // word's id to its position, the dictionary entries from the runs' first words (kcount.hpp)
            const unsigned ntile = nblk(np, kRunTile);
            uint32_t* tile_rel = ptr<uint32_t>(c->kc_runlen);       // (ntile + 1 each; 4 (np + 64) bytes are there)
            uint32_t* tile_words = tile_rel + ntile + 8;
            uint32_t* tile_heads = ptr<uint32_t>(c->kc_flag);
            uint32_t* tile_base = ptr<uint32_t>(c->kc_slot);
            uint32_t* tile_wbase = tile_base + ntile + 8;
            HIPCHK(c, hipMemsetAsync(tile_rel + ntile, 0, 4, c->stream));
            HIPCHK(c, hipMemsetAsync(tile_words + ntile, 0, 4, c->stream));
            k_run_count<<<ntile, kRunBlock, 0, c->stream>>>(sorted, np, pb, gb, lower, upper, mode != 0, tile_rel, tile_heads, tile_words);
            KCHK(c);
            rc = scan_u32(c, tile_rel, tile_base, (uint64_t)ntile + 1);
            if (rc) return rc;
            rc = scan_u32(c, tile_words, tile_wbase, (uint64_t)ntile + 1);
            if (rc) return rc;
            {
                hipcub::TransformInputIterator<uint64_t, CastU64, const uint32_t*> hit(tile_heads, CastU64());
                size_t tb4 = 0;
                HIPCHK(c, hipcub::DeviceReduce::Sum(nullptr, tb4, hit, ptr<uint64_t>(c->kc_cursor), (int)ntile, c->stream));
                ENSURE(c, c->cubtmp, tb4);
                HIPCHK(c, hipcub::DeviceReduce::Sum(c->cubtmp.p, tb4, hit, ptr<uint64_t>(c->kc_cursor), (int)ntile, c->stream));
            }
            uint32_t nrel = 0, nwords = 0;
            uint64_t nruns = 0;
            HIPCHK(c, hipMemcpyAsync(&nrel, tile_base + ntile, 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipMemcpyAsync(&nwords, tile_wbase + ntile, 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipMemcpyAsync(&nruns, c->kc_cursor.p, 8, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            mark("run count + scans");
            if (nk_total + nrel >= 0xFFFFFFF0ull) return fail(c, BELLA_ERR_BAD_ARG, "more than 2^32 reliable k-mers");
            if (nocc + nwords >= 0xFFFFFFF0ull) return fail(c, BELLA_ERR_BAD_ARG, "tuple count must be < 2^32");
            rc = grow_keep(c, c->kc_dcode, 8 * (nk_total + nrel), 8 * nk_total);
            if (rc) return rc;
            rc = grow_keep(c, c->kc_dcount, 2 * (nk_total + nrel), 2 * nk_total);
            if (rc) return rc;
            // the list of occurrences: with one pass (the usual case) it goes into the sort's free buffer (8 np bytes: positions, then
            // ids) and is later sorted against the buffer of the sorted words, which nobody reads after this kernel -- no allocation;
            // with several passes it grows in buffers of its own
            uint32_t *op = nullptr, *oi = nullptr;
            if (pass_n.size() == 1) {
                uint64_t* freebuf = sorted == ptr<uint64_t>(c->kc_keys) ? ptr<uint64_t>(c->kc_alt) : ptr<uint64_t>(c->kc_keys);
                op = (uint32_t*)freebuf; oi = op + np;
                occ_pos = op; occ_id = oi; occ_pos2 = (uint32_t*)sorted; occ_id2 = occ_pos2 + np;
            } else {
                rc = grow_keep(c, c->kc_opos, 4 * (nocc + nwords) + 16, 4 * nocc);
                if (rc) return rc;
                rc = grow_keep(c, c->kc_oid, 4 * (nocc + nwords) + 16, 4 * nocc);
                if (rc) return rc;
                op = ptr<uint32_t>(c->kc_opos) + nocc; oi = ptr<uint32_t>(c->kc_oid) + nocc;
                occ_pos = ptr<uint32_t>(c->kc_opos); occ_id = ptr<uint32_t>(c->kc_oid); occ_pos2 = nullptr; occ_id2 = nullptr;
            }
            mark("grow dict + list");
            k_run_assign<<<ntile, kRunBlock, 0, c->stream>>>(sorted, np, pb, gb, lower, upper, mode != 0, tile_base, tile_wbase, (uint32_t)nk_total, op, oi,
                                                             ptr<uint64_t>(c->kc_dcode) + nk_total, ptr<uint16_t>(c->kc_dcount) + nk_total);
            KCHK(c);
            mark("run assign");
            nocc += nwords;
            nk_total += nrel;
            ndistinct += nruns;
            continue;
        }
        uint64_t* run_code = sorted == ptr<uint64_t>(c->kc_keys) ? ptr<uint64_t>(c->kc_alt) : ptr<uint64_t>(c->kc_keys);
        size_t tb2 = 0;
        HIPCHK(c, hipcub::DeviceRunLengthEncode::Encode(nullptr, tb2, sorted, run_code, ptr<uint32_t>(c->kc_runlen),
                                                       ptr<uint32_t>(c->kc_nruns), (int)np, c->stream));
        ENSURE(c, c->cubtmp, tb2);
        HIPCHK(c, hipcub::DeviceRunLengthEncode::Encode(c->cubtmp.p, tb2, sorted, run_code, ptr<uint32_t>(c->kc_runlen),
                                                       ptr<uint32_t>(c->kc_nruns), (int)np, c->stream));
        uint32_t nruns = 0;
        HIPCHK(c, hipMemcpyAsync(&nruns, c->kc_nruns.p, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        k_flag_reliable<<<nblk((uint64_t)nruns + 1), 256, 0, c->stream>>>(ptr<uint32_t>(c->kc_runlen), nruns, lower, upper, mode != 0, ptr<uint32_t>(c->kc_flag));
        KCHK(c);
        rc = scan_u32(c, ptr<uint32_t>(c->kc_flag), ptr<uint32_t>(c->kc_slot), (uint64_t)nruns + 1);
        if (rc) return rc;
        uint32_t nrel = 0;
        HIPCHK(c, hipMemcpyAsync(&nrel, ptr<uint32_t>(c->kc_slot) + nruns, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (nk_total + nrel >= 0xFFFFFFF0ull) return fail(c, BELLA_ERR_BAD_ARG, "more than 2^32 reliable k-mers");
        rc = grow_keep(c, c->kc_dcode, 8 * (nk_total + nrel), 8 * nk_total);
        if (rc) return rc;
        rc = grow_keep(c, c->kc_dcount, 2 * (nk_total + nrel), 2 * nk_total);
        if (rc) return rc;
        if (nruns) {
            k_write_dict<<<nblk(nruns), 256, 0, c->stream>>>(run_code, ptr<uint32_t>(c->kc_runlen), ptr<uint32_t>(c->kc_flag), ptr<uint32_t>(c->kc_slot),
                                                             nruns, mode != 0, ptr<uint64_t>(c->kc_dcode) + nk_total, ptr<uint16_t>(c->kc_dcount) + nk_total);
            KCHK(c);
        }
        nk_total += nrel;
        ndistinct += nruns;
    }
    return 0;
    }();
    if (NR > 1) rc = comm_agree(c, rc);
    if (rc) return rc;
    mark("passes");
    if (NR > 1) {
        // partial dictionaries -> the whole dictionary on every rank
        ENSURE(c, c->comm_meta, 8 * 4 * ((size_t)NR + 1));
        uint64_t mine[4] = {nk_total, ndistinct, 0, 0};
        uint64_t* d_meta = ptr<uint64_t>(c->comm_meta);
        HIPCHK(c, hipMemcpyAsync(d_meta + 4 * (size_t)NR, mine, sizeof(mine), hipMemcpyHostToDevice, c->stream));
        NCCLCHK(c, c->api->AllGather(d_meta + 4 * (size_t)NR, d_meta, 4, ncclUint64, c->comm, c->stream));
        std::vector<uint64_t> meta(4 * (size_t)NR);
        HIPCHK(c, hipMemcpyAsync(meta.data(), d_meta, 8 * 4 * (size_t)NR, hipMemcpyDeviceToHost, c->stream));
        { const int sr = comm_sync(c, "dictionary sizes"); if (sr) return sr; }
        std::vector<uint64_t> off((size_t)NR + 1, 0);
        uint64_t nd_all = 0;
        for (int r = 0; r < NR; ++r) { off[r + 1] = off[r] + meta[4 * r]; nd_all += meta[4 * r + 1]; }
        Buf ncode, ncount;
        int r2 = off[NR] >= 0xFFFFFFF0ull ? fail(c, BELLA_ERR_BAD_ARG, "more than 2^32 reliable k-mers") : 0;
        if (!r2) r2 = ensure_bytes(c, ncode, 8 * off[NR]);
        if (!r2) r2 = ensure_bytes(c, ncount, 2 * off[NR] + 16);
        r2 = comm_agree(c, r2);                                    // all ranks hold their receive buffers, or all leave
        if (r2) { release(ncode); release(ncount); return r2; }
        hipError_t he = hipSuccess;
        if (nk_total) {
            he = hipMemcpyAsync(ptr<uint64_t>(ncode) + off[me], c->kc_dcode.p, 8 * nk_total, hipMemcpyDeviceToDevice, c->stream);
            if (he == hipSuccess) he = hipMemcpyAsync(ptr<uint8_t>(ncount) + 2 * off[me], c->kc_dcount.p, 2 * nk_total, hipMemcpyDeviceToDevice, c->stream);
        }
        ncclResult_t nr2 = c->api->GroupStart();
        for (int p2 = 0; p2 < NR && nr2 == ncclSuccess; ++p2) {
            if (p2 == me) continue;
            const uint64_t pn = meta[4 * p2];
            if (nk_total) {
                nr2 = c->api->Send(c->kc_dcode.p, nk_total, ncclUint64, p2, c->comm, c->stream);
                if (nr2 == ncclSuccess) nr2 = c->api->Send(c->kc_dcount.p, 2 * nk_total, ncclUint8, p2, c->comm, c->stream);
            }
            if (pn && nr2 == ncclSuccess) {
                nr2 = c->api->Recv(ptr<uint64_t>(ncode) + off[p2], pn, ncclUint64, p2, c->comm, c->stream);
                if (nr2 == ncclSuccess) nr2 = c->api->Recv(ptr<uint8_t>(ncount) + 2 * off[p2], 2 * pn, ncclUint8, p2, c->comm, c->stream);
            }
        }
        {   // the group is closed on every path
            const ncclResult_t ge = c->api->GroupEnd();
            if (nr2 == ncclSuccess) nr2 = ge;
        }
        if (he == hipSuccess && nr2 == ncclSuccess) { const int sr = comm_sync(c, "dictionary exchange"); if (sr) { release(ncode); release(ncount); return sr; } }
        if (nr2 != ncclSuccess || he != hipSuccess) {
            release(ncode); release(ncount);
            return fail(c, BELLA_ERR_HIP, "dictionary exchange failed: %s", nr2 != ncclSuccess ? (c->api->GetErrorString ? c->api->GetErrorString(nr2) : "RCCL error") : hipGetErrorString(he));
        }
        release(c->kc_dcode); release(c->kc_dcount);
        c->kc_dcode = ncode; c->kc_dcount = ncount;
        nk_total = off[NR];
        ndistinct = nd_all;
    }
    uint32_t* ids_rel = nullptr;
    const unsigned bgrid = brows < 16384u ? (brows ? brows : 1u) : 16384u;
    uint64_t nt = 0;
    if (fast) {
        // the list of occurrences, sorted by position = the tuples in the reference's order (main.cpp:393-416); where each read's begin
        const uint32_t* spos = occ_pos;
        const uint32_t* sid = occ_id;
        if (nocc) {
            if (!occ_pos2) {
                ENSURE(c, c->kc_opos2, 4 * nocc + 16);
                ENSURE(c, c->kc_oid2, 4 * nocc + 16);
                occ_pos2 = ptr<uint32_t>(c->kc_opos2); occ_id2 = ptr<uint32_t>(c->kc_oid2);
            }
            hipcub::DoubleBuffer<uint32_t> dk(occ_pos, occ_pos2);
            hipcub::DoubleBuffer<uint32_t> dv(occ_id, occ_id2);
            size_t tb = 0;
            HIPCHK(c, hipcub::DeviceRadixSort::SortPairs(nullptr, tb, dk, dv, (uint64_t)nocc, 0, (int)pb, c->stream));
            ENSURE(c, c->cubtmp, tb);
            HIPCHK(c, hipcub::DeviceRadixSort::SortPairs(c->cubtmp.p, tb, dk, dv, (uint64_t)nocc, 0, (int)pb, c->stream));
            spos = dk.Current();
            sid = dv.Current();
            mark("sort occurrences");
        }
        uint32_t* first = ptr<uint32_t>(c->kc_found);              // (nr + 2 words)
        k_occ_bounds<<<nblk((uint64_t)brows + 1), 256, 0, c->stream>>>(spos, nocc, ptr<uint64_t>(c->kc_koff) + bfirst, brows, first, ptr<uint64_t>(c->kc_tstart));
        KCHK(c);
        HIPCHK(c, hipMemcpyAsync(&nt, ptr<uint64_t>(c->kc_tstart) + brows, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        ENSURE(c, c->t_kmer, 4 * nt);
        ENSURE(c, c->t_read, 4 * nt);
        ENSURE(c, c->t_pos, 2 * nt);
        if (brows && nt) {
            k_write_tuples_sorted<<<bgrid, kBlock, 0, c->stream>>>(spos, sid, ptr<uint64_t>(c->kc_koff) + bfirst, first, brows, bfirst, ptr<uint32_t>(c->t_kmer),
                                                                   ptr<uint32_t>(c->t_read), ptr<uint16_t>(c->t_pos));
            KCHK(c);
        }
    } else {
    // countsreliable: open addressing at load factor <= 1/2
    uint64_t slots = 1024;
    while (slots < 2 * nk_total) slots <<= 1;
    ENSURE(c, c->kc_hkey, 8 * slots);
    ENSURE(c, c->kc_hval, 4 * slots);
    k_hash_fill<<<nblk(slots), 256, 0, c->stream>>>(ptr<uint64_t>(c->kc_hkey), slots);
    KCHK(c);
    if (nk_total) {
        k_hash_build<<<nblk(nk_total), 256, 0, c->stream>>>(ptr<uint64_t>(c->kc_dcode), (uint32_t)nk_total, ptr<uint64_t>(c->kc_hkey),
                                                            ptr<uint32_t>(c->kc_hval), slots - 1);
        KCHK(c);
    }
    // ids of every position and tuples (main.cpp:393-416) of the reads bfirst .. bfirst + brows - 1 (all reads unless distributed)
    uint64_t kb[2] = {0, 0};                                          // koff[bfirst], koff[bfirst + brows]
    HIPCHK(c, hipMemcpyAsync(&kb[0], ptr<uint64_t>(c->kc_koff) + bfirst, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&kb[1], ptr<uint64_t>(c->kc_koff) + bfirst + brows, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const uint64_t nblockpos = kb[1] - kb[0];
    ENSURE(c, c->kc_keys, 4 * nblockpos);
    ids_rel = ptr<uint32_t>(c->kc_keys) - kb[0];                      // the kernels index ids with the absolute position koff[r] + j
    const uint8_t* sel_rel = d_sel;
    k_lookup_ids<<<bgrid, kBlock, 0, c->stream>>>(ptr<uint32_t>(c->packed), ptr<uint64_t>(c->roff) + bfirst, ptr<uint32_t>(c->kc_nk) + bfirst,
                                                  ptr<uint64_t>(c->kc_koff) + bfirst, brows, k, ptr<uint64_t>(c->kc_hkey), ptr<uint32_t>(c->kc_hval),
                                                  slots - 1, sel_rel, ids_rel, ptr<uint32_t>(c->kc_found));
    KCHK(c);
    HIPCHK(c, hipMemsetAsync(ptr<uint32_t>(c->kc_found) + brows, 0, 4, c->stream));
    rc = scan_u32_to_u64(c, ptr<uint32_t>(c->kc_found), ptr<uint64_t>(c->kc_tstart), (uint64_t)brows + 1);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(&nt, ptr<uint64_t>(c->kc_tstart) + brows, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (nt >= 0xFFFFFFF0ull) return fail(c, BELLA_ERR_BAD_ARG, "tuple count must be < 2^32");
    ENSURE(c, c->t_kmer, 4 * nt);
    ENSURE(c, c->t_read, 4 * nt);
    ENSURE(c, c->t_pos, 2 * nt);
    k_write_tuples<<<bgrid, kBlock, 0, c->stream>>>(ids_rel, ptr<uint32_t>(c->kc_nk) + bfirst, ptr<uint64_t>(c->kc_koff) + bfirst,
                                                    ptr<uint64_t>(c->kc_tstart), brows, bfirst, ptr<uint32_t>(c->t_kmer), ptr<uint32_t>(c->t_read),
                                                    ptr<uint16_t>(c->t_pos));
    KCHK(c);
    }
    mark("tuples");
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev[1]));
    c->tm.kcount_ms = ev_ms(c->ev[0], c->ev[1]);
    release(c->kc_hkey); release(c->kc_hval); release(c->kc_sel);
    release(c->kc_opos); release(c->kc_oid); release(c->kc_opos2); release(c->kc_oid2);
    c->kc_ntuples = nt;
    c->kc_nkmers = (uint32_t)nk_total;
    c->kc_k = k;
    c->kc_bfirst = bfirst;
    c->kc_brows = brows;
    c->have_tuples = true;
    if (nkmers_out) *nkmers_out = (uint32_t)nk_total;
    if (ntuples_out) *ntuples_out = nt;
    if (ndistinct_out) *ndistinct_out = ndistinct;
    return 0;
}

int bella_hip_count_kmers(bella_ctx* c, uint16_t kmer_size, uint32_t lower, uint32_t upper, uint32_t* nkmers, uint64_t* ntuples,
                          uint64_t* ndistinct) {
    return count_kmers_impl(c, kmer_size, lower, upper, 0, 0, nkmers, ntuples, ndistinct);
}

int bella_hip_count_kmers_dist(bella_ctx* c, uint16_t kmer_size, uint32_t lower, uint32_t upper, uint32_t selector, uint32_t window,
                               uint32_t first_read, uint32_t nreads_block, uint32_t* nkmers, uint64_t* ntuples, uint64_t* ndistinct) {
    if (!c) return BELLA_ERR_BAD_ARG;
    if (!c->comm) return fail(c, BELLA_ERR_STATE, "bella_hip_comm_init first");
    // argument and state errors are local: the rank still takes part in a status exchange, so that its peers leave the call with it
    // instead of waiting for it in the dictionary exchange
    int rc = 0;
    if (selector > 2) rc = fail(c, BELLA_ERR_BAD_ARG, "selector: 0 = all k-mers, 1 = syncmers, 2 = minimizers");
    else if (!c->have_reads) rc = fail(c, BELLA_ERR_STATE, "set_reads first");
    else if (kmer_size < 1 || kmer_size > 32) rc = fail(c, BELLA_ERR_BAD_ARG, "k must be in [1,32]");
    else if (lower < 2 || upper < lower || upper > 65535) rc = fail(c, BELLA_ERR_BAD_ARG, "need 2 <= lower <= upper <= 65535");
    else if (selector == 2 && (window < 1 || window > 65535)) rc = fail(c, BELLA_ERR_BAD_ARG, "minimizer window must be in [1,65535]");
    else if (selector == 1 && kmer_size <= kSmerLen) rc = fail(c, BELLA_ERR_BAD_ARG, "syncmer selection needs k > 5 (smerlen, syncmer.hpp:45)");
    else if ((uint64_t)first_read + nreads_block > c->nreads) rc = fail(c, BELLA_ERR_BAD_ARG, "read block exceeds the read set");
    if (hipSetDevice(c->device) != hipSuccess && !rc) rc = fail(c, BELLA_ERR_HIP, "hipSetDevice failed");
    const std::string local_err = c->err;
    rc = comm_agree(c, rc);
    if (rc) { if (!local_err.empty()) c->err = local_err; return rc; }
    return count_kmers_impl(c, kmer_size, lower, upper, selector, window, nkmers, ntuples, ndistinct, true, first_read, nreads_block);
}

int bella_hip_count_syncmers(bella_ctx* c, uint16_t kmer_size, uint32_t lower, uint32_t upper, uint32_t* nkmers, uint64_t* ntuples,
                             uint64_t* ndistinct) {
    return count_kmers_impl(c, kmer_size, lower, upper, 1, 0, nkmers, ntuples, ndistinct);
}

int bella_hip_count_minimizers(bella_ctx* c, uint16_t kmer_size, uint32_t window, uint32_t lower, uint32_t upper, uint32_t* nkmers,
                               uint64_t* ntuples, uint64_t* ndistinct) {
    return count_kmers_impl(c, kmer_size, lower, upper, 2, window, nkmers, ntuples, ndistinct);
}

int bella_hip_get_dictionary(bella_ctx* c, uint64_t* codes, uint16_t* counts) {
    if (!c) return BELLA_ERR_BAD_ARG;
    if (!c->have_tuples) return fail(c, BELLA_ERR_STATE, "count_kmers first");
    HIPCHK(c, hipSetDevice(c->device));
    if (codes && c->kc_nkmers) HIPCHK(c, hipMemcpyAsync(codes, c->kc_dcode.p, 8 * (size_t)c->kc_nkmers, hipMemcpyDeviceToHost, c->stream));
    if (counts && c->kc_nkmers) HIPCHK(c, hipMemcpyAsync(counts, c->kc_dcount.p, 2 * (size_t)c->kc_nkmers, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int bella_hip_get_tuples(bella_ctx* c, uint32_t* t_kmer, uint32_t* t_read, uint16_t* t_pos) {
    if (!c) return BELLA_ERR_BAD_ARG;
    if (!c->have_tuples) return fail(c, BELLA_ERR_STATE, "count_kmers first");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = c->kc_ntuples;
    if (t_kmer && n) HIPCHK(c, hipMemcpyAsync(t_kmer, c->t_kmer.p, 4 * n, hipMemcpyDeviceToHost, c->stream));
    if (t_read && n) HIPCHK(c, hipMemcpyAsync(t_read, c->t_read.p, 4 * n, hipMemcpyDeviceToHost, c->stream));
    if (t_pos && n) HIPCHK(c, hipMemcpyAsync(t_pos, c->t_pos.p, 2 * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int bella_hip_assemble_counted(bella_ctx* c) {
    if (!c) return BELLA_ERR_BAD_ARG;
    if (!c->have_tuples) return fail(c, BELLA_ERR_STATE, "count_kmers first");
    if (c->kc_bfirst != 0 || c->kc_brows != c->nreads) return fail(c, BELLA_ERR_STATE, "the tuples cover a read block only: assemble_counted_panel + allgather_panels");
    HIPCHK(c, hipSetDevice(c->device));
    c->have_matrix = c->have_pairs = c->have_alns = false;
    c->have_panel = false;
    uint64_t nnz = 0;
    int rc = assemble_rows_device(c, 0, c->nreads, c->kc_ntuples, nullptr, nullptr, nullptr, &nnz);
    if (rc) return rc;
    c->nkmers = c->kc_nkmers;
    c->nnz = nnz;
    c->kmer_size = (uint16_t)c->kc_k;
    rc = build_layout(c);
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev[1]));
    c->tm.assemble_ms = ev_ms(c->ev[0], c->ev[1]);
    c->tm.rows_ms = ev_ms(c->ev[0], c->ev[6]);
    release(c->Bk_tmp); release(c->Bpos_tmp); release(c->asm_ws);
    return 0;
}

int bella_hip_assemble_panel(bella_ctx* c, uint16_t kmer_size, uint32_t nkmers, uint32_t first_read, uint32_t nreads_panel,
                             uint64_t ntuples, const uint32_t* t_kmer, const uint32_t* t_read, const uint16_t* t_pos) {
    if (!c || (ntuples && (!t_kmer || !t_read || !t_pos))) return fail(c, BELLA_ERR_BAD_ARG, "null argument");
    if (!c->have_reads) return fail(c, BELLA_ERR_STATE, "set_reads first");
    if (kmer_size < 1 || kmer_size > 32) return fail(c, BELLA_ERR_BAD_ARG, "k must be in [1,32]");
    if ((uint64_t)first_read + nreads_panel > c->nreads) return fail(c, BELLA_ERR_BAD_ARG, "panel exceeds the read set");
    if (ntuples >= 0xFFFFFFF0ull) return fail(c, BELLA_ERR_BAD_ARG, "tuple count must be < 2^32");
    HIPCHK(c, hipSetDevice(c->device));
    c->have_matrix = c->have_pairs = c->have_alns = false;
    c->have_panel = false;
    c->have_tuples = false;
    uint64_t nnz = 0;
    int rc = assemble_rows_device(c, first_read, nreads_panel, ntuples, t_kmer, t_read, t_pos, &nnz);
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev[1]));
    c->tm.assemble_ms = ev_ms(c->ev[0], c->ev[1]);
    c->tm.rows_ms = ev_ms(c->ev[0], c->ev[6]);
    release(c->t_kmer); release(c->t_read); release(c->t_pos); release(c->Bk_tmp); release(c->Bpos_tmp); release(c->asm_ws);
    c->nkmers = nkmers;
    c->kmer_size = kmer_size;
    c->panel_first = first_read;
    c->panel_rows = nreads_panel;
    c->panel_nnz = nnz;
    c->have_panel = true;
    return 0;
}

// a row-block panel straight from the reference's CSC arrays of B (the caller holds the whole matrix on the host and gives every
// context its block: the blocks then travel device to device, bella_hip_allgather_panels)
int bella_hip_set_B_panel(bella_ctx* c, uint16_t kmer_size, uint32_t nkmers, uint32_t first_read, uint32_t nreads_panel, const uint32_t* colptr,
                          const uint32_t* rowids, const uint16_t* values) {
    if (!c || !colptr) return fail(c, BELLA_ERR_BAD_ARG, "null argument");
    if (!c->have_reads) return fail(c, BELLA_ERR_STATE, "set_reads first");
    if (kmer_size < 1 || kmer_size > 32) return fail(c, BELLA_ERR_BAD_ARG, "k must be in [1,32]");
    if ((uint64_t)first_read + nreads_panel > c->nreads) return fail(c, BELLA_ERR_BAD_ARG, "panel exceeds the read set");
    HIPCHK(c, hipSetDevice(c->device));
    c->have_matrix = c->have_pairs = c->have_alns = false;
    c->have_panel = false;
    c->have_tuples = false;
    const uint32_t* cp = colptr + first_read;                      // colptr: the whole matrix's [nreads + 1]
    std::vector<uint32_t> cnt((size_t)nreads_panel + 2, 0);
    for (uint32_t r = 0; r < nreads_panel; ++r) {
        if (cp[r + 1] < cp[r]) return fail(c, BELLA_ERR_BAD_ARG, "colptr must be non-decreasing (column %u)", first_read + r);
        if (cp[r + 1] - cp[r] >= 65536u) return fail(c, BELLA_ERR_READ_TOO_LONG, "column %u has >= 65536 entries", first_read + r);
        cnt[r] = cp[r + 1] - cp[r];
    }
    const uint64_t nnz = (uint64_t)cp[nreads_panel] - cp[0];
    if (nnz && (!rowids || !values)) return fail(c, BELLA_ERR_BAD_ARG, "null argument");
    ENSURE(c, c->rowcnt, 4 * ((size_t)nreads_panel + 2));
    ENSURE(c, c->Bk, 4 * nnz);
    ENSURE(c, c->Bpos, 2 * nnz);
    HIPCHK(c, hipMemcpyAsync(c->rowcnt.p, cnt.data(), 4 * ((size_t)nreads_panel + 2), hipMemcpyHostToDevice, c->stream));
    if (nnz) {
        HIPCHK(c, c->stager.h2d(c->Bk.p, rowids + cp[0], 4 * nnz, c->stream));
        HIPCHK(c, c->stager.h2d(c->Bpos.p, values + cp[0], 2 * nnz, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));                    // cnt is a host vector
    c->nkmers = nkmers;
    c->kmer_size = kmer_size;
    c->panel_first = first_read;
    c->panel_rows = nreads_panel;
    c->panel_nnz = nnz;
    c->have_panel = true;
    return 0;
}

int bella_hip_assemble_counted_panel(bella_ctx* c, uint32_t first_read, uint32_t nreads_panel) {
    if (!c) return BELLA_ERR_BAD_ARG;
    if (!c->have_tuples) return fail(c, BELLA_ERR_STATE, "count_kmers first");
    if ((uint64_t)first_read + nreads_panel > c->nreads) return fail(c, BELLA_ERR_BAD_ARG, "panel exceeds the read set");
    if (first_read < c->kc_bfirst || (uint64_t)first_read + nreads_panel > (uint64_t)c->kc_bfirst + c->kc_brows)
        return fail(c, BELLA_ERR_BAD_ARG, "the device-resident tuples cover the reads %u .. %u only", c->kc_bfirst, c->kc_bfirst + c->kc_brows);
    HIPCHK(c, hipSetDevice(c->device));
    c->have_matrix = c->have_pairs = c->have_alns = false;
    c->have_panel = false;
    uint64_t t0 = 0, t1 = 0;                                       // the panel's tuples are contiguous: tstart of its first / last read
    const uint32_t rel = first_read - c->kc_bfirst;
    HIPCHK(c, hipMemcpyAsync(&t0, ptr<uint64_t>(c->kc_tstart) + rel, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&t1, ptr<uint64_t>(c->kc_tstart) + rel + nreads_panel, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    uint64_t nnz = 0;
    int rc = assemble_rows_device(c, first_read, nreads_panel, t1 - t0, nullptr, nullptr, nullptr, &nnz, t0);
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev[1]));
    c->tm.assemble_ms = ev_ms(c->ev[0], c->ev[1]);
    c->tm.rows_ms = ev_ms(c->ev[0], c->ev[6]);
    release(c->Bk_tmp); release(c->Bpos_tmp); release(c->asm_ws);
    c->nkmers = c->kc_nkmers;
    c->kmer_size = (uint16_t)c->kc_k;
    c->panel_first = first_read;
    c->panel_rows = nreads_panel;
    c->panel_nnz = nnz;
    c->have_panel = true;
    return 0;
}

int bella_hip_panel_device_ptrs(bella_ctx* c, uint32_t* first_read, uint32_t* nreads_panel, uint64_t* nnz, const void** d_rowcnt,
                                const void** d_rowids, const void** d_values) {
    if (!c) return BELLA_ERR_BAD_ARG;
    if (!c->have_panel) return fail(c, BELLA_ERR_STATE, "assemble_panel first");
    if (first_read) *first_read = c->panel_first;
    if (nreads_panel) *nreads_panel = c->panel_rows;
    if (nnz) *nnz = c->panel_nnz;
    if (d_rowcnt) *d_rowcnt = c->rowcnt.p;
    if (d_rowids) *d_rowids = c->Bk.p;
    if (d_values) *d_values = c->Bpos.p;
    return 0;
}

int bella_hip_set_B_device(bella_ctx* c, uint16_t kmer_size, uint32_t nkmers, const uint32_t* d_colptr, const uint32_t* d_rowids,
                           const uint16_t* d_values, uint64_t nnz) {
    if (!c || !d_colptr || (nnz && (!d_rowids || !d_values))) return fail(c, BELLA_ERR_BAD_ARG, "null argument");
    if (!c->have_reads) return fail(c, BELLA_ERR_STATE, "set_reads first");
    if (kmer_size < 1 || kmer_size > 32) return fail(c, BELLA_ERR_BAD_ARG, "k must be in [1,32]");
    HIPCHK(c, hipSetDevice(c->device));
    c->have_matrix = c->have_pairs = c->have_alns = false;
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    // the sources may alias our own panel buffers: stage through fresh allocations
    Buf nBptr, nBk, nBpos;
    int rc = ensure_bytes(c, nBptr, 4 * ((size_t)c->nreads + 2));
    if (!rc) rc = ensure_bytes(c, nBk, 4 * nnz);
    if (!rc) rc = ensure_bytes(c, nBpos, 2 * nnz);
    if (rc) { release(nBptr); release(nBk); release(nBpos); return rc; }
    hipError_t e = hipMemcpyAsync(nBptr.p, d_colptr, 4 * ((size_t)c->nreads + 1), hipMemcpyDeviceToDevice, c->stream);
    if (e == hipSuccess && nnz) e = hipMemcpyAsync(nBk.p, d_rowids, 4 * nnz, hipMemcpyDeviceToDevice, c->stream);
    if (e == hipSuccess && nnz) e = hipMemcpyAsync(nBpos.p, d_values, 2 * nnz, hipMemcpyDeviceToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { release(nBptr); release(nBk); release(nBpos); return fail(c, BELLA_ERR_HIP, "set_B_device: %s", hipGetErrorString(e)); }
    release(c->Bptr); release(c->Bk); release(c->Bpos);
    c->Bptr = nBptr; c->Bk = nBk; c->Bpos = nBpos;
    c->have_panel = false;
    c->nkmers = nkmers;
    c->nnz = nnz;
    c->kmer_size = kmer_size;
    rc = build_layout(c);
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev[1]));
    c->tm.assemble_ms = ev_ms(c->ev[0], c->ev[1]);
    return 0;
}

// ---- multi-GPU: RCCL communicator + panel all-gather -------------------------------------------------------------------------
int bella_hip_comm_available(void) { return rccl().ok() ? 1 : 0; }

static int comm_id_impl(const Rccl& api, uint8_t id[BELLA_HIP_COMM_ID_BYTES]) {
    if (!id) return BELLA_ERR_BAD_ARG;
    if (!api.ok()) return BELLA_ERR_STATE;
    ncclUniqueId u;
    static_assert(sizeof(u) == BELLA_HIP_COMM_ID_BYTES, "ncclUniqueId size");
    if (api.GetUniqueId(&u) != ncclSuccess) return BELLA_ERR_HIP;
    std::memcpy(id, &u, sizeof(u));
    return 0;
}
int bella_hip_comm_id(uint8_t id[BELLA_HIP_COMM_ID_BYTES]) { return comm_id_impl(rccl(), id); }
int bella_hip_comm_id_local(uint8_t id[BELLA_HIP_COMM_ID_BYTES]) { return comm_id_impl(loopback(), id); }

// Wait for what the collective calls put on the stream -- with a deadline (BELLA_HIP_COMM_TIMEOUT_S, default 120 s): a peer that never
// arrives must not hang the process.  On expiry the communicator is aborted (ncclCommAbort, where the transport has it) and the call
// fails like any other error: the caller falls back (bench.py, bella_amd/dist.py: the torch.distributed exchange).
static int comm_sync(bella_ctx* c, const char* what) {
    const double limit = loop_timeout_s();                      // BELLA_HIP_COMM_TIMEOUT_S, default 120 (comm.hpp)
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    for (;;) {
        const hipError_t q = hipStreamQuery(c->stream);
        if (q == hipSuccess) return 0;
        if (q != hipErrorNotReady) { (void)hipGetLastError(); return fail(c, BELLA_ERR_HIP, "%s: %s", what, hipGetErrorString(q)); }
        (void)hipGetLastError();
        if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(200));
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
            if (c->comm && c->api && c->api->CommAbort) { (void)c->api->CommAbort(c->comm); c->comm = nullptr; c->comm_ranks = 0; }
            return fail(c, BELLA_ERR_STATE, "%s: no completion within %.0f s (a peer did not arrive); the communicator was aborted", what, limit);
        }
    }
}

// Every entry point of the transport once, right after the communicator exists (all ranks are inside bella_hip_comm_init together): an
// all-gather of the ranks' numbers and a grouped send/recv to the rank ITSELF -- a missing symbol, an ABI mismatch or a communicator
// that does not move data fails HERE, at init, not inside the first exchange of a run.
static int comm_selftest(bella_ctx* c) {
    const int N = c->comm_ranks, me = c->comm_rank;
    // (sized here, once, for every later use -- the status exchanges and the metadata lines of the shared formation of A': no collective
    // entry point can find itself without the buffer its peers are waiting to hear from it through)
    ENSURE(c, c->comm_meta, 8 * (((size_t)N + 3) * ((size_t)N + 1) + 4 * ((size_t)N + 1)) + 8 * 4 * ((size_t)N + 1) + 64);
    uint64_t* d = ptr<uint64_t>(c->comm_meta);
    const uint64_t mine[4] = {0xBE11A000ull + (uint64_t)me, ~(uint64_t)me, 0, 0};
    HIPCHK(c, hipMemsetAsync(d, 0, 8 * 4 * ((size_t)N + 1) + 64, c->stream));
    HIPCHK(c, hipMemcpyAsync(d + 4 * (size_t)N, mine, 16, hipMemcpyHostToDevice, c->stream));
    NCCLCHK(c, c->api->AllGather(d + 4 * (size_t)N, d, 1, ncclUint64, c->comm, c->stream));
    ncclResult_t r = c->api->GroupStart();
    if (r == ncclSuccess) r = c->api->Send(d + 4 * (size_t)N + 1, 1, ncclUint64, me, c->comm, c->stream);
    if (r == ncclSuccess) r = c->api->Recv(d + 4 * (size_t)N + 2, 1, ncclUint64, me, c->comm, c->stream);
    const ncclResult_t ge = c->api->GroupEnd();
    if (r == ncclSuccess) r = ge;
    if (r != ncclSuccess) return fail(c, BELLA_ERR_HIP, "communicator self-test: %s", c->api->GetErrorString ? c->api->GetErrorString(r) : "RCCL error");
    std::vector<uint64_t> got((size_t)N + 4, 0);
    HIPCHK(c, hipMemcpyAsync(got.data(), d, 8 * (size_t)N, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(got.data() + N, d + 4 * (size_t)N, 24, hipMemcpyDeviceToHost, c->stream));
    int rc = comm_sync(c, "communicator self-test");
    if (rc) return rc;
    for (int q = 0; q < N; ++q)
        if (got[(size_t)q] != 0xBE11A000ull + (uint64_t)q) return fail(c, BELLA_ERR_STATE, "communicator self-test: the all-gather delivered %llx for rank %d", (unsigned long long)got[(size_t)q], q);
    if (got[(size_t)N + 2] != ~(uint64_t)me) return fail(c, BELLA_ERR_STATE, "communicator self-test: send/recv to self delivered %llx", (unsigned long long)got[(size_t)N + 2]);
    return 0;
}

static int comm_init_impl(bella_ctx* c, const Rccl& api, int nranks, int rank, const uint8_t id[BELLA_HIP_COMM_ID_BYTES]) {
    if (!c || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(c, BELLA_ERR_BAD_ARG, "bad communicator arguments");
    if (!api.ok()) return fail(c, BELLA_ERR_STATE, "librccl could not be loaded");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->comm && c->api) { (void)c->api->CommDestroy(c->comm); c->comm = nullptr; }
    c->api = &api;
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    NCCLCHK(c, api.CommInitRank(&c->comm, nranks, u, rank));
    c->comm_ranks = nranks;
    c->comm_rank = rank;
    const int rc = comm_selftest(c);
    if (rc) {                                                    // a communicator that fails its self-test is not kept
        if (c->comm) { (void)(api.CommAbort ? api.CommAbort(c->comm) : api.CommDestroy(c->comm)); c->comm = nullptr; }
        c->comm_ranks = 0;
    }
    return rc;
}
int bella_hip_comm_init(bella_ctx* c, int nranks, int rank, const uint8_t id[BELLA_HIP_COMM_ID_BYTES]) {
    return comm_init_impl(c, rccl(), nranks, rank, id);
}
int bella_hip_comm_init_local(bella_ctx* c, int nranks, int rank, const uint8_t id[BELLA_HIP_COMM_ID_BYTES]) {
    return comm_init_impl(c, loopback(), nranks, rank, id);
}

int bella_hip_comm_destroy(bella_ctx* c) {
    if (!c) return BELLA_ERR_BAD_ARG;
    if (c->comm && c->api) { (void)c->api->CommDestroy(c->comm); c->comm = nullptr; }
    c->comm_ranks = 0;
    return 0;
}

// Collective agreement: every rank contributes its local status; if any rank failed, ALL return an error here, together (a rank
// that left a collective entry point on a local error would leave its peers waiting in the next exchange).
static int comm_agree(bella_ctx* c, int local_rc) {
    const int N = c->comm_ranks;
    if (ensure_bytes(c, c->comm_meta, 8 * 4 * ((size_t)N + 1))) local_rc = local_rc ? local_rc : BELLA_ERR_NOMEM;
    if (!c->comm_meta.p) return local_rc ? local_rc : BELLA_ERR_NOMEM;      // (nothing to exchange through: cannot happen after the first call)
    uint64_t* d_meta = ptr<uint64_t>(c->comm_meta);
    const uint64_t mine = local_rc ? 1 : 0;
    std::vector<uint64_t> all((size_t)N, 0);
    hipError_t e = hipMemcpyAsync(d_meta + 4 * (size_t)N, &mine, 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    ncclResult_t r = c->api->AllGather(d_meta + 4 * (size_t)N, d_meta, 1, ncclUint64, c->comm, c->stream);
    if (r != ncclSuccess) return local_rc ? local_rc : fail(c, BELLA_ERR_HIP, "status exchange failed");
    if (e == hipSuccess) e = hipMemcpyAsync(all.data(), d_meta, 8 * (size_t)N, hipMemcpyDeviceToHost, c->stream);
    if (e != hipSuccess) return local_rc ? local_rc : fail(c, BELLA_ERR_HIP, "status exchange: %s", hipGetErrorString(e));
    { const int sr = comm_sync(c, "status exchange"); if (sr) return local_rc ? local_rc : sr; }
    if (local_rc) return local_rc;
    for (int q = 0; q < N; ++q)
        if (all[(size_t)q]) return fail(c, BELLA_ERR_STATE, "rank %d failed inside the collective call; all ranks leave it", q);
    return 0;
}

// ---- the formation of A' shared over the ranks (BELLA_TUNE_DIST_LAYOUT; kernels: assemble.hpp) ---------------------------------------
// Every rank holds the whole exchanged matrix B (6 B per nonzero) and computes the columns i % N == rank.  Replicated, the device layout
// sorts ALL entries by k-mer on every rank to form A' (prep + sort: 7.7 of a rank's 11.4 ms at N = 8, 100k reads).  Here rank g keeps the
// entries whose k-mer id lies in the g-th N-th of the id space, sorts those, emits ITS slice of A' (in place in the whole array) and
// the B' entries of ALL rows for those k-mers; then two grouped exchanges: the slices of A' (all-gather: 8 B per nonzero) and the B'
// entries to the owners of their rows (all-to-all: 12 B per nonzero).  Same layout as the replicated one, entry for entry.
extern "C++" {
namespace {
// what a rank needs for the shared formation (asked of every rank in bella_hip_allgather_panels; taken only if all can)
bool layout_dist_eligible(const bella_ctx* c) {
    const bool first_app = c->tune_layout_order == 1 || (c->debug & 1024u);
    const bool wanted = c->tune_dist_layout == 1 || (c->tune_dist_layout == 2 && c->api == &loopback());
    return wanted && c->comm && c->api && c->comm_ranks > 1 && (uint32_t)c->comm_ranks <= kOwnerMax &&
           c->part_stride == (uint32_t)c->comm_ranks && c->part_first == (uint32_t)c->comm_rank && !first_app && c->nreads <= (1u << 30);
}
// a rank that failed before its formation began: its line of the metadata exchange says so (the ranks inside layout_dist read it and leave)
void layout_dist_fail(bella_ctx* c) {
    const size_t N = (size_t)c->comm_ranks, W = N + 3;
    if (ensure_bytes(c, c->comm_meta, 8 * (W * (N + 1) + 4 * (N + 1))) || !c->comm_meta.p) return;
    uint64_t* d_meta = ptr<uint64_t>(c->comm_meta);
    std::vector<uint64_t> line(W, 0);
    line[0] = 1;
    if (hipMemcpyAsync(d_meta + W * N, line.data(), 8 * W, hipMemcpyHostToDevice, c->stream) != hipSuccess) return;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return;
    if (c->api->AllGather(d_meta + W * N, d_meta, W, ncclUint64, c->comm, c->stream) != ncclSuccess) return;
    (void)comm_sync(c, "layout metadata");
}
int layout_dist(bella_ctx* c, uint64_t nown_nnz, uint32_t rmask, uint32_t inl, int kbits, uint32_t** ekey_out, uint64_t** eval_out,
                uint32_t** fkey_out, uint64_t** fval_out) {
    const int N = c->comm_ranks, me = c->comm_rank;
    const uint32_t nr = c->nreads, nk = c->nkmers;
    const uint64_t nnz = c->nnz;
    const uint32_t klo = (uint32_t)((uint64_t)nk * (uint64_t)me / (uint64_t)N), khi = (uint32_t)((uint64_t)nk * (uint64_t)(me + 1) / (uint64_t)N);
    const uint32_t M = (nr + (uint32_t)N - 1) / (uint32_t)N;      // rows per owner, at most
    const size_t W = (size_t)N + 3;                               // words of a rank's line of the metadata exchange
    Buf pbuf, sbuf, hbuf;                                         // owner-major row lengths + their scan; per-owner counters and cursors
    auto done = [&](int rc) { release(pbuf); release(sbuf); release(hbuf); return rc; };
    // (1) this rank's entries: per row how many, where they go, how many for the rows of each rank.  Host synchronisations of the
    // whole formation: this read-back, the metadata exchange, the end of the data exchange, the closing agreement.
    int rc = ensure_bytes(c, pbuf, 4 * ((size_t)N * M + 2));
    if (!rc) rc = ensure_bytes(c, sbuf, 4 * ((size_t)N * M + 2));
    if (!rc) rc = ensure_bytes(c, hbuf, 8 * 2 * (size_t)kOwnerMax);
    if (!rc) rc = ensure_bytes(c, c->comm_meta, 8 * (W * ((size_t)N + 1) + 4 * ((size_t)N + 1)));
    uint32_t* cnt = ptr<uint32_t>(c->w);
    uint32_t* rowbase = ptr<uint32_t>(c->wscan);
    unsigned long long* hist = ptr<unsigned long long>(hbuf);
    std::vector<uint64_t> line(W, 0);                             // {failed, entries of my slice, entries of my rows, entries for the rows of rank 0 .. N-1}
    uint32_t nsel32 = 0;
    if (!rc) rc = [&]() -> int {
        HIPCHK(c, hipMemsetAsync(hist, 0, 8 * 2 * (size_t)kOwnerMax, c->stream));
        k_range_rowcount<<<nblk((uint64_t)nr + 1, kWaves), kBlock, 0, c->stream>>>(ptr<uint32_t>(c->Bptr), ptr<uint32_t>(c->Bk), nr, klo, khi, cnt, nk,
                                                                                   ptr<uint32_t>(c->status));
        KCHK(c);
        k_owner_hist_rows<<<nblk(nr ? nr : 1), 256, 0, c->stream>>>(cnt, nr, (uint32_t)N, hist);
        KCHK(c);
        const int sr = scan_u32(c, cnt, rowbase, (uint64_t)nr + 1);
        if (sr) return sr;
        HIPCHK(c, hipMemcpyAsync(&nsel32, rowbase + nr, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(line.data() + 3, hist, 8 * (size_t)N, hipMemcpyDeviceToHost, c->stream));
        // a k-mer id beyond the dictionary belongs to no rank's range: the slices would not add up and every rank would report a
        // STATE error; the rank that sees it reports the bad input instead (status bit 32, as the replicated formation does)
        uint32_t st0 = 0;
        const int r0 = read_status(c, &st0);
        if (r0) return r0;
        return status_to_error(c, st0);
    }();
    const uint64_t nsel = nsel32;
    // (2) buffers for the rank's share: what it sorts (nsel) and what it receives (nown_nnz)
    const uint64_t cap = std::max<uint64_t>(std::max<uint64_t>(nsel, nown_nnz), 1);
    if (!rc) rc = ensure_bytes(c, c->lk_key, 4 * cap);
    if (!rc) rc = ensure_bytes(c, c->lk_key2, 4 * cap);
    if (!rc) rc = ensure_bytes(c, c->lk_val, 8 * cap);
    if (!rc) rc = ensure_bytes(c, c->lk_val2, 8 * cap);
    // (3) every rank's line to every rank: all take the same decision from the same table
    if (!c->comm_meta.p) return done(rc ? rc : BELLA_ERR_NOMEM);  // (nothing to exchange through: cannot happen after the first collective call)
    uint64_t* d_meta = ptr<uint64_t>(c->comm_meta);
    line[0] = rc ? 1 : 0; line[1] = nsel; line[2] = nown_nnz;
    std::vector<uint64_t> T(W * (size_t)N, 0);
    {
        hipError_t e = hipMemcpyAsync(d_meta + W * (size_t)N, line.data(), 8 * W, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        const ncclResult_t r = c->api->AllGather(d_meta + W * (size_t)N, d_meta, W, ncclUint64, c->comm, c->stream);
        if (r != ncclSuccess) return done(rc ? rc : fail(c, BELLA_ERR_HIP, "layout metadata exchange failed"));
        if (e == hipSuccess) e = hipMemcpyAsync(T.data(), d_meta, 8 * W * (size_t)N, hipMemcpyDeviceToHost, c->stream);
        if (e != hipSuccess) return done(rc ? rc : fail(c, BELLA_ERR_HIP, "layout metadata exchange: %s", hipGetErrorString(e)));
        const int sr = comm_sync(c, "layout metadata");
        if (sr) return done(rc ? rc : sr);
    }
    if (rc) return done(rc);
    std::vector<uint64_t> abase((size_t)N + 1, 0), roffs((size_t)N + 1, 0), ooff((size_t)N + 1, 0);
    for (int p = 0; p < N; ++p) {
        if (T[W * (size_t)p]) return done(fail(c, BELLA_ERR_STATE, "rank %d failed inside the collective call; all ranks leave it", p));
        abase[(size_t)p + 1] = abase[(size_t)p] + T[W * (size_t)p + 1];
    }
    if (abase[(size_t)N] != nnz)
        return done(fail(c, BELLA_ERR_STATE, "the ranks' slices hold %llu of %llu entries (different matrices or dictionaries?)",
                         (unsigned long long)abase[(size_t)N], (unsigned long long)nnz));
    for (int q = 0; q < N; ++q) {                                 // (every rank checks every rank's rows: the same verdict everywhere)
        uint64_t got = 0;
        for (int p = 0; p < N; ++p) got += T[W * (size_t)p + 3 + (size_t)q];
        if (got != T[W * (size_t)q + 2])
            return done(fail(c, BELLA_ERR_STATE, "the ranks hold %llu entries for the rows of rank %d, its rows have %llu", (unsigned long long)got, q,
                             (unsigned long long)T[W * (size_t)q + 2]));
    }
    auto S = [&](int p, int q) { return T[W * (size_t)p + 3 + (size_t)q]; };    // entries rank p holds for the rows of rank q
    for (int p = 0; p < N; ++p) { roffs[(size_t)p + 1] = roffs[(size_t)p] + S(p, me); ooff[(size_t)p + 1] = ooff[(size_t)p] + S(me, p); }
    // (4) my entries in k-mer order
    hipcub::DoubleBuffer<uint32_t> dk(ptr<uint32_t>(c->lk_key), ptr<uint32_t>(c->lk_key2));
    hipcub::DoubleBuffer<uint64_t> dv(ptr<uint64_t>(c->lk_val), ptr<uint64_t>(c->lk_val2));
    // (valid buffers of the agreed sizes whatever happens below: a rank whose passes fail still moves its -- then meaningless -- share)
    uint32_t *okey = ptr<uint32_t>(c->lk_key2), *rkey = ptr<uint32_t>(c->lk_key);
    uint64_t *oval = ptr<uint64_t>(c->lk_val2), *rval = ptr<uint64_t>(c->lk_val);
    rc = [&]() -> int {
        if (nsel) {
            k_layout_prep_range<<<nblk(nr, kWaves), kBlock, 0, c->stream>>>(ptr<uint32_t>(c->Bptr), ptr<uint32_t>(c->Bk), ptr<uint16_t>(c->Bpos), nr,
                                                                           ptr<uint32_t>(c->packed), ptr<uint64_t>(c->roff), c->kmer_size, nk, klo, khi,
                                                                           rowbase, dk.Current(), dv.Current(), ptr<uint32_t>(c->status));
            KCHK(c);
            size_t tb = 0;
            HIPCHK(c, hipcub::DeviceRadixSort::SortPairs(nullptr, tb, dk, dv, (uint64_t)nsel, 0, kbits, c->stream));
            ENSURE(c, c->cubtmp, tb);
            HIPCHK(c, hipcub::DeviceRadixSort::SortPairs(c->cubtmp.p, tb, dk, dv, (uint64_t)nsel, 0, kbits, c->stream));
        }
        // (5) where a row starts in ITS owner's B': owner-major lengths, one scan
        uint2* rinfo = ptr<uint2>(c->lk_rinfo);
        k_owner_lengths<<<nblk((uint64_t)N * M + 1), 256, 0, c->stream>>>(ptr<uint32_t>(c->Bptr), nr, (uint32_t)N, M, ptr<uint32_t>(pbuf));
        KCHK(c);
        const int sr = scan_u32(c, ptr<uint32_t>(pbuf), ptr<uint32_t>(sbuf), (uint64_t)N * M + 1);
        if (sr) return sr;
        k_layout_rinfo_owner<<<nblk(nr ? nr : 1), 256, 0, c->stream>>>(ptr<uint32_t>(sbuf), ptr<uint64_t>(c->roff), nr, (uint32_t)N, M, rinfo);
        KCHK(c);
        // (6) my slice of A' (in place in the whole array) and the B' entries of all rows for my k-mers, each into its owner's range
        okey = dk.Alternate(); oval = dv.Alternate();
        rkey = dk.Current(); rval = dv.Current();                 // (the sorted arrays are free after this pass: what the rank receives goes there)
        if (nsel) {
            HIPCHK(c, hipMemcpyAsync(hist + kOwnerMax, ooff.data(), 8 * (size_t)N, hipMemcpyHostToDevice, c->stream));
            k_layout_emit<<<nblk(nsel, kLayoutEmitPartBlock), kLayoutEmitPartBlock, 0, c->stream>>>(
                dk.Current(), dv.Current(), nsel, ptr<uint32_t>(c->Bptr), ptr<uint32_t>(c->Bloc), ptr<uint32_t>(c->wscan), ptr<uint32_t>(c->packed),
                ptr<uint64_t>(c->roff), c->kmer_size, rmask, ptr<uint2>(c->Aent), okey, oval, 1u, (uint32_t)me, (uint32_t)N, nullptr, ptr<uint32_t>(c->status),
                inl, rinfo, (uint32_t)abase[(size_t)me], hist + kOwnerMax);
            KCHK(c);
        }
        return 0;
    }();
    // A rank that could not launch its passes still takes part in the exchange (the sizes are agreed: nobody waits); its error leaves
    // with the closing agreement.
    // (7) the two exchanges, one group: B' entries to their owners, slices of A' into place
    hipError_t he = hipSuccess;
    ncclResult_t nr2 = ncclSuccess;
    {
        if (S(me, me)) {
            he = hipMemcpyAsync(rkey + roffs[(size_t)me], okey + ooff[(size_t)me], 4 * (size_t)S(me, me), hipMemcpyDeviceToDevice, c->stream);
            if (he == hipSuccess) he = hipMemcpyAsync(rval + roffs[(size_t)me], oval + ooff[(size_t)me], 8 * (size_t)S(me, me), hipMemcpyDeviceToDevice, c->stream);
        }
        uint64_t* A64 = ptr<uint64_t>(c->Aent);
        nr2 = c->api->GroupStart();
        for (int p = 0; p < N && nr2 == ncclSuccess; ++p) {
            if (p == me) continue;
            const uint64_t to = S(me, p), from = S(p, me), theirs = T[W * (size_t)p + 1];
            if (to) {
                nr2 = c->api->Send(okey + ooff[(size_t)p], to, ncclUint32, p, c->comm, c->stream);
                if (nr2 == ncclSuccess) nr2 = c->api->Send(oval + ooff[(size_t)p], to, ncclUint64, p, c->comm, c->stream);
            }
            if (from && nr2 == ncclSuccess) {
                nr2 = c->api->Recv(rkey + roffs[(size_t)p], from, ncclUint32, p, c->comm, c->stream);
                if (nr2 == ncclSuccess) nr2 = c->api->Recv(rval + roffs[(size_t)p], from, ncclUint64, p, c->comm, c->stream);
            }
            if (nsel && nr2 == ncclSuccess) nr2 = c->api->Send(A64 + abase[(size_t)me], nsel, ncclUint64, p, c->comm, c->stream);
            if (theirs && nr2 == ncclSuccess) nr2 = c->api->Recv(A64 + abase[(size_t)p], theirs, ncclUint64, p, c->comm, c->stream);
        }
        {
            const ncclResult_t ge = c->api->GroupEnd();           // the group is closed on every path
            if (nr2 == ncclSuccess) nr2 = ge;
        }
    }
    uint32_t st = 0;
    if (he == hipSuccess && nr2 == ncclSuccess) he = hipMemcpyAsync(&st, ptr<uint32_t>(c->status), 4, hipMemcpyDeviceToHost, c->stream);
    if (nr2 != ncclSuccess || he != hipSuccess) {
        (void)hipStreamSynchronize(c->stream);
        if (!rc) rc = fail(c, BELLA_ERR_HIP, "layout exchange failed: %s", nr2 != ncclSuccess ? (c->api->GetErrorString ? c->api->GetErrorString(nr2) : "RCCL error") : hipGetErrorString(he));
    } else {
        const int sr = comm_sync(c, "layout exchange");
        if (!rc) rc = sr;
        if (!rc) rc = status_to_error(c, st);
    }
    rc = comm_agree(c, rc);                                       // all ranks hold the layout's inputs, or all leave
    if (rc) return done(rc);
    *ekey_out = rkey; *eval_out = rval; *fkey_out = okey; *fval_out = oval;
    return done(0);
}
}  // namespace
}  // extern "C++"

int bella_hip_allgather_panels(bella_ctx* c) {
    if (!c) return BELLA_ERR_BAD_ARG;
    if (!c->comm) return fail(c, BELLA_ERR_STATE, "bella_hip_comm_init first");
    HIPCHK(c, hipSetDevice(c->device));
    const int N = c->comm_ranks, me = c->comm_rank;
    c->dist_agreed = false;
    // a rank without a panel still takes part in the status exchange: all ranks leave the call together
    int rc = comm_agree(c, c->have_panel ? 0 : fail(c, BELLA_ERR_STATE, "assemble_panel first"));
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    // who holds what: {first read, rows, nnz} of every rank
    ENSURE(c, c->comm_meta, 8 * 4 * ((size_t)N + 1));
    // (the fourth word also says whether this rank can take the shared formation of A': all must, or none does)
    uint64_t mine[4] = {c->panel_first, c->panel_rows, c->panel_nnz, (uint64_t)c->nkmers | (layout_dist_eligible(c) ? 1ull << 32 : 0ull)};
    uint64_t* d_meta = ptr<uint64_t>(c->comm_meta);
    HIPCHK(c, hipMemcpyAsync(d_meta + 4 * (size_t)N, mine, sizeof(mine), hipMemcpyHostToDevice, c->stream));
    NCCLCHK(c, c->api->AllGather(d_meta + 4 * (size_t)N, d_meta, 4, ncclUint64, c->comm, c->stream));
    std::vector<uint64_t> meta(4 * (size_t)N);
    HIPCHK(c, hipMemcpyAsync(meta.data(), d_meta, 8 * 4 * (size_t)N, hipMemcpyDeviceToHost, c->stream));
    rc = comm_sync(c, "panel sizes");
    if (rc) return rc;
    // (the checks below see the same numbers on every rank: all ranks fail them together)
    std::vector<uint64_t> roff((size_t)N + 1, 0), eoff((size_t)N + 1, 0);
    c->dist_agreed = true;
    for (int r = 0; r < N; ++r) {
        if (!(meta[4 * r + 3] >> 32)) c->dist_agreed = false;
        meta[4 * r + 3] &= 0xFFFFFFFFull;
        if (meta[4 * r] != roff[r]) return fail(c, BELLA_ERR_BAD_ARG, "panels must be consecutive read blocks in rank order (rank %d starts at %llu, expected %llu)",
                                                r, (unsigned long long)meta[4 * r], (unsigned long long)roff[r]);
        if (meta[4 * r + 3] != meta[3]) return fail(c, BELLA_ERR_BAD_ARG, "rank %d counted a different k-mer dictionary", r);
        roff[r + 1] = roff[r] + meta[4 * r + 1];
        eoff[r + 1] = eoff[r] + meta[4 * r + 2];
    }
    if (roff[N] != c->nreads) return fail(c, BELLA_ERR_BAD_ARG, "the panels cover %llu of %u reads", (unsigned long long)roff[N], c->nreads);
    const uint64_t nnz = eoff[N];
    if (nnz >= 0xFFFFFFF0ull) return fail(c, BELLA_ERR_BAD_ARG, "nnz(A) must be < 2^32");
    Buf nCnt, nBk, nBpos, nBptr;
    rc = ensure_bytes(c, nCnt, 4 * ((size_t)c->nreads + 2));
    if (!rc) rc = ensure_bytes(c, nBptr, 4 * ((size_t)c->nreads + 2));
    if (!rc) rc = ensure_bytes(c, nBk, 4 * nnz);
    if (!rc) rc = ensure_bytes(c, nBpos, 2 * nnz + 16);
    rc = comm_agree(c, rc);                                       // all ranks hold the full arrays, or all leave
    if (rc) { release(nCnt); release(nBk); release(nBpos); release(nBptr); return rc; }
    // one grouped exchange: to every peer my block, from every peer its block, straight to its place in the full arrays
    uint32_t* cnt = ptr<uint32_t>(nCnt);
    uint32_t* bk = ptr<uint32_t>(nBk);
    uint8_t* bpos = ptr<uint8_t>(nBpos);
    hipError_t he = hipMemsetAsync(cnt + c->nreads, 0, 8, c->stream);
    if (he == hipSuccess && c->panel_rows) he = hipMemcpyAsync(cnt + roff[me], c->rowcnt.p, 4 * (size_t)c->panel_rows, hipMemcpyDeviceToDevice, c->stream);
    if (he == hipSuccess && c->panel_nnz) he = hipMemcpyAsync(bk + eoff[me], c->Bk.p, 4 * (size_t)c->panel_nnz, hipMemcpyDeviceToDevice, c->stream);
    if (he == hipSuccess && c->panel_nnz) he = hipMemcpyAsync(bpos + 2 * eoff[me], c->Bpos.p, 2 * (size_t)c->panel_nnz, hipMemcpyDeviceToDevice, c->stream);
    ncclResult_t nr2 = ncclSuccess;
    if (N > 1) {
        nr2 = c->api->GroupStart();
        for (int p = 0; p < N && nr2 == ncclSuccess; ++p) {
            if (p == me) continue;
            const uint64_t prow = meta[4 * p + 1], pnnz = meta[4 * p + 2];
            if (c->panel_rows) nr2 = c->api->Send(c->rowcnt.p, c->panel_rows, ncclUint32, p, c->comm, c->stream);
            if (prow && nr2 == ncclSuccess) nr2 = c->api->Recv(cnt + roff[p], prow, ncclUint32, p, c->comm, c->stream);
            if (c->panel_nnz && nr2 == ncclSuccess) {
                nr2 = c->api->Send(c->Bk.p, c->panel_nnz, ncclUint32, p, c->comm, c->stream);
                if (nr2 == ncclSuccess) nr2 = c->api->Send(c->Bpos.p, 2 * c->panel_nnz, ncclUint8, p, c->comm, c->stream);
            }
            if (pnnz && nr2 == ncclSuccess) {
                nr2 = c->api->Recv(bk + eoff[p], pnnz, ncclUint32, p, c->comm, c->stream);
                if (nr2 == ncclSuccess) nr2 = c->api->Recv(bpos + 2 * eoff[p], 2 * pnnz, ncclUint8, p, c->comm, c->stream);
            }
        }
        const ncclResult_t ge = c->api->GroupEnd();               // the group is closed on every path
        if (nr2 == ncclSuccess) nr2 = ge;
    }
    if (nr2 != ncclSuccess || he != hipSuccess) {
        (void)hipStreamSynchronize(c->stream);
        release(nCnt); release(nBk); release(nBpos); release(nBptr);
        return fail(c, BELLA_ERR_HIP, "panel exchange failed: %s", nr2 != ncclSuccess ? (c->api->GetErrorString ? c->api->GetErrorString(nr2) : "RCCL error") : hipGetErrorString(he));
    }
    rc = scan_u32(c, cnt, ptr<uint32_t>(nBptr), (uint64_t)c->nreads + 1);
    if (!rc) rc = comm_sync(c, "panel exchange");
    release(nCnt);
    if (rc) { release(nBk); release(nBpos); release(nBptr); return rc; }
    release(c->Bptr); release(c->Bk); release(c->Bpos);
    c->Bptr = nBptr; c->Bk = nBk; c->Bpos = nBpos;
    c->have_panel = false;
    c->have_matrix = c->have_pairs = c->have_alns = false;
    c->nnz = nnz;
    rc = build_layout(c, true);
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev[1]));
    c->tm.assemble_ms = ev_ms(c->ev[0], c->ev[1]);     // exchange + layout
    return 0;
}

int bella_hip_get_B(bella_ctx* c, uint64_t* nnz, uint32_t* colptr, uint32_t* rowids, uint16_t* values) {
    if (!c) return BELLA_ERR_BAD_ARG;
    if (!c->have_matrix) return fail(c, BELLA_ERR_STATE, "no matrix");
    HIPCHK(c, hipSetDevice(c->device));
    if (nnz) *nnz = c->nnz;
    if (colptr) HIPCHK(c, hipMemcpyAsync(colptr, c->Bptr.p, 4 * ((size_t)c->nreads + 1), hipMemcpyDeviceToHost, c->stream));
    if (rowids && c->nnz) HIPCHK(c, hipMemcpyAsync(rowids, c->Bk.p, 4 * c->nnz, hipMemcpyDeviceToHost, c->stream));
    if (values && c->nnz) HIPCHK(c, hipMemcpyAsync(values, c->Bpos.p, 2 * c->nnz, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

__global__ void k_wide_sizes(const uint32_t* cols, const uint32_t* flops, uint32_t nw, uint32_t* wf) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > nw) return;
    wf[s] = s < nw ? flops[cols[s]] : 0u;
}

// columns with >= 65536 products: expand -> sort by (column, partner) -> pairs -> slot order -> serial fold (wide.hpp)
// T: the batch's products (run_wide has the columns' sizes on the host: no second trip for their sum)
// A batch runs in two parts: wide_batch_begin enqueues everything up to the point where the host must know the batch's pair count (the
// columns' sizes, the row lists if the layout has none, the first grouping pass and the copy of its totals into pinned memory) and
// returns without waiting -- run_spgemm enqueues it BEFORE the LDS classes' row kernels, so that the first grouping pass starts with
// them instead of behind their queue --; wide_batch_finish waits for the totals and enqueues the rest.
struct WideBatch {
    WideArgs a{};
    uint32_t nw = 0;
    uint64_t T = 0;
    int rbits = 1, seg_bits = 1;
    bool tried_group = false;
};
constexpr uint32_t kPinWide = 112;                                  // c->pinned[112 ..]: pairs of the batch, overflow flag, runs of the sort-based path
static int wide_batch_begin(bella_ctx* c, const SpgemmArgs& sa, uint32_t nw, const uint32_t* d_cols, uint64_t T, WideBatch& wb) {
    wb.nw = nw; wb.T = T;
    ENSURE(c, c->w_f, 4 * ((size_t)nw + 2));
    ENSURE(c, c->w_off, 8 * ((size_t)nw + 2));
    ENSURE(c, c->w_nruns, 16);
    k_wide_sizes<<<nblk((uint64_t)nw + 1), 256, 0, c->stream>>>(d_cols, ptr<uint32_t>(c->flopsr), nw, ptr<uint32_t>(c->w_f));
    KCHK(c);
    int rc = scan_u32_to_u64(c, ptr<uint32_t>(c->w_f), ptr<uint64_t>(c->w_off), (uint64_t)nw + 1);
    if (rc) return rc;
    if (T >= 0x7FFF0000ull) return fail(c, BELLA_ERR_ROW_TOO_LARGE, "the columns with >= 65536 products hold %llu products together (limit 2^31)",
                                        (unsigned long long)T);
    ENSURE(c, c->w_plist, 8 * (T + 64)); ENSURE(c, c->w_scr, 2 * T);
    ENSURE(c, c->w_segfirst, 4 * ((size_t)nw + 2));
    ENSURE(c, c->w_toff, 8 * ((size_t)nw + 2));
    WideArgs& a = wb.a;
    a.cols = d_cols; a.nw = nw; a.woff = ptr<uint64_t>(c->w_off);
    a.Bptr = sa.Bptr; a.Bent = sa.Bent; a.inl = sa.inl; a.Aent = sa.Aent; a.Aent2 = sa.Aent2; a.Aov = sa.Aov; a.Arow = sa.Arow; a.roff = sa.roff; a.packed = sa.packed; a.flopptr = sa.flopptr;
    a.k = sa.k; a.binSize = sa.binSize;
    a.plist = ptr<uint2>(c->w_plist); a.plist_pad = T; a.sort_scratch = ptr<uint16_t>(c->w_scr);
    a.tmp_pairs = sa.tmp_pairs; a.tmp_ext = sa.tmp_ext; a.nnzC = sa.nnzC; a.status = sa.ctl + kCtlStatus;
    while ((1ull << wb.rbits) < (uint64_t)c->nreads) ++wb.rbits;   // a partner read id fits rbits bits
    a.rbits = (uint32_t)wb.rbits;
    while ((1u << wb.seg_bits) < nw) ++wb.seg_bits;
    a.key32 = wb.rbits + wb.seg_bits <= 32 && !(c->debug & 64u) ? 1u : 0u;   // debug bit 6: tests, 64-bit keys on any input
    a.R_first = nullptr;
    // wide columns with few partners (HiFi-like input): group in LDS, two streaming passes over the columns' products (wide.hpp) --
    // unless a column's partners do not fit the table, the lane-order self-test failed, or debug bit 12 asks for the sort-based path
    // (tests).  The products are read from the row lists; a layout without them (the default) expands the BATCH's columns into a
    // temporary list first (10 bytes per product of the batch, k_layout_rowlists on the batch's columns).
    bool lists = c->have_rowlists;
    if (!lists && c->lane_order_ok && !(c->debug & 4096u) && c->nreads <= (1u << 30)) {
        if (!ensure_bytes(c, c->w_aent2, 8 * T + 64) && !ensure_bytes(c, c->w_aov, 2 * T + 64)) {
            k_layout_rowlists<<<nw < 4096u ? nw : 4096u, kRowListBlock, 0, c->stream>>>(sa.Bptr, sa.Bent, sa.Aent, ptr<uint64_t>(c->w_off), sa.roff, d_cols, nw,
                                                                                        ptr<uint2>(c->w_aent2), ptr<uint16_t>(c->w_aov), sa.inl);
            KCHK(c);
            a.Aent2 = ptr<uint2>(c->w_aent2); a.Aov = ptr<uint16_t>(c->w_aov); a.Arow = nullptr;
            lists = true;
        } else {
            c->err.clear();                                        // (no room: the sort-based path expands for itself)
            release(c->w_aent2); release(c->w_aov);
        }
    }
    if (lists && c->lane_order_ok && !(c->debug & 4096u)) {
        ENSURE(c, c->w_gtab, sizeof(uint4) * (size_t)nw * kWideGroupSlots);
        ENSURE(c, c->w_gcount, 4 * ((size_t)nw + 4));
        ENSURE(c, c->w_gbase, 4 * ((size_t)nw + 4));
        a.gtab = ptr<uint4>(c->w_gtab); a.gcount = ptr<uint32_t>(c->w_gcount); a.gbase = ptr<uint32_t>(c->w_gbase);
        HIPCHK(c, hipMemsetAsync(a.gcount + nw, 0, 8, c->stream));          // the scan's last element and the overflow flag
        k_wide_group1<<<nw < 4096u ? nw : 4096u, kWideGroup1Block, 0, c->stream>>>(a);
        KCHK(c);
        rc = scan_u32(c, ptr<uint32_t>(c->w_gcount), ptr<uint32_t>(c->w_gbase), (uint64_t)nw + 1);
        if (rc) return rc;
        // (8 bytes into pinned memory: the pairs of the batch and the overflow flag; the slot-order tables are sized on the device)
        HIPCHK(c, hipMemcpyAsync(c->pinned + kPinWide, ptr<uint32_t>(c->w_gbase) + nw, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->pinned + kPinWide + 1, ptr<uint32_t>(c->w_gcount) + nw + 1, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipEventRecord(c->join[kNumTiers - 3], c->stream));
        // The append pass needs nothing from the host (the table images and the batch's product count): it follows at once and leaves by
        // itself if a column's partners did not fit the table (the flag); the host meanwhile waits for the totals and prepares the
        // kernels that need the pair count on a second stream.
        k_wide_group2_chunks<<<nw < 8192u ? nw : 8192u, kWideGroup2Block, 0, c->stream>>>(a);
        KCHK(c);
        wb.tried_group = true;
    }
    return 0;
}

// sync_at_end: another batch follows (its buffers may be reallocated: the kernels of this one must be done)
static int wide_batch_finish(bella_ctx* c, WideBatch& wb, bool sync_at_end) {
    WideArgs& a = wb.a;
    const uint32_t nw = wb.nw;
    const uint64_t T = wb.T;
    hipStream_t const s2 = c->side[kNumTiers - 1];
    uint32_t np = 0;
    bool grouped = false;
    int rc = 0;
    // slot-order tables (pow2 >= max(16, pairs) slots per column: at most 16 nw + 2 np in all), ranks, descriptors by class: small kernels
    auto tables_and_descriptors = [&](hipStream_t st) -> int {
        const uint64_t tab_cap = 16ull * nw + 2ull * np;
        ENSURE(c, c->w_table, 8 * tab_cap);
        a.table = ptr<uint64_t>(c->w_table); a.toff = ptr<uint64_t>(c->w_toff);
        k_wide_toff<<<1, 1024, 0, st>>>(a.seg_first, nw, ptr<uint64_t>(c->w_toff));
        KCHK(c);
        k_wide_table_fill<<<nblk(tab_cap), 256, 0, st>>>(a.table, tab_cap);
        KCHK(c);
        if (np) {
            k_wide_insert<<<nblk(np), 256, 0, st>>>(a);
            KCHK(c);
        }
        k_wide_ranks<<<nw < 1024u ? nw : 1024u, kBlock, 0, st>>>(a);
        KCHK(c);
        if (np) {
            ENSURE(c, c->w_redo, 4 * ((size_t)np + 4));
            a.redo = ptr<uint32_t>(c->w_redo);
            HIPCHK(c, hipMemsetAsync(a.redo + np, 0, 16, st));    // (the redo count and the three class counts of k_wide_desc)
            ENSURE(c, c->w_desc, 3 * sizeof(WidePairAddr) * (size_t)np);
            a.desc = (WidePairAddr*)c->w_desc.p;
            k_wide_desc<<<nblk(np), 256, 0, st>>>(a);
            KCHK(c);
        }
        return 0;
    };
    if (wb.tried_group) {
        HIPCHK(c, hipEventSynchronize(c->join[kNumTiers - 3]));      // (the totals are there; the append pass is running)
        if (c->pinned[kPinWide + 1] == 0) {
            np = c->pinned[kPinWide];
            ENSURE(c, c->w_rlen, 4 * ((size_t)np + 1)); ENSURE(c, c->w_rstart, 4 * ((size_t)np + 2)); ENSURE(c, c->w_rrank, 4 * ((size_t)np + 1));
            ENSURE(c, c->w_rfirst, 4 * ((size_t)np + 1));
            ENSURE(c, c->w_key2, 8 * ((size_t)np + 1));
            a.R_len_w = ptr<uint32_t>(c->w_rlen); a.R_start_w = ptr<uint32_t>(c->w_rstart); a.R_key_w = c->w_key2.p;
            a.R_first = ptr<uint32_t>(c->w_rfirst);
            a.R_key = c->w_key2.p; a.R_len = ptr<uint32_t>(c->w_rlen); a.R_start = ptr<uint32_t>(c->w_rstart); a.npairs = np;
            a.R_rank = ptr<uint32_t>(c->w_rrank);
            a.seg_first = ptr<uint32_t>(c->w_gbase);                 // first pair of every column = the prefix sums of the pair counts
            grouped = true;
            if (np) {
                // NEXT to the append pass (which wide_batch_begin enqueued), on a stream of their own: the pairs' keys / lists / first
                // products (from the table images) and the small kernels that only need those -- tables, slot-order insertion, ranks,
                // descriptors
                HIPCHK(c, hipStreamWaitEvent(s2, c->join[kNumTiers - 3], 0));
                k_wide_pairs<<<nw < 8192u ? nw : 8192u, kWideGroup2Block, 0, s2>>>(a);
                KCHK(c);
                rc = tables_and_descriptors(s2);
                if (rc) return rc;
                HIPCHK(c, hipEventRecord(c->join[kNumTiers - 2], s2));
                HIPCHK(c, hipStreamWaitEvent(c->stream, c->join[kNumTiers - 2], 0));
            } else {
                rc = tables_and_descriptors(c->stream);
                if (rc) return rc;
            }
        }
    }
    if (!grouped) {
        // the sort-based path: 38 bytes per product of the batch in flight (only allocated when this path runs)
        const int rbits = wb.rbits, seg_bits = wb.seg_bits;
        ENSURE(c, c->w_key, 8 * T); ENSURE(c, c->w_key2, 8 * T);
        ENSURE(c, c->w_idx, 4 * T); ENSURE(c, c->w_idx2, 4 * T);
        ENSURE(c, c->w_hv, 8 * T);
        ENSURE(c, c->w_rlen, 4 * (T + 1)); ENSURE(c, c->w_rstart, 4 * (T + 2)); ENSURE(c, c->w_rrank, 4 * T);
        a.R_first = nullptr;
        a.W_key = c->w_key.p; a.W_idx = ptr<uint32_t>(c->w_idx); a.W_rec = ptr<uint2>(c->w_hv);
        k_wide_expand<<<nw < 2048u ? nw : 2048u, kWideExpandBlock, 0, c->stream>>>(a);
        KCHK(c);
        // sort by (column, partner read) over the meaningful bits, run-length encode into pairs; u32 keys when they fit
        auto sort_and_encode = [&](auto key_tag) -> int {
            using K = decltype(key_tag);
            hipcub::DoubleBuffer<K> dk((K*)c->w_key.p, (K*)c->w_key2.p);
            hipcub::DoubleBuffer<uint32_t> dv(ptr<uint32_t>(c->w_idx), ptr<uint32_t>(c->w_idx2));
            size_t tb = 0;
            HIPCHK(c, hipcub::DeviceRadixSort::SortPairs(nullptr, tb, dk, dv, (int)T, 0, rbits + seg_bits, c->stream));
            ENSURE(c, c->cubtmp, tb);
            HIPCHK(c, hipcub::DeviceRadixSort::SortPairs(c->cubtmp.p, tb, dk, dv, (int)T, 0, rbits + seg_bits, c->stream));
            a.S_key = dk.Current(); a.S_idx = dv.Current();
            K* rkey = dk.Current() == (K*)c->w_key.p ? (K*)c->w_key2.p : (K*)c->w_key.p;
            size_t tb2 = 0;
            HIPCHK(c, hipcub::DeviceRunLengthEncode::Encode(nullptr, tb2, (const K*)a.S_key, rkey, ptr<uint32_t>(c->w_rlen), ptr<uint32_t>(c->w_nruns), (int)T, c->stream));
            ENSURE(c, c->cubtmp, tb2);
            HIPCHK(c, hipcub::DeviceRunLengthEncode::Encode(c->cubtmp.p, tb2, (const K*)a.S_key, rkey, ptr<uint32_t>(c->w_rlen), ptr<uint32_t>(c->w_nruns), (int)T, c->stream));
            a.R_key = rkey;
            return 0;
        };
        {
            const int rc2 = a.key32 ? sort_and_encode(uint32_t{}) : sort_and_encode(uint64_t{});
            if (rc2) return rc2;
        }
        HIPCHK(c, hipMemcpyAsync(c->pinned + kPinWide + 2, c->w_nruns.p, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        np = c->pinned[kPinWide + 2];
        HIPCHK(c, hipMemsetAsync(ptr<uint32_t>(c->w_rlen) + np, 0, 4, c->stream));
        rc = scan_u32(c, ptr<uint32_t>(c->w_rlen), ptr<uint32_t>(c->w_rstart), (uint64_t)np + 1);
        if (rc) return rc;
        a.R_len = ptr<uint32_t>(c->w_rlen); a.R_start = ptr<uint32_t>(c->w_rstart); a.npairs = np;
        a.R_rank = ptr<uint32_t>(c->w_rrank); a.seg_first = ptr<uint32_t>(c->w_segfirst);
        k_wide_gather<<<nblk(T), 256, 0, c->stream>>>(a, T);
        KCHK(c);
        k_wide_segments<<<nblk((uint64_t)np + 1), 256, 0, c->stream>>>(a);
        KCHK(c);
        rc = tables_and_descriptors(c->stream);
        if (rc) return rc;
    }
    if (np) {
        // one workgroup per pair (closed-form fold); what it leaves over (lists of >= 32768 products, > 16 bins) to the serial fold
#ifdef BELLA_WF_CLOCK
        static unsigned long long* d_clk = nullptr;
        if (!d_clk) (void)hipMalloc(&d_clk, 64);
        (void)hipMemsetAsync(d_clk, 0, 64, c->stream);
        a.clk = d_clk;
#endif
        // (three instances by list length, wide.hpp: the class counts are on the device, workgroups beyond a class's pairs leave at once)
        // The instances run side by side (other pairs, other lists; the redo list and its counter take atomics): the short lists on a
        // stream of their own, so that one instance's last pairs do not hold up the next one's first.
        HIPCHK(c, hipEventRecord(c->fork, c->stream));
        HIPCHK(c, hipStreamWaitEvent(s2, c->fork, 0));
        k_wide_fold_wg<kWideFoldBlockLarge, kWideFoldMid, kGridBucketsLarge, 1, 6><<<np < 16384u ? np : 16384u, kWideFoldBlockLarge, 0, c->stream>>>(a);
        KCHK(c);
        k_wide_fold_wg<kWideFoldBlockSmall, kWideFoldSmall, 2048, 0, 6><<<np < 16384u ? np : 16384u, kWideFoldBlockSmall, 0, s2>>>(a);
        KCHK(c);
        k_wide_fold_wg<kWideFoldBlockLarge, kWideFoldLdsLarge, kGridBucketsLarge, 2, 4><<<np < 4096u ? np : 4096u, kWideFoldBlockLarge, 0, c->stream>>>(a);
        KCHK(c);
        HIPCHK(c, hipEventRecord(c->join[kNumTiers - 1], s2));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->join[kNumTiers - 1], 0));
#ifdef BELLA_WF_CLOCK
        {
            unsigned long long h[8];
            (void)hipMemcpyAsync(h, d_clk, 64, hipMemcpyDeviceToHost, c->stream);
            (void)hipStreamSynchronize(c->stream);
            unsigned long long tot = 0;
            for (int i = 0; i < 8; ++i) tot += h[i];
            std::fprintf(stderr, "[wf clock] pairs %u:", np);
            for (int i = 0; i < 8; ++i) std::fprintf(stderr, " p%d %.1f%%", i, 100.0 * (double)h[i] / (double)tot);
            std::fprintf(stderr, " | cycles per pair %.0f\n", (double)tot / np);
        }
#endif
        k_wide_fold<<<nblk(np, 64), 64, 0, c->stream>>>(a);
        KCHK(c);
    }
    if (sync_at_end) HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}
static int run_wide_batch(bella_ctx* c, const SpgemmArgs& sa, uint32_t nw, const uint32_t* d_cols, uint64_t T, bool sync_at_end) {
    WideBatch wb;
    int rc = wide_batch_begin(c, sa, nw, d_cols, T, wb);
    if (rc) return rc;
    return wide_batch_finish(c, wb, sync_at_end);
}

// batches of wide columns holding at most `budget` products each (38 bytes of HBM per product in flight)
// Tall: the products of all nw columns (known from the symbolic kernels' control block): when they fit one batch -- the usual case -- the
// columns' sizes need not come to the host at all
static int run_wide(bella_ctx* c, const SpgemmArgs& sa, uint32_t nw, const uint32_t* d_cols, uint64_t Tall) {
    if (Tall && Tall <= c->wide_budget) return run_wide_batch(c, sa, nw, d_cols, Tall, false);
    ENSURE(c, c->w_f, 4 * ((size_t)nw + 2));
    k_wide_sizes<<<nblk((uint64_t)nw + 1), 256, 0, c->stream>>>(d_cols, ptr<uint32_t>(c->flopsr), nw, ptr<uint32_t>(c->w_f));
    KCHK(c);
    std::vector<uint32_t> wf((size_t)nw + 1);
    HIPCHK(c, hipMemcpyAsync(wf.data(), c->w_f.p, 4 * ((size_t)nw + 1), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const uint64_t budget = c->wide_budget;
    for (uint32_t b = 0; b < nw;) {
        uint64_t t = wf[b];
        uint32_t e = b + 1;
        while (e < nw && t + wf[e] <= budget) t += wf[e++];
        int rc = run_wide_batch(c, sa, e - b, d_cols + b, t, e < nw);
        if (rc) return rc;
        b = e;
    }
    return 0;
}

// The pass's partition must lie inside the one the layout was built for (B' only holds the owned columns): a whole layout serves any
// partition; otherwise the layout is rebuilt from the resident B for the partition now set.
static int layout_for_partition(bella_ctx* c) {
    if (c->layout_stride == 1 || (c->layout_stride == c->part_stride && c->layout_first == c->part_first)) return 0;
    const bella_timings keep = c->tm;
    int rc = build_layout(c);
    const float lm = c->tm.layout_ms, em = c->tm.expand_ms;
    c->tm = keep;
    c->tm.layout_ms = lm; c->tm.expand_ms = em;
    return rc;
}

static int run_spgemm(bella_ctx* c, const bella_params* p, uint32_t* status_out, int depth = 0) {
    const uint32_t nr = c->nreads;
    const bool force_global = (c->debug & 1u) != 0 || c->tune_row_path == 1 || !c->lane_order_ok;   // (self-test failed: every column on the repairing path)
    const bool want_ext = (c->debug & 2u) == 0;
    const uint64_t ws_stride = (row_mem_bytes(65535, 65535, false) + 255) & ~(size_t)255;
    ENSURE(c, c->flopsr, 4 * ((size_t)nr + 2));
    ENSURE(c, c->flopptr, 8 * ((size_t)nr + 2));
    ENSURE(c, c->nnzC, 4 * ((size_t)nr + 2));
    ENSURE(c, c->colptrC, 8 * ((size_t)nr + 2));
    ENSURE(c, c->rowlists, (16 * (size_t)kNumTiers + 4) * nr + 64);   // tier descriptor lists, then the list of wide columns
    ENSURE(c, c->tiercaps, 4 * kNumTiers);
    ENSURE(c, c->ctl, 4 * kCtlWords);
    ENSURE(c, c->ws, ws_stride * kGlobalGrid);
    ENSURE(c, c->retry, 4 * ((size_t)nr + 1));
    ENSURE(c, c->orderlist, 4 * ((size_t)nr + 1));
    ENSURE(c, c->order_ws, kOrderWsBytes * kOrderBigGrid);
    if (!c->order_attr) {
        HIPCHK(c, hipFuncSetAttribute((const void*)k_order_block, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)6 * kOrderLdsHt)));
        c->order_attr = true;
    }
    {   // temp storage of the pass's one scan (colptrC), so that nothing is allocated between the kernels
        size_t tb1 = 0;
        hipcub::TransformInputIterator<uint64_t, CountU64, const uint32_t*> it(ptr<uint32_t>(c->nnzC), CountU64());
        HIPCHK(c, hipcub::DeviceScan::ExclusiveSum(nullptr, tb1, it, ptr<uint64_t>(c->colptrC), (int)nr + 1, c->stream));
        ENSURE(c, c->cubtmp, tb1 + 256);
    }

    uint32_t caps[kNumTiers] = {};
    const bool half_tables = !(c->pair_ratio1024 * 5 < 1024);
    const uint32_t* tier_caps = (half_tables && !c->tiers_from_env) ? kTierCapsHalf : c->tier_caps;
    uint32_t g_ntiers = c->ntiers;
    if (!c->tiers_from_env)
        for (g_ntiers = 1; tier_caps[g_ntiers - 1] != 65535; ++g_ntiers) {}
    // Long-list inputs (at most one pair in 64 products on the assembly-time sample: HiFi-like sets): the fold of a pair is
    // quadratic in its list inside the row kernels and has a position grid on the sort-based path, so the larger LDS tiers are
    // closed and their columns take that path with the columns above the tiers.
    // (either every sampled column is like that, or the sample as a whole: since round 6 the path above the tiers is the faster one for
    // such columns even when a few short-list columns stand among them -- HiFi-like 10k reads, -u 40: 6.35 -> 5.9 ms per pass)
    const bool long_lists = !force_global && !c->tiers_from_env && (c->pair_ratio1024 < 16 || c->long_list_sample);
    for (uint32_t t = 0; t < g_ntiers; ++t)
        caps[t] = (force_global || (long_lists && tier_caps[t] > kLongListMaxCap)) && t + 1 < g_ntiers ? 0 : tier_caps[t];
    const int want_state = (force_global ? 2 : 1) + (half_tables ? 2 : 0) + (long_lists ? 4 : 0);
    const uint64_t psig[6] = {c->layout_gen, ((uint64_t)c->part_first << 32) | c->part_stride, ((uint64_t)c->range_lo << 32) | c->range_hi, nr,
                              (uint64_t)want_state, (uint64_t)g_ntiers};
    const bool warm = c->pass_known && std::memcmp(psig, c->pass_sig, sizeof(psig)) == 0;
    // Rarely needed kernels: the rerun of columns whose pair count overflowed an LDS tier's key table or whose lists came out of
    // order (list and count produced on the device), and the serial fold of pairs with > 16 final bins.  Whether a pass needs them
    // is again a function of the operands: a warm pass whose predecessor needed neither leaves them out and checks the counters
    // in the final control block (should they be non-zero after all, the pass is run again with them).
    const bool skip_rare = warm && c->pass_rare_free && !(c->debug & 4u);
    const bool skip_big = warm && c->pass_big_free && !(c->debug & 4u);
    uint32_t launches = 0;
    int rc = 0;
    uint32_t* const ctl_host = c->pinned + 32;
#define EVREC(i) HIPCHK(c, hipEventRecord(c->ev[i], c->stream))
    struct Launch { int lo, hi; size_t lds; uint32_t rows; int cls; };
    Launch ln[kNumTiers];
    int nl = 0;
    auto enqueue = [&]() -> int {
    EVREC(2);
    if (c->caps_state != want_state) {                         // the tier caps only change with the operands or the debug switch
        HIPCHK(c, hipMemcpyAsync(c->tiercaps.p, caps, sizeof(caps), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));            // caps[] is a stack array
        c->caps_state = want_state;
    }
    // one control block per pass (counters, tier lengths, status, totals): one fill, one read back
    uint32_t* d_ctl = ptr<uint32_t>(c->ctl);
    // the columns of this context: i0 + j * stride, j < nown.  The symbolic kernels only visit those; the entries of all other
    // columns in flops / nnzC are zero from the one clear made when the operands, the partition or the stage changed.
    const uint32_t hi = c->range_hi < nr ? c->range_hi : nr;
    uint32_t i0 = c->range_lo + (c->part_first + c->part_stride - c->range_lo % c->part_stride) % c->part_stride;
    const uint32_t nown = i0 < hi ? (hi - i0 + c->part_stride - 1) / c->part_stride : 0;
    if (!nown) i0 = 0;
    {
        const uint64_t sig[6] = {(uint64_t)(uintptr_t)c->flopsr.p, (uint64_t)(uintptr_t)c->nnzC.p, c->layout_gen, ((uint64_t)c->part_first << 32) | c->part_stride,
                                 ((uint64_t)c->range_lo << 32) | c->range_hi, nr};
        if (std::memcmp(sig, c->sym_sig, sizeof(sig)) != 0) {
            HIPCHK(c, hipMemsetAsync(c->flopsr.p, 0, 4 * ((size_t)nr + 2), c->stream));
            HIPCHK(c, hipMemsetAsync(c->nnzC.p, 0, 4 * ((size_t)nr + 2), c->stream));
            std::memcpy(c->sym_sig, sig, sizeof(sig));
        }
    }
    // (the control block is cleared by the first kernel of the pass)
    if (nown && c->have_rowlists) {
        k_row_flops_rl<<<nblk(nown), kBlock, 0, c->stream>>>(ptr<uint64_t>(c->Arow), i0, c->part_stride, nown, ptr<uint32_t>(c->flopsr), ptr<uint32_t>(c->nnzC), d_ctl);
        KCHK(c);
    } else if (nown) {
        k_row_flops<<<nblk(nown), kBlock, 0, c->stream>>>(ptr<uint32_t>(c->rowF), i0, c->part_stride, nown,
                                                          ptr<uint32_t>(c->flopsr), ptr<uint32_t>(c->nnzC), d_ctl);
        KCHK(c);
    } else {
        HIPCHK(c, hipMemsetAsync(d_ctl, 0, 4 * kCtlWords, c->stream));
    }
    k_tier_lists<<<nblk(nown ? nown : 1), kBlock, 0, c->stream>>>(ptr<uint32_t>(c->flopsr), nr, i0, c->part_stride, nown, ptr<uint32_t>(c->tiercaps), g_ntiers,
                                                  layout_bptr(c), ptr<uint64_t>(c->roff), ptr<uint4>(c->rowlists),
                                                  (uint32_t*)(ptr<uint4>(c->rowlists) + (size_t)kNumTiers * nr), d_ctl + kCtlTierCnt,
                                                  (unsigned long long*)(d_ctl + kCtlTotals) + 1, ptr<uint64_t>(c->flopptr),
                                                  c->have_rowlists ? ptr<uint64_t>(c->Arow) : nullptr);
    KCHK(c);
    // The host needs two things from the symbolic kernels before it can launch the row kernels: the tiers' lengths (exact grids)
    // and the product total (buffer sizes).  Both are functions of the operands, the partition and the stage only, so they are
    // read back (96 bytes into pinned memory, one host round trip) the first time a pass runs on them and remembered; later
    // passes on the same operands enqueue everything without waiting.  The final control block is checked against them.
    uint32_t* const tcnt = c->pass_tcnt;
    if (!warm) {
        c->pass_known = false;
        HIPCHK(c, hipMemcpyAsync(c->pinned, d_ctl + kCtlTierCnt, 4 * (kCtlTotals + 8 - kCtlTierCnt), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        std::memcpy(c->pass_tcnt, c->pinned, sizeof(c->pass_tcnt));
        std::memcpy(&c->pass_products, c->pinned + (kCtlTotals - kCtlTierCnt) + 2, 8);
        std::memcpy(c->pass_wide_products, c->pinned + (kCtlTotals - kCtlTierCnt) + 4, 16);
        std::memcpy(c->pass_sig, psig, sizeof(psig));
    }
    // the product-sized buffers follow THIS pass's product count (a column partition or a stage only pays for its share; the
    // reference sizes its stages from the same number, overlap.hpp:365-404,682-710); they only ever grow
    const uint64_t Fub = c->pass_products;
    ENSURE(c, c->tmp_pairs, sizeof(bella_pair) * Fub);
    if (want_ext) ENSURE(c, c->tmp_ext, sizeof(bella_pair_ext) * Fub);
    ENSURE(c, c->plist_hv, 8 * Fub);
    ENSURE(c, c->overflow, 16 * ((Fub >> 4) + 64));   // a pair on this list has > 16 products
    ENSURE(c, c->sortscr, 2 * Fub);
    ENSURE(c, c->pairs, sizeof(bella_pair) * Fub);     // nnz(C) <= products
    if (want_ext) ENSURE(c, c->ext, sizeof(bella_pair_ext) * Fub);
    EVREC(3);

    SpgemmArgs a;
    a.Bptr = layout_bptr(c);
    a.Bent = ptr<uint2>(c->Bent);
    a.inl = c->layout_inline;
    a.Aent = ptr<uint2>(c->Aent);
    a.Aent2 = c->have_rowlists ? ptr<uint2>(c->Aent2) : nullptr;
    a.Aov = c->have_rowlists ? ptr<uint16_t>(c->Aov) : nullptr;
    a.Arow = c->have_rowlists ? ptr<uint64_t>(c->Arow) : nullptr;
    a.roff = ptr<uint64_t>(c->roff);
    a.packed = ptr<uint32_t>(c->packed);
    a.flopptr = ptr<uint64_t>(c->flopptr);
    a.flops = ptr<uint32_t>(c->flopsr);
    a.tmp_pairs = ptr<bella_pair>(c->tmp_pairs);
    a.tmp_ext = want_ext ? ptr<bella_pair_ext>(c->tmp_ext) : nullptr;
    a.nnzC = ptr<uint32_t>(c->nnzC);
    a.plist = ptr<uint2>(c->plist_hv);
    a.overflow = ptr<uint4>(c->overflow);
    a.ctl = ptr<uint32_t>(c->ctl);
    a.retry = ptr<uint32_t>(c->retry);
    a.nrows_dev = nullptr;
    a.ws = ptr<uint8_t>(c->ws);
    a.ws_stride = ws_stride;
    a.nrows = 0;
    a.k = p->kmer_size;
    a.binSize = p->bin_size;
    a.inject_unordered = (c->debug & 4u) ? 1 : 0;
#ifdef BELLA_DEV_PROF
    ENSURE(c, c->prof, 8 * 10 * kNumTiers);
    HIPCHK(c, hipMemsetAsync(c->prof.p, 0, 8 * 10 * kNumTiers, c->stream));
    a.prof = nullptr;
    a.stop_phase = getenv("BELLA_DEV_STOP") ? atoi(getenv("BELLA_DEV_STOP")) : -1;
#endif
    // LDS classes: consecutive tiers whose workgroups take the same share of a CU; one launch per class (largest class first, on
    // side streams so that the classes overlap; the class of the smallest columns, which finishes last, on the main stream), and
    // the global-workspace tier.
    auto dcap_of = [&](uint32_t cap) -> uint32_t {
        const bool half = (half_tables || (c->debug & 16u)) && cap <= kHalfTableMaxCap;     // debug bit 4: tests
        // pair-rich inputs: the smallest columns are almost all chance pairs (one product each): their tier gets a full-size table
        if (half_tables && cap <= kFullTableMaxCap) return cap;
        return half ? cap / 2 : cap / 4;
    };
    nl = 0;
    {
        int cls_prev = -1;
        for (int t = 0; t + 1 < (int)g_ntiers; ++t) {
            const size_t lds = row_mem_bytes(tier_caps[t], dcap_of(tier_caps[t]), true);
            int cls = (int)kNumClasses - 1;
            for (int k2 = 0; k2 < (int)kNumClasses; ++k2)
                if (lds <= kClassLds[k2]) { cls = k2; break; }
            if (cls != cls_prev || nl == 0) { ln[nl++] = Launch{t, t, lds, 0, cls}; cls_prev = cls; }
            ln[nl - 1].hi = t;
            ln[nl - 1].lds = lds;                                 // the class is carved for its largest tier
            ln[nl - 1].rows += tcnt[t];
        }
    }
    // Streams: at most four run concurrently on this device (hardware queues), and a workgroup of the whole-CU class only starts
    // on an EMPTY CU.  So the whole-CU class (few columns, if any) is launched first, on side stream 0, and takes its CUs before the
    // others arrive; then the next classes go to the remaining side streams, the global-workspace tier to side stream 2, and the
    // class of the smallest columns (it finishes last) stays on the main stream.
    int main_l = -1;
    for (int l = 0; l < nl; ++l) if (ln[l].rows) { main_l = l; break; }
    // The columns above the LDS tiers (<= 65,535 products): a handful runs on the global-workspace path next to the LDS classes; when
    // a pass has many of them (HiFi-like inputs, deep coverage) they join the wide columns on the sort-based path of wide.hpp, which
    // works at HBM speed instead of L2 latency (debug bit 5: tests, any number of them).
    const uint32_t n_mid = tcnt[g_ntiers - 1];
    const bool mid_to_wide = !force_global && n_mid != 0 && (n_mid >= kMidToWideMin || (c->debug & 32u));
    const bool global_tier = n_mid != 0 && !mid_to_wide;
    bool used[3] = {};
    auto side_stream = [&](int which) -> int {                    // first use: make the stream wait for what the main stream did so far
        // (the fork point is ev[3], recorded on the main stream right after the symbolic kernels: no event of its own)
        if (!used[which]) { HIPCHK(c, hipStreamWaitEvent(c->side[which], c->ev[3], 0)); used[which] = true; }
        return 0;
    };
    a.rowlist = nullptr;
    a.nrows_dev = nullptr;
    a.nreads = nr;
    int nside = 0;                                                // classes launched off the main stream so far
    auto launch_class = [&](int l, bool on_main) -> int {
        hipStream_t sst = c->stream;
        if (!on_main) {
            const int which = nside;                              // 0, 1, 2: one class each
            int r2 = side_stream(which);
            if (r2) return r2;
            sst = c->side[which];
            nside++;
        }
        a.rowlist = nullptr;
        a.nrows_dev = nullptr;
        a.rowdesc = ptr<uint4>(c->rowlists);
        a.tier_lo = (uint32_t)ln[l].lo;
        a.tier_hi = (uint32_t)ln[l].hi;
        a.tcount_valid = ln[l].hi - ln[l].lo <= 3 ? 1u : 0u;
        for (int k2 = 0; k2 < 4; ++k2) a.tcount[k2] = ln[l].lo + k2 <= ln[l].hi ? tcnt[ln[l].lo + k2] : 0u;
        a.nrows = ln[l].rows;
        a.cap = tier_caps[ln[l].hi];
        a.dcap = dcap_of(a.cap);
        const size_t lds = ln[l].lds;
        const int blk = (c->debug & 8u) ? 512 : kClassBlock[ln[l].cls];     // debug bit 3: tests, 512 threads everywhere
        int ki;
        void (*kern)(SpgemmArgs);
        const bool rl = c->have_rowlists;
#define BELLA_ROWS_KERN(NX, B) (rl ? k_spgemm_rows_lds<NX, B, true> : k_spgemm_rows_lds<NX, B, false>)
        if (blk == 512) {
            ki = a.cap <= 8 * 512 ? 0 : a.cap <= 16 * 512 ? 1 : 2;
            kern = ki == 0 ? BELLA_ROWS_KERN(8, 512) : ki == 1 ? BELLA_ROWS_KERN(16, 512) : BELLA_ROWS_KERN(22, 512);
        } else if (blk == 1024) {
            ki = a.cap <= 4 * 1024 ? 3 : a.cap <= 8 * 1024 ? 4 : 5;
            kern = ki == 3 ? BELLA_ROWS_KERN(4, 1024) : ki == 4 ? BELLA_ROWS_KERN(8, 1024) : BELLA_ROWS_KERN(11, 1024);
        } else if (blk == 256 && a.cap <= 6 * 256) {              // cap <= 1394 <= 6 * 256
            ki = 6;
            kern = BELLA_ROWS_KERN(6, 256);
        } else if (blk == 256) {                                  // cap <= 2752 <= 11 * 256
            ki = 7;
            kern = BELLA_ROWS_KERN(11, 256);
        } else if (blk == 128) {                                  // cap <= 689 <= 6 * 128
            ki = 8;
            kern = BELLA_ROWS_KERN(6, 128);
        } else if (a.cap <= 11 * 64) {                            // ONE wavefront per column (no workgroup barriers): cap <= 689 <= 11 * 64
            ki = 18;
            kern = BELLA_ROWS_KERN(11, 64);
        } else {                                                  // cap <= 1394 <= 22 * 64
            ki = 19;
            kern = BELLA_ROWS_KERN(22, 64);
        }
#undef BELLA_ROWS_KERN
        if (rl) ki += ki >= 18 ? 2 : 9;
        if (lds > c->lds_attr[ki]) {                             // once per kernel and size, not per launch
            HIPCHK(c, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            c->lds_attr[ki] = lds;
        }
#ifdef BELLA_DEV_PROF
        a.prof = (unsigned long long*)c->prof.p + 10 * l;
        {   // development builds only: leave classes out (timing experiments; the results are then incomplete)
            static const int skip = getenv("BELLA_DEV_SKIP_CLASSES") ? atoi(getenv("BELLA_DEV_SKIP_CLASSES")) : 0;
            if (skip & (1 << l)) { a.prof = nullptr; return 0; }
        }
#endif
        kern<<<ln[l].rows, blk, lds, sst>>>(a);
#ifdef BELLA_DEV_PROF
        a.prof = nullptr;
#endif
        KCHK(c);
        launches++;
        return 0;
    };
    // The wide path's first part goes BEFORE the row kernels when it is a large part of the pass (long-list inputs: HiFi-like reads with
    // a raised -u): its small kernels and the first grouping pass start at once instead of behind the classes' queues (five streams on
    // four hardware queues: every small kernel waited ~40 us for its turn, 0.3 ms before the first grouping pass began), the rest
    // follows when the classes are enqueued.  Otherwise (a few wide columns in a pass of LDS columns) the classes go first as before.
    c->n_wide = tcnt[g_ntiers] + (mid_to_wide ? n_mid : 0);
    uint32_t* const widelist = (uint32_t*)(ptr<uint4>(c->rowlists) + (size_t)kNumTiers * nr);
    const uint64_t wide_products = c->pass_wide_products[1] + (mid_to_wide ? c->pass_wide_products[0] : 0);
    auto wide_columns = [&]() -> int {                             // (the columns above the LDS tiers join the wide columns' list)
        if (mid_to_wide) {
            k_desc_cols<<<nblk(n_mid), 256, 0, c->stream>>>(ptr<uint4>(c->rowlists) + (size_t)(g_ntiers - 1) * nr, n_mid, widelist + tcnt[g_ntiers]);
            if (hipGetLastError() != hipSuccess) return fail(c, BELLA_ERR_HIP, "k_desc_cols launch failed");
        }
        return 0;
    };
    WideBatch wide_batch;
    bool wide_begun = false;
    if (c->n_wide && wide_products && wide_products <= c->wide_budget && wide_products * 4 >= c->pass_products) {
        hipStream_t const main_stream = c->stream, wst = c->side[kNumTiers];
        HIPCHK(c, hipStreamWaitEvent(wst, c->ev[3], 0));
        c->stream = wst;
        int rw = wide_columns();
        if (!rw) rw = wide_batch_begin(c, a, c->n_wide, widelist, wide_products, wide_batch);
        c->stream = main_stream;
        if (rw) { (void)hipStreamSynchronize(wst); return rw; }
        wide_begun = true;
    }
    auto whole_cu = [&](int l) { return 2 * ln[l].lds > kClassLds[kNumClasses - 1]; };
    // the whole-CU class (few columns, if any) goes FIRST, on a side stream of its own: its workgroups take their CUs before the other
    // classes' workgroups arrive, and those fill the rest of the chip next to it instead of waiting behind it
    int nwhole = 0;
    for (int l = nl - 1; l >= 0; --l)
        if (ln[l].rows && whole_cu(l) && l != main_l && nwhole == 0) { rc = launch_class(l, false); if (rc) return rc; nwhole++; }
        else if (ln[l].rows && whole_cu(l) && l != main_l) { rc = launch_class(l, true); if (rc) return rc; }
    if (global_tier) {                                            // the columns above the LDS tiers (or all of them: debug bit 0)
        hipStream_t sst = c->stream;
        if (main_l >= 0) { rc = side_stream(2); if (rc) return rc; sst = c->side[2]; }      // (launch_class uses sides 0, 1 then)
        a.rowdesc = ptr<uint4>(c->rowlists) + (size_t)(g_ntiers - 1) * nr;
        a.nrows = tcnt[g_ntiers - 1];
        a.cap = tier_caps[g_ntiers - 1];
        a.dcap = a.cap;
        const unsigned grid = a.nrows < kGlobalGrid ? a.nrows : kGlobalGrid;
        if (c->have_rowlists) k_spgemm_rows_global<true><<<grid, kGlobalBlock, 0, sst>>>(a);
        else k_spgemm_rows_global<false><<<grid, kGlobalBlock, 0, sst>>>(a);
        KCHK(c);
        launches++;
    }
    {   // the three largest remaining classes each get a side stream; what is left (the classes of the smallest columns) runs on
        // the main stream, larger class first
        int side_left = 3 - nwhole - (global_tier ? 1 : 0);
        for (int l = nl - 1; l >= 0; --l) {
            if (!ln[l].rows || (whole_cu(l) && l != main_l)) continue;
            int later = 0;                                        // non-empty classes below this one
            for (int m2 = 0; m2 < l; ++m2) later += ln[m2].rows ? 1 : 0;
            const bool on_main = l == main_l || side_left == 0 || later == 0;
            if (!on_main) side_left--;
            rc = launch_class(l, on_main);
            if (rc) return rc;
        }
    }
    // The columns of the sort-based path (wide.hpp) are other columns than the LDS classes': the path -- host-driven, several
    // synchronisations -- runs on a stream of its own NEXT to the class launches above instead of after them (HiFi-like input with a
    // raised -u, 10k reads: the classes take 1.35 ms of a pass whose wide path takes 8.7)
    if (c->n_wide) {
        hipStream_t const main_stream = c->stream, wst = c->side[kNumTiers];
        c->stream = wst;                                           // (everything run_wide enqueues, allocates for and waits on)
        int rw = 0;
        if (wide_begun) rw = wide_batch_finish(c, wide_batch, false);
        else {
            HIPCHK(c, hipStreamWaitEvent(wst, c->ev[3], 0));
            rw = wide_columns();
            if (!rw) rw = run_wide(c, a, c->n_wide, widelist, wide_products);
        }
        c->stream = main_stream;
        if (rw) { (void)hipStreamSynchronize(wst); return rw; }
        HIPCHK(c, hipEventRecord(c->join[kNumTiers], wst));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->join[kNumTiers], 0));
    }
    for (int w = 0; w < 3; ++w)
        if (used[w]) {
            HIPCHK(c, hipEventRecord(c->join[w], c->side[w]));
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->join[w], 0));
        }
    FoldArgs fa;
    fa.ctl = a.ctl;
    fa.overflow = a.overflow;
    fa.flopptr = a.flopptr;
    fa.plist = a.plist;
    fa.roff = a.roff;
    fa.packed = a.packed;
    fa.tmp_pairs = a.tmp_pairs;
    fa.tmp_ext = a.tmp_ext;
    fa.sort_scratch = ptr<uint16_t>(c->sortscr);
    fa.k = a.k;
    fa.binSize = a.binSize;
    auto rerun_columns = [&]() -> int {
        a.rowlist = ptr<uint32_t>(c->retry);
        a.rowdesc = nullptr;
        a.nrows_dev = ptr<uint32_t>(c->ctl) + kCtlRetry;
        if (c->have_rowlists) k_spgemm_rows_global<true><<<kRerunGrid, kGlobalBlock, 0, c->stream>>>(a);
        else k_spgemm_rows_global<false><<<kRerunGrid, kGlobalBlock, 0, c->stream>>>(a);   // (the list is usually empty or a handful of columns)
        KCHK(c);
        return 0;
    };
    auto finish = [&]() -> int {                                  // colptrC, slot order + move to the final place, the pass's host round trip
        if ((uint64_t)nr + 1 <= kScanSingleMax) {
            k_scan_counts<<<1, 1024, 0, c->stream>>>(ptr<uint32_t>(c->nnzC), ptr<uint64_t>(c->colptrC), nr + 1);
            KCHK(c);
        } else {
            hipcub::TransformInputIterator<uint64_t, CountU64, const uint32_t*> it(ptr<uint32_t>(c->nnzC), CountU64());
            size_t tb = c->cubtmp.cap;
            HIPCHK(c, hipcub::DeviceScan::ExclusiveSum(c->cubtmp.p, tb, it, ptr<uint64_t>(c->colptrC), (int)nr + 1, c->stream));
        }
        EVREC(6);
        if (nr) {
            OrderArgs oa;
            oa.flopptr = ptr<uint64_t>(c->flopptr); oa.colptrC = ptr<uint64_t>(c->colptrC); oa.nnzC = ptr<uint32_t>(c->nnzC);
            oa.flops = ptr<uint32_t>(c->flopsr);
            oa.nreads = nr; oa.i0 = i0; oa.stride = c->part_stride; oa.nown = nown;
            oa.tmp_pairs = ptr<bella_pair>(c->tmp_pairs); oa.tmp_ext = want_ext ? ptr<bella_pair_ext>(c->tmp_ext) : nullptr;
            oa.pairs = ptr<bella_pair>(c->pairs); oa.ext = want_ext ? ptr<bella_pair_ext>(c->ext) : nullptr;
            oa.totals = (uint64_t*)(d_ctl + kCtlTotals);
            oa.nbig = d_ctl + kCtlOrderBig; oa.biglist = ptr<uint32_t>(c->orderlist); oa.ws = ptr<uint8_t>(c->order_ws);
            // two instances share the columns by table size (the small one keeps more columns in flight per CU: tuned at 100k reads);
            // on a small pass the second launch costs more than it buys (10k reads: two launches and the gap between them for 0.3 M
            // records), so one instance takes all tables up to 1,024 slots there
            if (nown <= kOrderOneLaunchMax) {
                k_order_wave<kOrderWaveHt, 0><<<nblk(nown ? nown : 1, kOrderBlock / 64), kOrderBlock, 0, c->stream>>>(oa);
                KCHK(c);
            } else {
                k_order_wave<512, 0><<<nblk(nown ? nown : 1, kOrderBlock / 64), kOrderBlock, 0, c->stream>>>(oa);
                KCHK(c);
                k_order_wave<kOrderWaveHt, 512><<<nblk(nown ? nown : 1, kOrderBlock / 64), kOrderBlock, 0, c->stream>>>(oa);
                KCHK(c);
            }
            // columns with more than 1,024 pairs: a warm pass whose predecessor had none leaves the kernel out and checks the count in the
            // final control block (like the rerun and overflow kernels above)
            if (!skip_big) {
                k_order_block<<<kOrderBigGrid, kOrderBlock, (size_t)6 * kOrderLdsHt, c->stream>>>(oa);
                KCHK(c);
            }
        }
        EVREC(7);
        HIPCHK(c, hipMemcpyAsync(ctl_host, d_ctl, 4 * kCtlWords, hipMemcpyDeviceToHost, c->stream));
        return 0;
    };
    if (!skip_rare) { rc = rerun_columns(); if (rc) return rc; }
    EVREC(5);
    // the row kernels fold every pair themselves; only pairs that end with > 16 bins are left (their count stays on the device)
    if (!skip_rare) { k_fold_overflow<<<256, 64, 0, c->stream>>>(fa); KCHK(c); EVREC(8); }
    return finish();
    };   // enqueue
#undef EVREC
    // (Capturing the warm pass into a hipGraph and replaying it was measured: 0.54 ms per step against 0.41 ms for these direct
    // launches at 10k reads -- the multi-branch graph replays slower on ROCm 7.2 than the host can issue the launches.)
    rc = enqueue();
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if ((skip_rare && (ctl_host[kCtlRetry] || ctl_host[kCtlOverflow])) || (skip_big && ctl_host[kCtlOrderBig])) {   // the pass needed the kernels it left out: once more, with them
        c->pass_rare_free = false;
        c->pass_big_free = false;
        if (depth > 0) return fail(c, BELLA_ERR_STATE, "internal: rare kernels still skipped");
        return run_spgemm(c, p, status_out, depth + 1);
    }
    const uint32_t* const tcnt = c->pass_tcnt;
    const uint64_t Fub = c->pass_products;
    uint64_t P = 0, F = 0;
    std::memcpy(&P, ctl_host + kCtlTotals, 8);
    std::memcpy(&F, ctl_host + kCtlTotals + 2, 8);
    rc = status_to_error(c, ctl_host[kCtlStatus] & ~1u);
    if (rc) return rc;
    *status_out = ctl_host[kCtlStatus];
    c->n_overflow = ctl_host[kCtlOverflow];
    c->n_retry = ctl_host[kCtlRetry];
    c->npairs = P;
    c->flops = F;
    if (F != Fub || std::memcmp(ctl_host + kCtlTierCnt, tcnt, 4 * (g_ntiers + 1)) != 0) {
        c->pass_known = false;
        return fail(c, BELLA_ERR_STATE, "internal: product total %llu / tier lengths differ from the ones the pass was sized with (%llu)",
                    (unsigned long long)F, (unsigned long long)Fub);
    }
    c->pass_known = true;
    c->pass_rare_free = c->n_retry == 0 && c->n_overflow == 0 && c->n_wide == 0;
    c->pass_big_free = ctl_host[kCtlOrderBig] == 0;
    c->tm.symbolic_ms = ev_ms(c->ev[2], c->ev[3]);
    c->tm.spgemm_ms = ev_ms(c->ev[3], c->ev[5]);
    c->tm.fold_ms = skip_rare ? 0.f : ev_ms(c->ev[5], c->ev[8]);
    c->tm.compact_ms = ev_ms(c->ev[6], c->ev[7]);
    c->tm.overlap_total_ms = ev_ms(c->ev[2], c->ev[7]);
    c->tm.spgemm_launches = launches;
    c->tm.retry_columns = c->n_retry;
    c->tm.overflow_pairs = c->n_overflow;
#ifdef BELLA_DEV_PROF
    {
        unsigned long long ph[10 * kNumTiers];
        HIPCHK(c, hipMemcpy(ph, c->prof.p, sizeof(ph), hipMemcpyDeviceToHost));
        static const char* nm[9] = {"expand", "gather+insert", "count-scan", "scatter+singles", "rank/overlay", "parents", "walks", "emit", "head (descriptor + first entries + table init)"};
        for (int l = 0; l < nl; ++l) {
            const unsigned long long* q = ph + 10 * l;
            if (!q[9]) continue;
            double tot = 1e-9;
            for (int n = 0; n < 9; ++n) tot += (double)q[n];
            fprintf(stderr, "[bella_hip prof] class cap %u (%llu columns, %.0f cycles each):", tier_caps[ln[l].hi], q[9], tot / (double)q[9]);
            for (int n = 0; n < 9; ++n) fprintf(stderr, " %s %.1f%%", nm[n], 100.0 * (double)q[n] / tot);
            fprintf(stderr, "\n");
        }
    }
#endif
    return 0;
}

// the columns of this context under the current partition and stage: i0 + j * stride, j < nown
static uint32_t owned_columns(const bella_ctx* c, uint32_t* i0_out) {
    const uint32_t nr = c->nreads;
    const uint32_t hi = c->range_hi < nr ? c->range_hi : nr;
    uint32_t i0 = c->range_lo + (c->part_first + c->part_stride - c->range_lo % c->part_stride) % c->part_stride;
    const uint32_t nown = i0 < hi ? (hi - i0 + c->part_stride - 1) / c->part_stride : 0;
    if (!nown) i0 = 0;
    if (i0_out) *i0_out = i0;
    return nown;
}

int bella_hip_overlap(bella_ctx* c, const bella_params* p, uint64_t* npairs, uint64_t* flops) {
    if (!c) return BELLA_ERR_BAD_ARG;
    if (!c->have_matrix) return fail(c, BELLA_ERR_STATE, "set_B / assemble_tuples first");
    int rc = check_params(c, p);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    rc = layout_for_partition(c);
    if (rc) return rc;
    uint32_t st = 0;
    rc = run_spgemm(c, p, &st);
    if (rc == BELLA_ERR_NOMEM && c->have_rowlists) {
        // the row lists are optional: give their memory to the pass and let it expand the products itself
        (void)hipDeviceSynchronize();
        release(c->Aent2); release(c->Aov); release(c->Arow);
        c->have_rowlists = false;
        c->layout_gen++;
        c->pass_known = false;
        rc = run_spgemm(c, p, &st);
    }
    if (rc) { (void)hipDeviceSynchronize(); return rc; }     // side-stream kernels of the pass may still be in flight
    if (st & 1u) return fail(c, BELLA_ERR_BINS, "internal: > 16 bins without sort scratch");
    c->have_pairs = true;
    c->have_alns = false;
    c->tm.numeric_passes++;
    c->tm.numeric_columns += owned_columns(c, nullptr);
    if (npairs) *npairs = c->npairs;
    if (flops) *flops = c->flops;
    return 0;
}

// the symbolic phase alone (spgemm.hpp: k_count_pairs)
int bella_hip_count_pairs(bella_ctx* c, const bella_params* p, uint64_t* colptrC, uint64_t* npairs, uint64_t* flops) {
    if (!c) return BELLA_ERR_BAD_ARG;
    if (!c->have_matrix) return fail(c, BELLA_ERR_STATE, "set_B / assemble_tuples first");
    int rc = check_params(c, p);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    rc = layout_for_partition(c);
    if (rc) return rc;
    const uint32_t nr = c->nreads;
    ENSURE(c, c->flopsr, 4 * ((size_t)nr + 2));
    ENSURE(c, c->nnzC, 4 * ((size_t)nr + 2));
    ENSURE(c, c->colptrC, 8 * ((size_t)nr + 2));
    ENSURE(c, c->ctl, 4 * kCtlWords);
    std::memset(c->sym_sig, 0, sizeof(c->sym_sig));          // (the next numeric pass clears flops / nnzC for itself)
    c->have_pairs = c->have_alns = false;
    HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
    HIPCHK(c, hipMemsetAsync(c->flopsr.p, 0, 4 * ((size_t)nr + 2), c->stream));
    HIPCHK(c, hipMemsetAsync(c->nnzC.p, 0, 4 * ((size_t)nr + 2), c->stream));
    HIPCHK(c, hipMemsetAsync(c->ctl.p, 0, 4 * kCtlWords, c->stream));
    uint32_t i0 = 0;
    const uint32_t nown = owned_columns(c, &i0);
    if (!colptrC && !npairs) {
        // estimateFLOP alone (overlap.hpp:157-202): the products per column were summed when the operands were laid out (rowF), no gather at all
        if (nown && c->have_rowlists) {
            k_row_flops_rl<<<nblk(nown), kBlock, 0, c->stream>>>(ptr<uint64_t>(c->Arow), i0, c->part_stride, nown, ptr<uint32_t>(c->flopsr), ptr<uint32_t>(c->nnzC), ptr<uint32_t>(c->ctl));
            KCHK(c);
        } else if (nown) {
            k_row_flops<<<nblk(nown), kBlock, 0, c->stream>>>(ptr<uint32_t>(c->rowF), i0, c->part_stride, nown, ptr<uint32_t>(c->flopsr),
                                                              ptr<uint32_t>(c->nnzC), ptr<uint32_t>(c->ctl));
            KCHK(c);
        }
        hipcub::TransformInputIterator<uint64_t, CastU64, const uint32_t*> it(ptr<uint32_t>(c->flopsr), CastU64());
        uint64_t* d_sum = ptr<uint64_t>(c->colptrC);
        size_t tb = 0;
        HIPCHK(c, hipcub::DeviceReduce::Sum(nullptr, tb, it, d_sum, (int)nr + 1, c->stream));
        ENSURE(c, c->cubtmp, tb + 256);
        tb = c->cubtmp.cap;
        HIPCHK(c, hipcub::DeviceReduce::Sum(c->cubtmp.p, tb, it, d_sum, (int)nr + 1, c->stream));
        uint64_t F = 0;
        HIPCHK(c, hipMemcpyAsync(&F, d_sum, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (flops) *flops = F;
        return 0;
    }
    if (nown) {
        CountArgs a;
        a.Bptr = layout_bptr(c); a.Bent = ptr<uint2>(c->Bent); a.Aent = ptr<uint2>(c->Aent); a.inl = c->layout_inline;
        a.Aent2 = c->have_rowlists ? ptr<uint2>(c->Aent2) : nullptr;
        a.Arow = c->have_rowlists ? ptr<uint64_t>(c->Arow) : nullptr;
        a.i0 = i0; a.stride = c->part_stride; a.nown = nown;
        a.words = (nr + 31) / 32;
        a.nnzC = ptr<uint32_t>(c->nnzC); a.flops = ptr<uint32_t>(c->flopsr);
        a.totals = (unsigned long long*)(ptr<uint32_t>(c->ctl) + kCtlTotals);
        const size_t map_bytes = 4 * (size_t)a.words;
        const bool in_lds = map_bytes <= 150 * 1024 && !(c->debug & 8192u);   // debug bit 13: tests, bitmaps in global memory
        if (in_lds) {
            // as many workgroups per CU as the bitmaps allow (at most 4 x 512 threads)
            const uint32_t per_cu = (uint32_t)std::min<size_t>(4, (160 * 1024) / (map_bytes + 64));
            const uint32_t grid = std::min<uint32_t>(nown, 256u * (per_cu ? per_cu : 1u));
            HIPCHK(c, hipFuncSetAttribute((const void*)k_count_pairs<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)map_bytes));
            a.gmap = nullptr;
            k_count_pairs<true><<<grid, kCountBlock, map_bytes, c->stream>>>(a);
        } else {
            const uint32_t grid = std::min<uint32_t>(nown, 1024u);
            ENSURE(c, c->ws, map_bytes * grid);
            a.gmap = ptr<uint32_t>(c->ws);
            k_count_pairs<false><<<grid, kCountBlock, 0, c->stream>>>(a);
        }
        KCHK(c);
    }
    {
        hipcub::TransformInputIterator<uint64_t, CountU64, const uint32_t*> it(ptr<uint32_t>(c->nnzC), CountU64());
        size_t tb = 0;
        HIPCHK(c, hipcub::DeviceScan::ExclusiveSum(nullptr, tb, it, ptr<uint64_t>(c->colptrC), (int)nr + 1, c->stream));
        ENSURE(c, c->cubtmp, tb + 256);
        tb = c->cubtmp.cap;
        HIPCHK(c, hipcub::DeviceScan::ExclusiveSum(c->cubtmp.p, tb, it, ptr<uint64_t>(c->colptrC), (int)nr + 1, c->stream));
    }
    HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
    uint64_t tot[2] = {0, 0};
    HIPCHK(c, hipMemcpyAsync(tot, ptr<uint32_t>(c->ctl) + kCtlTotals, 16, hipMemcpyDeviceToHost, c->stream));
    if (colptrC) HIPCHK(c, hipMemcpyAsync(colptrC, c->colptrC.p, 8 * ((size_t)nr + 1), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->tm.symbolic_ms = ev_ms(c->ev[2], c->ev[3]);
    c->tm.symbolic_passes++;
    if (npairs) *npairs = tot[0];
    if (flops) *flops = tot[1];
    return 0;
}

static int get_memory_impl(bella_ctx* c, bella_memory* m);
// (the struct grew from 64 to 72 bytes between ABI versions 4 and 5 with no size handshake.)  The sized form writes at most
// struct_size bytes: a caller built against an older, shorter bella_memory gets the fields it knows and nothing behind its struct.
int bella_hip_get_memory_sized(bella_ctx* c, void* out, uint64_t struct_size) {
    if (!c || !out || struct_size < 8) return BELLA_ERR_BAD_ARG;
    bella_memory m{};
    const int rc = get_memory_impl(c, &m);
    if (rc) return rc;
    std::memcpy(out, &m, (size_t)std::min<uint64_t>(struct_size, sizeof(m)));
    return 0;
}
int bella_hip_get_memory(bella_ctx* c, bella_memory* m) { return get_memory_impl(c, m); }
static int get_memory_impl(bella_ctx* c, bella_memory* m) {
    if (!c || !m) return BELLA_ERR_BAD_ARG;
    auto sum = [](std::initializer_list<const Buf*> l) { uint64_t t = 0; for (const Buf* b : l) t += b->cap; return t; };
    m->reads_bytes = sum({&c->packed, &c->roff});
    m->matrix_bytes = sum({&c->Bptr, &c->Bk, &c->Bpos});
    m->layout_A_bytes = sum({&c->Aent});
    m->layout_B_bytes = sum({&c->Bent, &c->rowF, &c->Bloc});
    m->rowlist_bytes = sum({&c->Aent2, &c->Aov, &c->Arow});
    m->pass_bytes = sum({&c->flopsr, &c->flopptr, &c->nnzC, &c->colptrC, &c->rowlists, &c->tiercaps, &c->tmp_pairs, &c->tmp_ext, &c->pairs, &c->ext, &c->sortscr, &c->ws,
                         &c->cubtmp, &c->plist_hv, &c->overflow, &c->ctl, &c->retry, &c->orderlist, &c->order_ws, &c->w_f, &c->w_off, &c->w_key, &c->w_key2, &c->w_idx,
                         &c->w_idx2, &c->w_hv, &c->w_ovfl, &c->w_plist, &c->w_scr, &c->w_rlen, &c->w_rstart, &c->w_rrank, &c->w_redo, &c->w_segfirst, &c->w_toff, &c->w_table,
                         &c->w_nruns, &c->w_desc, &c->w_gtab, &c->w_gcount, &c->w_gbase, &c->w_rfirst, &c->w_aent2, &c->w_aov});
    m->other_bytes = sum({&c->t_kmer, &c->t_read, &c->t_pos, &c->tstart, &c->Bk_tmp, &c->Bpos_tmp, &c->rowcnt, &c->asm_ws, &c->asm_cls, &c->lk_key, &c->lk_key2, &c->lk_val,
                          &c->lk_val2, &c->lk_rinfo, &c->w, &c->wscan, &c->kc_nk, &c->kc_koff, &c->kc_hist, &c->kc_keys, &c->kc_alt, &c->kc_runlen, &c->kc_flag, &c->kc_slot, &c->kc_nruns,
                          &c->kc_dcode, &c->kc_dcount, &c->kc_hkey, &c->kc_hval, &c->kc_found, &c->kc_tstart, &c->kc_cursor, &c->kc_sel, &c->kc_ringo, &c->kc_ringp, &c->kc_opos, &c->kc_oid, &c->kc_opos2, &c->kc_oid2, &c->alns,
                          &c->seeds, &c->xest, &c->xest2, &c->xids, &c->xorder, &c->xres, &c->xstate, &c->xlive, &c->lg_res, &c->lg_redo, &c->lg_scratch, &c->status, &c->comm_meta});
    m->other_bytes += c->pool.bytes;                               // (released buffers waiting for the next request they fit)
    m->owned_nnz = c->have_matrix ? c->owned_nnz : 0;
    m->layout_shared = c->have_matrix && c->layout_dist ? 1 : 0;
    m->live_nnz = c->have_matrix ? c->live_nnz : 0;
    return 0;
}

int bella_hip_get_pairs(bella_ctx* c, bella_pair* pairs, bella_pair_ext* ext, uint64_t* colptrC) {
    if (!c) return BELLA_ERR_BAD_ARG;
    if (!c->have_pairs) return fail(c, BELLA_ERR_STATE, "overlap first");
    HIPCHK(c, hipSetDevice(c->device));
    if (pairs && c->npairs)
        HIPCHK(c, c->stager.d2h(pairs, c->pairs.p, sizeof(bella_pair) * c->npairs, c->stream));
    if (ext && c->npairs) {
        if (c->debug & 2u) return fail(c, BELLA_ERR_STATE, "pair_ext output disabled by debug flag");
        HIPCHK(c, c->stager.d2h(ext, c->ext.p, sizeof(bella_pair_ext) * c->npairs, c->stream));
    }
    if (colptrC)
        HIPCHK(c, hipMemcpyAsync(colptrC, c->colptrC.p, 8 * ((size_t)c->nreads + 1), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// the exact gapped X-drop of the reference's CUDA build (logan.hpp): one wavefront per extension, LDS rings; the rare extension
// whose band outgrows them is redone on rings in HBM
static int run_logan(bella_ctx* c, const bella_params* p, const bella_seed* d_seeds, const bella_pair* d_pairs, uint64_t n, bella_aln* d_out) {
    if (2 * n >= 0xFFFFFFF0ull) return fail(c, BELLA_ERR_BAD_ARG, "too many pairs for one batch (2^31; the redo list indexes extensions with 32 bits)");
    LoganArgs a;
    a.seeds = d_seeds; a.pairs = d_pairs; a.n = n;
    a.packed = ptr<uint32_t>(c->packed); a.roff = ptr<uint64_t>(c->roff);
    a.k = p->kmer_size; a.xdrop = p->xdrop;
    const double p_mat = pow(1 - p->error_rate, 2), p_mis = 1 - p_mat;
    a.ratiophi = 1.0 * p_mat - 1.0 * p_mis;
    a.delta = p->delta_chernoff;
    a.out = d_out;
    ENSURE(c, c->lg_res, 16 * (2 * n + 1));
    ENSURE(c, c->lg_redo, 4 * (2 * n + 2));
    a.res = ptr<int4>(c->lg_res);
    a.redo = ptr<uint32_t>(c->lg_redo) + 1;
    a.nredo = ptr<uint32_t>(c->lg_redo);
    a.scratch = nullptr; a.scratch_cap = 0;
    HIPCHK(c, hipEventRecord(c->ev[8], c->stream));
    if (n) {
        HIPCHK(c, hipMemsetAsync(a.nredo, 0, 4, c->stream));
        // one 64-thread workgroup per extension; grid x block must stay below 2^32 threads: chunks of 2^24 extensions
        HIPCHK(c, hipMemsetAsync(ptr<uint32_t>(c->status) + 14, 0, 4, c->stream));
        a.status = ptr<uint32_t>(c->status) + 14;
        const uint64_t kChunk = (c->debug & 256u) ? 1000ull : (1ull << 24);   // debug bit 8: tests, tiny chunks
        for (uint64_t eb = 0; eb < 2 * n; eb += kChunk) {
            a.ebase = eb;
            const uint64_t cnt = 2 * n - eb < kChunk ? 2 * n - eb : kChunk;
            k_logan_lds<<<(unsigned)cnt, 64, 0, c->stream>>>(a);
            KCHK(c);
        }
        uint32_t nredo = 0;
        HIPCHK(c, hipMemcpyAsync(&nredo, a.nredo, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (nredo) {
            uint32_t longest = 0;
            for (uint32_t l : c->host_lens) longest = l > longest ? l : longest;
            const uint32_t cap = longest + 8;                     // a band is never wider than the shorter sequence (+ the rings' slack)
            const unsigned grid = nredo < 256u ? nredo : 256u;
            ENSURE(c, c->lg_scratch, (size_t)grid * 3 * cap * 2);
            a.scratch = ptr<int16_t>(c->lg_scratch);
            a.scratch_cap = cap;
            k_logan_hbm<<<grid, 64, 0, c->stream>>>(a);
            KCHK(c);
        }
        k_logan_finish<<<nblk(n), 256, 0, c->stream>>>(a);
        KCHK(c);
        uint32_t lst = 0;
        HIPCHK(c, hipMemcpyAsync(&lst, a.status, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (lst & 512u) return fail(c, BELLA_ERR_STATE, "internal: an extension's band outgrew the rings in HBM");
    }
    HIPCHK(c, hipEventRecord(c->ev[9], c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev[9]));
    c->tm.xdrop_ms = ev_ms(c->ev[8], c->ev[9]);
    return 0;
}

static int run_xdrop(bella_ctx* c, const bella_params* p, const bella_seed* d_seeds, const bella_pair* d_pairs, uint64_t n,
                     bella_aln* d_out) {
    XdropArgs a;
    a.seeds = d_seeds;
    a.pairs = d_pairs;
    a.n = n;
    a.packed = ptr<uint32_t>(c->packed);
    a.roff = ptr<uint64_t>(c->roff);
    a.k = p->kmer_size;
    a.xdrop = p->xdrop;
    a.out = d_out;
    // ratiophi = slope(e) (align.hpp:72-80), computed in double on the host exactly as main.cpp:323 does
    const double p_mat = pow(1 - p->error_rate, 2), p_mis = 1 - p_mat;
    a.ratiophi = 1.0 * p_mat - 1.0 * p_mis;
    a.delta = p->delta_chernoff;
    HIPCHK(c, hipEventRecord(c->ev[8], c->stream));
    if (n) {
        if (c->xdrop_variant == 3) {
            k_xdrop<<<nblk(n, kXdropPairsPerBlock), kXdropBlock, 0, c->stream>>>(a);
            KCHK(c);
        } else if (c->xdrop_variant == 2) {
            k_xdrop_packed<<<nblk(n, kXdropPairsPerBlock), kXdropBlock, 0, c->stream>>>(a);
            KCHK(c);
        } else {
            // plan -> sort extensions by their step bound (longest first) -> run -> combine
            const uint64_t ne = 2 * n;
            ENSURE(c, c->xest, 4 * ne);
            ENSURE(c, c->xest2, 4 * ne);
            ENSURE(c, c->xids, 4 * ne);
            ENSURE(c, c->xorder, 4 * ne);
            ENSURE(c, c->xres, 16 * ne);
            XdropSortedArgs sa;
            sa.a = a;
            sa.est = ptr<uint32_t>(c->xest);
            sa.ids = ptr<uint32_t>(c->xids);
            sa.order = ptr<uint32_t>(c->xorder);
            sa.res = ptr<int4>(c->xres);
            HIPCHK(c, hipMemsetAsync(c->xres.p, 0xFF, 16 * ne, c->stream));   // (a result nobody wrote is not a result of an earlier call)
            k_xdrop_plan<<<nblk(ne), 256, 0, c->stream>>>(sa);
            KCHK(c);
            size_t tb = 0;
            HIPCHK(c, hipcub::DeviceRadixSort::SortPairsDescending(nullptr, tb, ptr<uint32_t>(c->xest), ptr<uint32_t>(c->xest2),
                                                                  ptr<uint32_t>(c->xids), ptr<uint32_t>(c->xorder), (int)ne, 0, 18,
                                                                  c->stream));
            ENSURE(c, c->cubtmp, tb);
            HIPCHK(c, hipcub::DeviceRadixSort::SortPairsDescending(c->cubtmp.p, tb, ptr<uint32_t>(c->xest), ptr<uint32_t>(c->xest2),
                                                                  ptr<uint32_t>(c->xids), ptr<uint32_t>(c->xorder), (int)ne, 0, 18,
                                                                  c->stream));
            // In one launch a wavefront runs until its longest lane ends; chance pairs (one shared k-mer: 92 % of the pairs of the 100k-read
            // set) end long before the step bound they are scheduled by, and their wavefronts idle behind the few lanes that run on: 5.2 s
            // against 3.1 s in slices on the 100k set, 75.3 against 71.4 ms on the 10k set.
            const bool one_launch = c->xdrop_variant == 0;
            if (one_launch) {                                      // a wavefront runs until its longest lane ends
                k_xdrop_sorted<<<nblk(ne, kXdropBlock), kXdropBlock, 0, c->stream>>>(sa);
                KCHK(c);
            } else {
                // slices (xdrop_packed.hpp): the state of the live extensions lives in HBM between launches of at most kXdropSlice steps;
                // every launch runs full wavefronts of survivors.  Batches bound the state memory (288 B per extension).
                // (at most 16 M extensions = 4.8 GB, at most a quarter of the free device memory, at least 64 k; an allocation that
                // fails all the same halves the batch, and below 64 k extensions the one-launch kernel, which needs no state, takes over)
                uint64_t capB = ne < (16ull << 20) ? ne : (16ull << 20);
                {
                    size_t mfree = 0, mtotal = 0;
                    if (hipMemGetInfo(&mfree, &mtotal) == hipSuccess) {
                        const uint64_t held = c->xstate.cap + c->xlive.cap + c->pool.bytes + c->pool.arena.largest_hole();   // (one buffer: the slab's largest hole, not the sum of its holes)
                        const uint64_t fit = ((uint64_t)mfree + held) / 4 / (4 * (uint64_t)kXStateWords + 8);
                        if (capB > fit) capB = fit;
                    }
                    if (capB < (64ull << 10)) capB = ne < (64ull << 10) ? ne : (64ull << 10);
                }
                bool have_state = false;
                for (;;) {
                    if (!ensure_bytes(c, c->xstate, 4 * (size_t)kXStateWords * capB) && !ensure_bytes(c, c->xlive, 2 * 4 * capB + 64)) { have_state = true; break; }
                    if (capB <= (64ull << 10)) break;
                    capB >>= 1;
                }
                if (!have_state) {
                    c->err.clear();
                    release(c->xstate); release(c->xlive);
                    k_xdrop_sorted<<<nblk(ne, kXdropBlock), kXdropBlock, 0, c->stream>>>(sa);
                    KCHK(c);
                } else {
                // Classes: the extensions are sorted by their step estimate, longest first.  Advanced together -- one slice per ROUND of
                // the whole batch -- the longest ones would finish alone, one wavefront per SIMD, long after everybody else (a quarter of the
                // stage at 10k reads: 41 rounds, the last 25 of them on a fifth of the machine).  So a big batch is cut, in sorted order, into
                // four classes -- 1/16, 2/16, 4/16 of the extensions and the rest -- and the three classes of long extensions run their own
                // slices back to back on side streams, at a higher wave priority (s_setprio) and queue priority, while the bulk fills the
                // machine: every class ends within lenH + lenV steps, so it gets that many launches and the empty ones return at once.
                XdropSliceArgs xa;
                xa.s = sa;
                xa.cap = capB;
                xa.steps = kXdropSlice;                              // (64 / 128 steps for the first two slices: 3.41 s against 3.35 s on the 100k set)
                uint32_t* const L = ptr<uint32_t>(c->xlive);         // per class: two lists of its slots, back to back; then two counters per class
                uint32_t* const counters = L + 2 * capB;
                uint32_t longest = 0;
                for (uint32_t l : c->host_lens) longest = l > longest ? l : longest;
                const int nsl_bound = (int)((2u * longest) / (uint32_t)kXdropSlice) + 3;
                for (uint64_t first = 0; first < ne; first += capB) {
                    const uint64_t count = ne - first < capB ? ne - first : capB;
                    xa.state = ptr<uint32_t>(c->xstate);
                    xa.first = first;
                    xa.count = count;
                    xa.live_in = nullptr; xa.nlive_in = nullptr; xa.live_out = nullptr; xa.nlive_out = nullptr; xa.prio = 0;
                    k_xdrop_begin<<<nblk(count, kXdropBlock), kXdropBlock, 0, c->stream>>>(xa);
                    KCHK(c);
                    // class boundaries (slots of the batch, sorted order): only a batch several times the machine's 4,096 resident wavefronts
                    // is worth cutting
                    uint64_t cb[5] = {0, 0, 0, 0, count};
                    const bool classes = count >= c->xdrop_class_min && count >= 1024 && c->xdrop_variant == 1;
                    if (classes) { cb[1] = (count / 16) & ~63ull; cb[2] = (3 * count / 16) & ~63ull; cb[3] = (7 * count / 16) & ~63ull; }
                    if (classes) HIPCHK(c, hipEventRecord(c->fork, c->stream));
                    int nside = 0;
                    for (int k2 = 0; k2 < 3 && classes; ++k2) {      // the long classes: fixed number of launches, no host round trip
                        const uint64_t c0 = cb[k2], cn = cb[k2 + 1] - cb[k2];
                        if (!cn) continue;
                        hipStream_t sst = c->side[k2];
                        HIPCHK(c, hipStreamWaitEvent(sst, c->fork, 0));
                        XdropSliceArgs ya = xa;
                        ya.state = ptr<uint32_t>(c->xstate) + c0;     // (word w of slot t at state[w * cap + t]: the class's slots start at c0)
                        ya.count = cn;
                        ya.prio = 3 - k2;
                        uint32_t* const lists[2] = {L + 2 * c0, L + 2 * c0 + cn};
                        uint32_t* const cnts = counters + 2 * (k2 + 1);
                        ya.live_in = nullptr; ya.nlive_in = nullptr;
                        for (int s2 = 0; s2 < nsl_bound; ++s2) {
                            ya.live_out = lists[s2 & 1]; ya.nlive_out = cnts + (s2 & 1);
                            HIPCHK(c, hipMemsetAsync(ya.nlive_out, 0, 4, sst));
                            k_xdrop_slice<<<nblk(cn, kXdropBlock), kXdropBlock, 0, sst>>>(ya);
                            KCHK(c);
                            ya.live_in = ya.live_out; ya.nlive_in = ya.nlive_out;
                        }
                        HIPCHK(c, hipEventRecord(c->join[k2], sst));
                        nside = k2 + 1;
                    }
                    {   // the bulk (or the whole batch): as many launches as it takes, the survivor count read back every fourth slice
                        const uint64_t c0 = cb[3], cn = cb[4] - cb[3];
                        XdropSliceArgs za = xa;
                        za.state = ptr<uint32_t>(c->xstate) + c0;
                        za.count = cn;
                        uint32_t* const lists[2] = {L + 2 * c0, L + 2 * c0 + cn};
                        uint64_t live = cn;
                        za.live_in = nullptr; za.nlive_in = nullptr;
                        for (int it = 0; live; ++it) {
                            const int o = it & 1;
                            za.live_out = lists[o]; za.nlive_out = counters + o;
                            HIPCHK(c, hipMemsetAsync(za.nlive_out, 0, 4, c->stream));
                            k_xdrop_slice<<<nblk(live, kXdropBlock), kXdropBlock, 0, c->stream>>>(za);   // (live: an upper bound; the kernel reads the count)
                            KCHK(c);
                            if ((it & 3) == 3 || live <= 4096) {          // the survivor count comes back every fourth slice: it only ever shrinks
                                HIPCHK(c, hipMemcpyAsync(c->pinned + 100, za.nlive_out, 4, hipMemcpyDeviceToHost, c->stream));
                                HIPCHK(c, hipStreamSynchronize(c->stream));
                                live = c->pinned[100];
                            }
                            za.live_in = za.live_out; za.nlive_in = za.nlive_out;
                            if (it > 4096) return fail(c, BELLA_ERR_STATE, "internal: X-drop slices do not terminate");
                        }
                    }
                    for (int k2 = 0; k2 < nside; ++k2)              // (the next batch reuses the state slots and the lists; k_xdrop_finish reads every result)
                        HIPCHK(c, hipStreamWaitEvent(c->stream, c->join[k2], 0));
                    if (nside) {
                        // the long classes ran a FIXED number of slices (the bound lenH + lenV of an extension's steps): their final survivor
                        // counts are checked, and a class that the bound did not finish -- it never happened; a change of the slice size or of
                        // Phase 4's length could make it happen -- goes on slicing here, like the bulk, until nobody is left
                        const int fin = (nsl_bound - 1) & 1;
                        HIPCHK(c, hipMemcpyAsync(c->pinned + 101, counters + 2, 4 * 6, hipMemcpyDeviceToHost, c->stream));
                        HIPCHK(c, hipStreamSynchronize(c->stream));
                        for (int k2 = 0; k2 < nside; ++k2) {
                            uint64_t live = c->pinned[101 + 2 * k2 + fin];
                            if (!live) continue;
                            const uint64_t c0 = cb[k2], cn = cb[k2 + 1] - cb[k2];
                            XdropSliceArgs ya = xa;
                            ya.state = ptr<uint32_t>(c->xstate) + c0;
                            ya.count = cn;
                            uint32_t* const lists[2] = {L + 2 * c0, L + 2 * c0 + cn};
                            uint32_t* const cnts = counters + 2 * (k2 + 1);
                            ya.live_in = lists[fin]; ya.nlive_in = cnts + fin;
                            for (int it = 0, o = fin ^ 1; live; ++it, o ^= 1) {
                                ya.live_out = lists[o]; ya.nlive_out = cnts + o;
                                HIPCHK(c, hipMemsetAsync(ya.nlive_out, 0, 4, c->stream));
                                k_xdrop_slice<<<nblk(live, kXdropBlock), kXdropBlock, 0, c->stream>>>(ya);
                                KCHK(c);
                                HIPCHK(c, hipMemcpyAsync(c->pinned + 100, ya.nlive_out, 4, hipMemcpyDeviceToHost, c->stream));
                                HIPCHK(c, hipStreamSynchronize(c->stream));
                                live = c->pinned[100];
                                ya.live_in = ya.live_out; ya.nlive_in = ya.nlive_out;
                                if (it > 4096) return fail(c, BELLA_ERR_STATE, "internal: X-drop slices do not terminate");
                            }
                        }
                    }
                }
                }   // have_state
            }
            k_xdrop_finish<<<nblk(n), 256, 0, c->stream>>>(sa);
            KCHK(c);
        }
    }
    HIPCHK(c, hipEventRecord(c->ev[9], c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev[9]));
    c->tm.xdrop_ms = ev_ms(c->ev[8], c->ev[9]);
    return 0;
}

static int align_pairs_impl(bella_ctx* c, const bella_params* p, uint64_t* npassed, bool exact);
int bella_hip_align_pairs(bella_ctx* c, const bella_params* p, uint64_t* npassed) { return align_pairs_impl(c, p, npassed, false); }
int bella_hip_align_pairs_exact(bella_ctx* c, const bella_params* p, uint64_t* npassed) { return align_pairs_impl(c, p, npassed, true); }

static int align_pairs_impl(bella_ctx* c, const bella_params* p, uint64_t* npassed, bool exact) {
    if (!c) return BELLA_ERR_BAD_ARG;
    if (!c->have_pairs) return fail(c, BELLA_ERR_STATE, "overlap first");
    int rc = check_params(c, p);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    ENSURE(c, c->alns, sizeof(bella_aln) * c->npairs);
    rc = exact ? run_logan(c, p, nullptr, ptr<bella_pair>(c->pairs), c->npairs, ptr<bella_aln>(c->alns))
               : run_xdrop(c, p, nullptr, ptr<bella_pair>(c->pairs), c->npairs, ptr<bella_aln>(c->alns));
    if (rc) { (void)hipDeviceSynchronize(); return rc; }      // (an error may leave class slices in flight on the side streams)
    c->nalns = c->npairs;
    c->have_alns = true;
    if (npassed) {
        // small reduction on the host side of the ABI would need a copy; count on device instead
        ENSURE(c, c->cubtmp, 64);
        HIPCHK(c, hipMemsetAsync(c->status.p, 0, 16, c->stream));
        if (c->nalns) {
            k_count_passed<<<nblk(c->nalns), 256, 0, c->stream>>>(ptr<bella_aln>(c->alns), c->nalns,
                                                                  (unsigned long long*)(ptr<uint32_t>(c->status) + 2));
            KCHK(c);
        }
        unsigned long long cnt = 0;
        HIPCHK(c, hipMemcpyAsync(&cnt, ptr<uint32_t>(c->status) + 2, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        *npassed = cnt;
    }
    return 0;
}

int bella_hip_get_alignments(bella_ctx* c, bella_aln* out) {
    if (!c || !out) return BELLA_ERR_BAD_ARG;
    if (!c->have_alns) return fail(c, BELLA_ERR_STATE, "align_pairs first");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->nalns) HIPCHK(c, c->stager.d2h(out, c->alns.p, sizeof(bella_aln) * c->nalns, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

static int xdrop_batch_impl(bella_ctx* c, const bella_seed* seeds, uint64_t n, const bella_params* p, bella_aln* out, bool exact);
int bella_hip_xdrop_batch(bella_ctx* c, const bella_seed* seeds, uint64_t n, const bella_params* p, bella_aln* out) {
    return xdrop_batch_impl(c, seeds, n, p, out, false);
}
int bella_hip_xdrop_batch_exact(bella_ctx* c, const bella_seed* seeds, uint64_t n, const bella_params* p, bella_aln* out) {
    return xdrop_batch_impl(c, seeds, n, p, out, true);
}

static int xdrop_batch_impl(bella_ctx* c, const bella_seed* seeds, uint64_t n, const bella_params* p, bella_aln* out, bool exact) {
    if (!c || !p || (n && (!seeds || !out))) return fail(c, BELLA_ERR_BAD_ARG, "null argument");
    if (!c->have_reads) return fail(c, BELLA_ERR_STATE, "set_reads first");
    if (p->kmer_size < 1 || p->kmer_size > 32) return fail(c, BELLA_ERR_BAD_ARG, "k must be in [1,32]");
    HIPCHK(c, hipSetDevice(c->device));
    // validate on the host: the kernel trusts seeds
    std::vector<uint64_t> off((size_t)c->nreads + 1);
    HIPCHK(c, hipMemcpy(off.data(), c->roff.p, 8 * ((size_t)c->nreads + 1), hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < n; ++i) {
        const bella_seed& s = seeds[i];
        if (s.rid >= c->nreads || s.cid >= c->nreads) return fail(c, BELLA_ERR_BAD_ARG, "seed %llu: read id out of range", (unsigned long long)i);
        const uint64_t lh = off[s.rid + 1] - off[s.rid], lv = off[s.cid + 1] - off[s.cid];
        if ((uint64_t)s.seedH + p->kmer_size > lh || (uint64_t)s.seedV + p->kmer_size > lv)
            return fail(c, BELLA_ERR_BAD_ARG, "seed %llu: k-mer past the end of a read", (unsigned long long)i);
    }
    Buf dalns;
    ENSURE(c, c->seeds, sizeof(bella_seed) * n);
    int rc = ensure_bytes(c, dalns, sizeof(bella_aln) * n);
    if (rc) return rc;
    hipError_t e = hipSuccess;
    if (n) e = hipMemcpyAsync(c->seeds.p, seeds, sizeof(bella_seed) * n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        rc = exact ? run_logan(c, p, ptr<bella_seed>(c->seeds), nullptr, n, ptr<bella_aln>(dalns))
                   : run_xdrop(c, p, ptr<bella_seed>(c->seeds), nullptr, n, ptr<bella_aln>(dalns));
        if (rc == 0 && n) e = hipMemcpyAsync(out, dalns.p, sizeof(bella_aln) * n, hipMemcpyDeviceToHost, c->stream);
        if (rc == 0 && e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    release(dalns);
    if (rc) return rc;
    if (e != hipSuccess) return fail(c, BELLA_ERR_HIP, "xdrop_batch: %s", hipGetErrorString(e));
    return 0;
}

int bella_hip_write_output(const char* path, const bella_params* p, int paf, uint32_t nreads, const char* const* names, const uint32_t* lens,
                           const bella_pair* pairs, const bella_aln* alns, uint64_t npairs, int nthreads, bella_write_stats* stats) {
    std::string err;
    const int rc = write_output_impl(path, p, paf, nreads, names, lens, pairs, alns, npairs, nthreads, stats, err);
    if (rc) fprintf(stderr, "bella_hip_write_output: %s\n", err.c_str());
    return rc;
}

int bella_hip_get_timings(bella_ctx* c, bella_timings* t) {
    if (!c || !t) return BELLA_ERR_BAD_ARG;
    *t = c->tm;
    return 0;
}

}  // extern "C"
