// wide.hpp -- output columns with 65,536 or more products (the row kernels index a column's products with 16 bits).
//
// Same computation as spgemm.hpp (overlap.hpp:281-363 LocalSpGEMM + chain.hpp:74-150), arranged for columns of any size:
// expand the column's products to HBM, radix-sort them by (column, partner read) -- stable, so a pair's products stay in
// product order --, run-length encode into pairs, reproduce the reference's hash-slot order per column with the same
// atomicMin insertion as phase O of the row kernel (64-bit items), fold every pair -- one workgroup per pair with the closed
// form of the row kernel's phase P (parents + independent walks) on the pair's list in HBM; the serial statement of the
// semiring (core.hpp:fold_pair, one lane per pair) only for lists of >= 32768 products and pairs that end with > 16 bins --
// and write the records at the column's ranks.  Also the path of the columns above the LDS tiers (11,009 ... 65,535 products)
// when a pass has more than a handful of them (HiFi sets with a raised -u, repeats, deep coverage).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/bella_hip.h"
#include "core.hpp"
#include "util.hpp"

namespace bella {

struct WidePairAddr;
struct WideArgs {
    const uint32_t* cols;        // [nw] the wide columns
    uint32_t nw;
    const uint64_t* woff;        // [nw+1] start of each column's products in the W arrays
    const uint32_t* Bptr;
    const uint2* Bent;
    uint32_t inl;                // B' entries in the INLINE form (util.hpp)
    const uint2* Aent;
    const uint2* Aent2;          // ready-made products (assemble.hpp: k_layout_rowlists; nullptr: expand from B' x A')
    const uint16_t* Aov;
    const uint64_t* Arow;        // nullptr: the lists are the batch's own (column cols[s] starts at woff[s]; bella_hip.hip: run_wide_batch)
    const uint64_t* roff;
    const uint32_t* packed;
    const uint64_t* flopptr;
    int k;
    int binSize;
    // products, product order
    void* W_key;                 // segment << rbits | partner read (rbits = bits of a read id: the sort runs over rbits + segment bits only);
                                 // u32 words when rbits + segment bits <= 32 (key32), else u64
    uint32_t key32;
    uint32_t rbits;
    uint32_t* W_idx;             // global product index (woff[seg] + p)
    uint2* W_rec;                // {posH | posV << 16, overlap estimate | flags << 16}
    // sorted
    const void* S_key;
    const uint32_t* S_idx;
    uint2* plist;                // [totalF + 64] {hv, ov} in sorted order = per-pair lists in product order
    uint64_t plist_pad;          // totalF: 64 entries behind the lists that k_wide_group2_chunks' idle lanes write
    // pairs (runs of S_key)
    const void* R_key;           // [npairs]
    const uint32_t* R_len;
    const uint32_t* R_start;     // exclusive scan of R_len
    uint32_t npairs;
    uint32_t* seg_first;         // [nw+1] first pair of each segment
    uint64_t* table;             // slot-order tables, all segments back to back
    const uint64_t* toff;        // [nw+1]
    uint32_t* R_rank;            // [npairs] output rank inside the column
    uint16_t* sort_scratch;      // [totalF]
    bella_pair* tmp_pairs;
    bella_pair_ext* tmp_ext;
    uint32_t* nnzC;
    uint32_t* status;            // ctl word: bit0 = > 16 bins without scratch (never here)
#ifdef BELLA_WF_CLOCK
    unsigned long long* clk;
#endif
    WidePairAddr* desc;          // [npairs] (k_wide_desc)
    // grouping in LDS (k_wide_group1 / _group2): per column a table image of kWideGroupSlots entries {key, list start, products, first product},
    // the columns' pair counts and their prefix sums, a flag for columns whose pairs do not fit the table, the pairs' first products
    uint4* gtab;
    uint32_t* gcount;            // [nw + 2]: pairs per column; gcount[nw + 1] = overflow flag
    const uint32_t* gbase;       // [nw + 1] exclusive scan of gcount
    uint32_t* R_first;           // [npairs] first product of the pair inside its column (nullptr: S_idx[R_start[r]] - woff[seg])
    uint32_t* R_len_w;           // writable views of R_len / R_start / R_key for k_wide_group2_chunks
    uint32_t* R_start_w;
    void* R_key_w;
    uint32_t* redo;              // [npairs] pairs left to the serial fold (k_wide_fold) by k_wide_fold_wg; redo[npairs] = their number
};

// products of the wide columns in the reference's order (B' entry order, then the k-mer's read list)
constexpr int kWideExpandBlock = 1024;                     // one workgroup per column: its entries in rounds of 1024
__global__ __launch_bounds__(kWideExpandBlock) void k_wide_expand(WideArgs a) {
    __shared__ uint32_t scr[kWideExpandBlock / 64];
    __shared__ uint32_t s_off[kWideExpandBlock + 1];        // first product of each entry of the round
    __shared__ uint2 s_be[kWideExpandBlock];
    for (uint32_t s = blockIdx.x; s < a.nw; s += gridDim.x) {
        const uint32_t i = a.cols[s];
        const uint32_t b0 = a.Bptr[i], n = a.Bptr[i + 1] - b0;
        const uint32_t lenV = (uint32_t)(a.roff[i + 1] - a.roff[i]);
        const uint64_t o = a.woff[s];
        const uint64_t arow = a.Aent2 ? (a.Arow ? a.Arow[i] : o) : 0ull;
        if (a.Aent2) {                                          // ready-made products: a stream copy into the sort's input
            const uint64_t Fc = a.woff[s + 1] - o;
            for (uint64_t q = threadIdx.x; q < Fc; q += kWideExpandBlock) {
                const uint2 r2 = a.Aent2[arow + q];
                const uint32_t key = r2.x & 0x3FFFFFFFu, fl = (r2.x >> 31) | (((r2.x >> 30) & 1u) << 1);
                const uint64_t p = o + q;
                if (a.key32) ((uint32_t*)a.W_key)[p] = (s << a.rbits) | key;
                else ((uint64_t*)a.W_key)[p] = ((uint64_t)s << a.rbits) | key;
                a.W_idx[p] = (uint32_t)p;
                const uint32_t ov = (uint32_t)overlap_estimate(r2.y & 0xFFFFu, r2.y >> 16, a.Aov[arow + q], lenV, (r2.x >> 31) != 0, (uint32_t)a.k) & 0xFFFFu;
                a.W_rec[p] = make_uint2(r2.y, ov | (fl << 16));
            }
            continue;
        }
        uint32_t running = 0;
        for (uint32_t jb = 0; jb < n; jb += kWideExpandBlock) {
            const uint32_t j = jb + threadIdx.x;
            uint2 be = make_uint2(0u, 0u);
            if (j < n) be = a.Bent[b0 + j];
            const uint32_t cnt = bent_count(be, a.inl);
            uint32_t tot;
            const uint32_t ex = block_excl_scan<kWideExpandBlock / 64>(cnt, scr, &tot);
            s_off[threadIdx.x] = ex; s_be[threadIdx.x] = be;
            if (threadIdx.x == 0) s_off[kWideExpandBlock] = tot;
            __syncthreads();
            // product-parallel: product q of the round belongs to the last entry whose first product is <= q (binary search over
            // the round's offsets); neighbouring lanes read neighbouring A' entries and write neighbouring products
            for (uint32_t q = threadIdx.x; q < tot; q += kWideExpandBlock) {
                uint32_t lo = 0, hi = kWideExpandBlock;        // s_off[lo] <= q < s_off[hi]
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (s_off[mid] <= q) lo = mid; else hi = mid;
                }
                const uint2 eb = s_be[lo];
                const uint32_t t = q - s_off[lo];
                const BProduct pr = bent_product(eb, t, a.Aent, a.roff, a.inl);
                const uint32_t posV = pr.posV, pal = pr.pal, key = pr.key, posH = pr.posH, lenH = pr.lenH;
                const bool oriented = pr.oriented;
                const uint32_t ov = (uint32_t)overlap_estimate(posH, posV, lenH, lenV, oriented, (uint32_t)a.k) & 0xFFFFu;
                const uint32_t fl = (oriented ? 1u : 0u) | (pal << 1);
                const uint32_t hvw = posH | (posV << 16);
                const uint64_t p = o + running + q;
                if (a.key32) ((uint32_t*)a.W_key)[p] = (s << a.rbits) | key;
                else ((uint64_t*)a.W_key)[p] = ((uint64_t)s << a.rbits) | key;
                a.W_idx[p] = (uint32_t)p;
                a.W_rec[p] = make_uint2(hvw, ov | (fl << 16));
            }
            __syncthreads();
            running += tot;
        }
    }
}

// ---- grouping by partner in LDS (row lists only) ---------------------------------------------------------------------------------
// A wide column has many products but few partners (HiFi reads with a raised -u: 36,000 ... 200,000 products on ~50 ... 500 partners):
// instead of sorting 16 bytes per product three or four times, the column's products are streamed twice.  k_wide_group1: one
// workgroup per column inserts the partner of every product into an LDS table (compare-and-swap), counts the pair's products and
// notes its first product; a scan over the slots gives every pair its list inside the column's part of plist.
// k_wide_group2_chunks: the products again, appended to their pairs' lists in product order (below).  A column whose partners do not
// fit the table raises a flag and the batch takes the sort-based path.
constexpr uint32_t kWideGroupSlots = 2048;                 // table slots per column (at most kWideGroupPairs pairs)
constexpr uint32_t kWideGroupPairs = 1536;
constexpr int kWideGroup1Block = 1024, kWideGroup2Block = 256;
constexpr uint32_t kG1Per = 8;                             // products per thread and round: their loads, then their insertions, together
__global__ __launch_bounds__(kWideGroup1Block) void k_wide_group1(WideArgs a) {
    __shared__ uint32_t s_key[kWideGroupSlots + 1];        // (the last slot: products beyond the column's end)
    __shared__ uint32_t s_cnt[kWideGroupSlots + 1];
    __shared__ uint32_t s_first[kWideGroupSlots + 1];
    __shared__ uint32_t scr[kWideGroup1Block / 64];
    __shared__ uint32_t s_n, s_fail;
    const uint32_t tid = threadIdx.x;
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    for (uint32_t s = blockIdx.x; s < a.nw; s += gridDim.x) {
        const uint32_t i = a.cols[s];
        const uint64_t F = a.woff[s + 1] - a.woff[s], arow = a.Arow ? a.Arow[i] : a.woff[s];
        for (uint32_t h = tid; h <= kWideGroupSlots; h += kWideGroup1Block) { s_key[h] = kNone; s_first[h] = kNone; s_cnt[h] = 0; }
        if (tid == 0) { s_n = 0; s_fail = 0; }
        __syncthreads();
        uint32_t mine = 0;
        bool failed = false;
        const uint64_t plast = F ? F - 1 : 0;
        for (uint64_t p0 = 0; p0 < F && !failed; p0 += (uint64_t)kWideGroup1Block * kG1Per) {
            uint32_t key[kG1Per], h[kG1Per], old[kG1Per];
#pragma unroll
            for (uint32_t u = 0; u < kG1Per; ++u) {
                const uint64_t p = p0 + (uint64_t)u * kWideGroup1Block + tid;
                key[u] = a.Aent2[arow + (p < plast ? p : plast)].x & 0x3FFFFFFFu;
            }
#pragma unroll
            for (uint32_t u = 0; u < kG1Per; ++u) {
                const bool live = p0 + (uint64_t)u * kWideGroup1Block + tid < F;
                h[u] = live ? hash_range(key[u], kWideGroupSlots) : kWideGroupSlots;
                if (!live) key[u] = kNone;
            }
            // (a plain read first: the lanes that meet the same pair share one read, and after a column's first rounds every pair is in
            // the table -- a compare-and-swap only into a slot seen empty, a minimum only below the first product seen so far: what is left
            // per product is the count's atomic)
#pragma unroll
            for (uint32_t u = 0; u < kG1Per; ++u) old[u] = s_key[h[u]];
#pragma unroll
            for (uint32_t u = 0; u < kG1Per; ++u) {
                uint32_t probes = 0;
                for (;;) {
                    if (old[u] == key[u]) break;
                    if (old[u] == kNone) {
                        const uint32_t o = atomicCAS(&s_key[h[u]], kNone, key[u]);
                        if (o == kNone) { ++mine; break; }
                        if (o == key[u]) break;
                    }
                    if (++probes == kWideGroupSlots) { failed = true; break; }
                    h[u] = h[u] + 1 == kWideGroupSlots ? 0 : h[u] + 1;
                    old[u] = s_key[h[u]];
                }
            }
            if (failed) { s_fail = 1; break; }
#pragma unroll
            for (uint32_t u = 0; u < kG1Per; ++u) old[u] = s_first[h[u]];
#pragma unroll
            for (uint32_t u = 0; u < kG1Per; ++u) {
                const uint32_t p = (uint32_t)(p0 + (uint64_t)u * kWideGroup1Block + tid);
                atomicAdd(&s_cnt[h[u]], 1u);
                if (p < old[u]) atomicMin(&s_first[h[u]], p);
            }
        }
        {
            const uint32_t mw = wave_incl_scan(mine);
            if (lane_id() == 63 && mw) atomicAdd(&s_n, mw);
        }
        __syncthreads();
        const uint32_t d = s_n;
        if (s_fail || d > kWideGroupPairs) {                        // (the table would run above 75 % load, or is full)
            if (tid == 0) { a.gcount[s] = 0; a.gcount[a.nw + 1] = 1; }
            __syncthreads();
            continue;
        }
        // lists in slot order: exclusive scan of the pairs' product counts
        constexpr uint32_t kPer = kWideGroupSlots / kWideGroup1Block;
        uint32_t m[kPer], sum = 0;
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) { m[u] = s_cnt[tid * kPer + u]; sum += m[u]; }
        uint32_t tot;
        uint32_t ex = block_excl_scan<kWideGroup1Block / 64>(sum, scr, &tot);
        uint4* G = a.gtab + (size_t)s * kWideGroupSlots;
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) {
            const uint32_t h = tid * kPer + u;
            G[h] = make_uint4(s_key[h], ex, m[u], s_first[h]);       // {key (0xFFFFFFFF: empty), list start inside the column, products, first product}
            ex += m[u];
        }
        if (tid == 0) a.gcount[s] = d;
        __syncthreads();
    }
}

// the pairs of the grouped columns, numbered column by column in slot order (any order does: the output order comes from
// k_wide_insert / k_wide_ranks): key, list, first product -- all that the slot-order tables, ranks and descriptors need, so that those
// run next to the append pass instead of behind it
__global__ __launch_bounds__(kWideGroup2Block) void k_wide_pairs(WideArgs a) {
    __shared__ uint32_t scr[kWideGroup2Block / 64];
    const uint32_t tid = threadIdx.x;
    for (uint32_t s = blockIdx.x; s < a.nw; s += gridDim.x) {
        const uint64_t wo = a.woff[s];
        const uint4* G = a.gtab + (size_t)s * kWideGroupSlots;
        constexpr uint32_t kPer = kWideGroupSlots / kWideGroup2Block;
        uint4 g[kPer];
        uint32_t occ = 0;
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) { g[u] = G[tid * kPer + u]; occ += g[u].x != 0xFFFFFFFFu ? 1u : 0u; }
        uint32_t d;
        uint32_t r = a.gbase[s] + block_excl_scan<kWideGroup2Block / 64>(occ, scr, &d);
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) {
            if (g[u].x == 0xFFFFFFFFu) continue;
            if (a.key32) ((uint32_t*)a.R_key_w)[r] = (s << a.rbits) | g[u].x;
            else ((uint64_t*)a.R_key_w)[r] = ((uint64_t)s << a.rbits) | g[u].x;
            a.R_len_w[r] = g[u].z;
            a.R_start_w[r] = (uint32_t)wo + g[u].y;
            a.R_first[r] = g[u].w;
            ++r;
        }
        __syncthreads();
    }
}

// The append pass: the column's products in CHUNKS of 2,048, every chunk counting-sorted by pair in LDS before it leaves, so that a
// pair's products of the chunk (a HiFi-like column has ~100 partners: ~20 products per pair and chunk) are written as ONE run of
// consecutive list entries by neighbouring lanes.  (Until round 6 four wavefronts per column appended a quarter of the column each,
// every product an 8-byte store into a line of its own: 291 M scattered stores at 10k HiFi-like reads, 2.1 .. 2.5 ms against 1.1 ms
// for the same kernel storing sequentially; this form 1.4 ms.)  Per chunk:
//   A  wavefront q takes the chunk's q-th quarter, 64 products per step in product order: slot of the product's pair, rank among
//      the pair's products of (chunk, wavefront) from an LDS atomic on the pair's packed counters (same-address atomics of one
//      instruction are applied in lane order, successive instructions of a wavefront in program order: rank order = product order);
//   B  one lane per pair of the column: the chunk's products of the pair per wavefront -> where the pair's run starts in the staged
//      chunk (an LDS allocator: the order of the runs is irrelevant), the counts summed from each wavefront on, the list cursor moves on;
//   C  every product to its place in the stage together with its destination (list cursor - products from its wavefront on + rank);
//   D  the stage leaves in stage order: neighbouring lanes, neighbouring list entries.
constexpr uint32_t kG2Chunk = 2048, kG2Per = kG2Chunk / kWideGroup2Block, kG2Quarter = kG2Chunk / (kWideGroup2Block / 64);
static_assert(kWideGroup2Block == 256 && kG2Quarter == 64 * kG2Per, "four wavefronts, a quarter of the chunk each");
__global__ __launch_bounds__(kWideGroup2Block) void k_wide_group2_chunks(WideArgs a) {
    __shared__ uint32_t s_key[kWideGroupSlots];
    __shared__ uint32_t s_cur[kWideGroupSlots + 1];              // the pair's next list entry (inside the column)
    __shared__ uint2 s_cnt[kWideGroupSlots + 1];                 // products of the chunk per wavefront (4 x u16); [kWideGroupSlots]: beyond the column's end
    __shared__ uint2 s_suf[kWideGroupSlots + 1];                 // ... summed from each wavefront on (the first = all of them)
    __shared__ uint16_t s_loc[kWideGroupSlots + 2];              // start of the pair's run in the stage
    __shared__ uint16_t s_occ[kWideGroupPairs];                  // the occupied slots
    __shared__ uint2 s_pay[kG2Chunk];
    __shared__ uint32_t s_dst[kG2Chunk];
    __shared__ uint32_t scr[kWideGroup2Block / 64];
    __shared__ uint32_t s_alloc;
    const uint32_t tid = threadIdx.x, q = wave_id(), lane = lane_id();
    if (a.gcount[a.nw + 1]) return;                                // a column's partners did not fit the table: the batch takes the sort-based path
    for (uint32_t s = blockIdx.x; s < a.nw; s += gridDim.x) {
        const uint32_t i = a.cols[s];
        const uint64_t wo = a.woff[s], F = a.woff[s + 1] - wo, arow = a.Arow ? a.Arow[i] : wo;
        const uint32_t lenV = (uint32_t)(a.roff[i + 1] - a.roff[i]);
        const uint4* G = a.gtab + (size_t)s * kWideGroupSlots;
        constexpr uint32_t kPer = kWideGroupSlots / kWideGroup2Block;
        uint4 g[kPer];
        uint32_t occ = 0;
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) { g[u] = G[tid * kPer + u]; occ += g[u].x != 0xFFFFFFFFu ? 1u : 0u; }
        uint32_t d;
        uint32_t o = block_excl_scan<kWideGroup2Block / 64>(occ, scr, &d);
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) {
            const uint32_t h = tid * kPer + u;
            s_key[h] = g[u].x; s_cur[h] = g[u].y; s_cnt[h] = make_uint2(0u, 0u);
            if (g[u].x != 0xFFFFFFFFu) { s_occ[o] = (uint16_t)h; ++o; }
        }
        if (tid == 0) { s_alloc = 0; s_cnt[kWideGroupSlots] = make_uint2(0u, 0u); s_suf[kWideGroupSlots] = make_uint2(0u, 0u); s_loc[kWideGroupSlots] = 0; s_cur[kWideGroupSlots] = 0; }
        __syncthreads();
        if (F) {
            const uint64_t plast = F - 1;
            // (the next chunk's products are on their way while a chunk is sorted; loads and stores unconditional, a memory operation inside a branch makes the compiler wait for everything in flight)
            uint2 nx[kG2Per];
            uint32_t nl[kG2Per];
#pragma unroll
            for (uint32_t u = 0; u < kG2Per; ++u) {
                const uint64_t p = (uint64_t)q * kG2Quarter + 64ull * u + lane, pc = p < plast ? p : plast;
                nx[u] = a.Aent2[arow + pc]; nl[u] = a.Aov[arow + pc];
            }
            for (uint64_t cb = 0; cb < F; cb += kG2Chunk) {
                // (every phase in batches of the thread's kG2Per products: all LDS reads of a batch, then all atomics, ... -- the
                // round trips of a batch overlap instead of following one another; a product beyond the column's end counts into a
                // slot of its own behind the table and is never staged)
                uint2 pay[kG2Per];
                uint32_t key[kG2Per], hs[kG2Per], seen[kG2Per], old[kG2Per];
                bool live[kG2Per];
#pragma unroll
                for (uint32_t u = 0; u < kG2Per; ++u) {
                    const uint64_t p = cb + (uint64_t)q * kG2Quarter + 64ull * u + lane;
                    const uint2 r2 = nx[u];
                    const uint32_t lenH = nl[u];
                    const uint64_t pn = p + kG2Chunk, pc = pn < plast ? pn : plast;
                    nx[u] = a.Aent2[arow + pc]; nl[u] = a.Aov[arow + pc];
                    const bool oriented = (r2.x >> 31) != 0;
                    const uint32_t ov = (uint32_t)overlap_estimate(r2.y & 0xFFFFu, r2.y >> 16, lenH, lenV, oriented, (uint32_t)a.k) & 0xFFFFu;
                    const uint32_t fl = (oriented ? 1u : 0u) | (((r2.x >> 30) & 1u) << 1);
                    pay[u] = make_uint2(r2.y, ov | (fl << 16));
                    live[u] = p < F;
                    key[u] = r2.x & 0x3FFFFFFFu;
                    hs[u] = hash_range(key[u], kWideGroupSlots);
                }
#pragma unroll
                for (uint32_t u = 0; u < kG2Per; ++u) seen[u] = s_key[hs[u]];
#pragma unroll
                for (uint32_t u = 0; u < kG2Per; ++u) {
                    while (live[u] && seen[u] != key[u]) { hs[u] = hs[u] + 1 == kWideGroupSlots ? 0 : hs[u] + 1; seen[u] = s_key[hs[u]]; }
                    if (!live[u]) hs[u] = kWideGroupSlots;
                }
#pragma unroll
                for (uint32_t u = 0; u < kG2Per; ++u) old[u] = atomicAdd(&((uint32_t*)&s_cnt[hs[u]])[q >> 1], 1u << (16u * (q & 1u)));
                __syncthreads();
                for (uint32_t t0 = q * 64u; t0 < d; t0 += kWideGroup2Block) {
                    const uint32_t t = t0 + lane;
                    uint32_t h = 0, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
                    if (t < d) {
                        h = s_occ[t];
                        const uint2 c = s_cnt[h];
                        c0 = c.x & 0xFFFFu; c1 = c.x >> 16; c2 = c.y & 0xFFFFu; c3 = c.y >> 16;
                    }
                    const uint32_t tot = c0 + c1 + c2 + c3;
                    const uint32_t inc = wave_incl_scan(tot);
                    uint32_t base = 0;
                    if (lane == 63 && inc) base = atomicAdd(&s_alloc, inc);
                    base = (uint32_t)__shfl((int)base, 63, 64);
                    if (tot) {
                        s_cnt[h] = make_uint2(0u, 0u);
                        s_loc[h] = (uint16_t)(base + inc - tot);
                        s_suf[h] = make_uint2(tot | ((c1 + c2 + c3) << 16), (c2 + c3) | (c3 << 16));
                        s_cur[h] += tot;
                    }
                }
                __syncthreads();
                if (tid == 0) s_alloc = 0;
                {
                    uint2 sf[kG2Per];
                    uint32_t lc[kG2Per], cu[kG2Per];
#pragma unroll
                    for (uint32_t u = 0; u < kG2Per; ++u) { sf[u] = s_suf[hs[u]]; lc[u] = s_loc[hs[u]]; cu[u] = s_cur[hs[u]]; }
#pragma unroll
                    for (uint32_t u = 0; u < kG2Per; ++u) {
                        const uint32_t rank = (old[u] >> (16u * (q & 1u))) & 0xFFFFu;
                        const uint32_t from = q == 0 ? sf[u].x & 0xFFFFu : q == 1 ? sf[u].x >> 16 : q == 2 ? sf[u].y & 0xFFFFu : sf[u].y >> 16;
                        const uint32_t pos = lc[u] + (sf[u].x & 0xFFFFu) - from + rank;
                        if (live[u]) { s_pay[pos] = pay[u]; s_dst[pos] = cu[u] - from + rank; }
                    }
                }
                __syncthreads();
                const uint32_t n = F - cb < kG2Chunk ? (uint32_t)(F - cb) : kG2Chunk;
#pragma unroll
                for (uint32_t u = 0; u < kG2Per; ++u) {
                    const uint32_t j = tid + u * kWideGroup2Block;
                    const uint64_t dst = j < n ? wo + s_dst[j] : a.plist_pad + lane;
                    a.plist[dst] = s_pay[j];
                }
            }
        }
        __syncthreads();
    }
}

__device__ __forceinline__ uint64_t wide_rkey(const WideArgs& a, uint32_t r) {
    return a.key32 ? (uint64_t)((const uint32_t*)a.R_key)[r] : ((const uint64_t*)a.R_key)[r];
}

// per-pair lists in product order + first pair of each segment
__global__ void k_wide_gather(WideArgs a, uint64_t totalF) {
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= totalF) return;
    const uint32_t src = a.S_idx[x];
    a.plist[x] = a.W_rec[src];                                       // .y keeps the flags in its upper half until the fold
}
__global__ void k_wide_segments(WideArgs a) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > a.npairs) return;
    const uint32_t seg = r < a.npairs ? (uint32_t)(wide_rkey(a, r) >> a.rbits) : a.nw;
    const uint32_t prev = r == 0 ? 0xFFFFFFFFu : (uint32_t)(wide_rkey(a, r - 1) >> a.rbits);
    if (r == 0) { for (uint32_t s = 0; s <= seg && s <= a.nw; ++s) a.seg_first[s] = 0; }
    else if (seg != prev) { for (uint32_t s = prev + 1; s <= seg && s <= a.nw; ++s) a.seg_first[s] = r; }
}

// the reference's slot order (overlap.hpp:289-361): keys enter a table of pow2 >= max(16, pairs) slots at (key*107) & mask in
// order of their first product; parallel form: atomicMin on (first product << 32 | pair), the displaced entry probes on
// sizes and starts of the columns' tables on the device (one workgroup; the host only knows the bound 16 nw + 2 npairs)
__global__ __launch_bounds__(1024) void k_wide_toff(const uint32_t* seg_first, uint32_t nw, uint64_t* toff) {
    __shared__ uint32_t scr[16];
    __shared__ uint64_t s_run;
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    for (uint32_t b = 0; b <= nw; b += 1024) {
        const uint32_t s = b + threadIdx.x;
        uint32_t ht = 0;
        if (s < nw) {
            const uint32_t d = seg_first[s + 1] - seg_first[s];
            ht = 16;
            while (ht < d) ht <<= 1;                                  // overlap.hpp:291-295
        }
        uint32_t tot;
        const uint32_t ex = block_excl_scan<16>(ht, scr, &tot);       // (a round's sum < 16 * 1024 + 2 * pairs < 2^32)
        const uint64_t run = s_run;
        if (s <= nw) toff[s] = run + ex;
        __syncthreads();
        if (threadIdx.x == 0) s_run = run + tot;
        __syncthreads();
    }
}
__global__ void k_wide_table_fill(uint64_t* table, uint64_t n) {
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x < n) table[x] = ~0ull;
}
__global__ void k_wide_insert(WideArgs a) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.npairs) return;
    const uint32_t seg = (uint32_t)(wide_rkey(a, r) >> a.rbits);
    const uint32_t key = (uint32_t)(wide_rkey(a, r) & ((1ull << a.rbits) - 1ull));
    const uint64_t ht = a.toff[seg + 1] - a.toff[seg];
    unsigned long long* T = (unsigned long long*)(a.table + a.toff[seg]);
    const uint32_t first = a.R_first ? a.R_first[r] : a.S_idx[a.R_start[r]] - (uint32_t)a.woff[seg];    // product index inside the column
    unsigned long long item = ((unsigned long long)first << 32) | r;
    uint64_t h = (uint64_t)(key * 107u) & (ht - 1);
    for (;;) {
        const unsigned long long old = atomicMin(&T[h], item);
        if (old == ~0ull) break;
        if (old > item) {                                       // we took the slot: the displaced pair resumes with ITS key
            item = old;
        }
        h = (h + 1) & (ht - 1);
    }
}
// rank = number of occupied slots before the pair's slot.  One workgroup per segment.
__global__ __launch_bounds__(kBlock) void k_wide_ranks(WideArgs a) {
    __shared__ uint32_t scr[kWaves];
    for (uint32_t s = blockIdx.x; s < a.nw; s += gridDim.x) {
        const uint64_t t0 = a.toff[s], ht = a.toff[s + 1] - t0;
        uint32_t running = 0;
        for (uint64_t base = 0; base < ht; base += kBlock) {
            const uint64_t x = base + threadIdx.x;
            const uint64_t it = x < ht ? a.table[t0 + x] : ~0ull;
            const uint32_t occ = it != ~0ull ? 1u : 0u;
            uint32_t tot;
            const uint32_t ex = block_excl_scan<kWaves>(occ, scr, &tot);
            if (occ) a.R_rank[(uint32_t)it] = running + ex;
            running += tot;
        }
        if (threadIdx.x == 0) a.nnzC[a.cols[s]] = running | kOrderedBit;   // (records at their ranks, cid = the column: k_order_* only moves them)
    }
}

// a pair's list (start, length), where its record goes and where its two reads start: everything the fold of the pair needs
// besides the list itself.  k_wide_desc fills one per pair (48 bytes), so that the fold reads a single record per pair, two pairs
// ahead, instead of chasing key -> column -> read offsets when the pair is done.
struct __attribute__((aligned(16))) WidePairAddr { uint64_t lo; uint32_t mm, key; uint32_t cid, pad; uint64_t roffH, roffV, out; };
static_assert(sizeof(WidePairAddr) == 48, "three 16-byte words");
__device__ __forceinline__ WidePairAddr wide_pair_addr(const WideArgs& a, uint32_t r) {
    WidePairAddr q;
    const uint32_t seg = (uint32_t)(wide_rkey(a, r) >> a.rbits);
    q.lo = a.R_start[r]; q.mm = a.R_len[r];
    q.key = (uint32_t)(wide_rkey(a, r) & ((1ull << a.rbits) - 1ull));
    q.cid = a.cols[seg]; q.pad = 0;
    q.roffH = a.roff[q.key]; q.roffV = a.roff[q.cid];
    q.out = a.flopptr[q.cid] + a.R_rank[r];
    return q;
}
// A descriptor travels as one dword per lane (lane i holds word i of the 12): a per-lane load is not waited for where it is issued
// (a wave-uniform one is, the compiler moves it to scalar registers at once), only where desc_decode reads the lanes an iteration later.
__device__ __forceinline__ uint32_t desc_load_raw(const WidePairAddr* d) { return ((const uint32_t*)d)[lane_id() % 12u]; }
__device__ __forceinline__ WidePairAddr desc_decode(uint32_t raw) {
    uint32_t w[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) w[i] = (uint32_t)__builtin_amdgcn_readlane((int)raw, i);
    WidePairAddr q;
    q.lo = w[0] | ((uint64_t)w[1] << 32); q.mm = w[2]; q.key = w[3]; q.cid = w[4]; q.pad = w[5];
    q.roffH = w[6] | ((uint64_t)w[7] << 32); q.roffV = w[8] | ((uint64_t)w[9] << 32); q.out = w[10] | ((uint64_t)w[11] << 32);
    return q;
}
// The descriptors leave in three classes by list length (k_wide_fold_wg has an instance per class): class c fills
// desc[c * npairs ..) in the order its workgroups arrive (neighbouring pairs stay neighbours),
// the counts stand behind the redo list (redo[npairs + 1 + c]); the pair's own index travels in the descriptor (pad).
constexpr uint32_t kWideFoldSmall = 1024, kWideFoldMid = 2048, kWideFoldTiny = 16;
__device__ __forceinline__ uint32_t wide_fold_class(uint32_t mm) { return mm <= kWideFoldSmall ? 0u : mm <= kWideFoldMid ? 1u : 2u; }
__global__ __launch_bounds__(256) void k_wide_desc(WideArgs a) {
    // class 3: the shortest lists (a third of the pairs of a HiFi-like set have ONE product) are not worth a workgroup and its barriers:
    // straight to the list of the serial fold (k_wide_fold: one lane per pair).  One global atomic per WORKGROUP and class (the four
    // counters share a cache line: one per wavefront was 30,000 atomics queueing on it, 0.17 ms at 479 k pairs).
    __shared__ uint32_t s_cnt[4], s_base[4];
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = r < a.npairs;
    WidePairAddr q{};
    uint32_t cls = 4;
    if (ok) { q = wide_pair_addr(a, r); q.pad = r; cls = q.mm <= kWideFoldTiny ? 3u : wide_fold_class(q.mm); }
    const uint32_t lane = lane_id();
    uint32_t at = 0;                                               // my place among the workgroup's pairs of my class
#pragma unroll
    for (uint32_t cc = 0; cc < 4; ++cc) {
        const unsigned long long mask = __ballot(cls == cc);
        if (!mask) continue;
        const uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&s_cnt[cc], (uint32_t)__popcll(mask));
        base = (uint32_t)__shfl((int)base, (int)leader, 64);
        if (cls == cc) at = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    if (threadIdx.x < 4 && s_cnt[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(&a.redo[a.npairs + (threadIdx.x == 3 ? 0u : 1u + threadIdx.x)], s_cnt[threadIdx.x]);
    __syncthreads();
    if (cls == 3) a.redo[s_base[3] + at] = r;
    else if (cls < 3) a.desc[(size_t)cls * a.npairs + s_base[cls] + at] = q;
}
__device__ __forceinline__ void wide_write_pair(const WideArgs& a, const WidePairAddr& q, const FoldResult& fr) {
    const uint32_t key = q.key, cid = q.cid;
    const uint32_t k = (uint32_t)a.k;
    const uint32_t seedH = fr.seed & 0xFFFFu, seedV = fr.seed >> 16;
    const uint64_t leH = kmer_le(a.packed, q.roffH + seedH, k);
    const uint64_t leV = kmer_le(a.packed, q.roffV + seedV, k);
    const uint32_t flags = (leH == leV ? 1u : 0u) | (kmer_rc_from_le(leH, k) == kmer_fw_from_le(leV, k) ? 2u : 0u);
    const uint64_t o = q.out;
    bella_pair pr;
    pr.rid = key; pr.cid = cid; pr.count = fr.count; pr.seedH = (uint16_t)seedH; pr.seedV = (uint16_t)seedV; pr.flags = (uint16_t)flags;
    a.tmp_pairs[o] = pr;
    if (a.tmp_ext) {
        bella_pair_ext ex;
        ex.nbins = fr.nbins; ex.support = fr.support; ex.binov = fr.binov; ex.pad = 0;
        a.tmp_ext[o] = ex;
    }
}

// The fold of one pair by one workgroup: the closed form of the row kernel's phase P (spgemm.hpp) on the pair's own list in HBM
// (uint2 {posH | posV << 16, overlap estimate | flags << 16}, product order):
//   parent(l) = first later product whose overlap estimate is within binSize of l's (none: a final bin, a "root");
//   a position walks its ancestor path and is dropped at the first ancestor within k of it in either coordinate; count = m + the
//   ancestors passed alive (mod 2^16); support(root) = positions that reach it; the winner is the first maximum among the roots in
//   descending product order (std::sort's insertion-sort regime, <= 16 bins).
// The upper half of a product's second word (the flags, which the record recomputes from the reads) serves as its support counter,
// a.sort_scratch as the parent links (u16: lists of >= 32768 products, and pairs with > 16 bins, go to the serial fold).
#ifndef BELLA_WIDE_FOLD_BLOCK
#define BELLA_WIDE_FOLD_BLOCK 512
#endif
#ifndef BELLA_WIDE_FOLD_W
#define BELLA_WIDE_FOLD_W 16
#endif
#ifndef BELLA_WIDE_FOLD_LDS
#define BELLA_WIDE_FOLD_LDS 4096
#endif
#ifndef BELLA_WIDE_VISIT
#define BELLA_WIDE_VISIT 6
#endif
#ifndef BELLA_WIDE_GRID_BUCKETS
#define BELLA_WIDE_GRID_BUCKETS 4096
#endif
constexpr int kWideFoldBlockLarge = BELLA_WIDE_FOLD_BLOCK;
constexpr uint32_t kWideFoldLdsLarge = BELLA_WIDE_FOLD_LDS, kGridBucketsLarge = BELLA_WIDE_GRID_BUCKETS;
constexpr uint32_t kGridMin = 96;                                         // lists of kGridMin < m <= (staged products, at most 4,096) use the grid
// Three instances share the pairs by list length (k_wide_desc sorts the descriptors into the classes): lists of up to kWideFoldSmall
// products are staged with 2,048 buckets in 22 KB of LDS -- six 256-thread workgroups per CU --, lists of up to kWideFoldMid in 44 KB
// (4,096 buckets, 512 threads, three per CU), the rest in 72 KB (two per CU, as all pairs until round 5).  A pair's ~8 phases are a
// chain of barriers and LDS round trips: the pairs in flight are what fills a CU (HiFi-like reads with -u 40: 72 % of the products
// stand in lists of 1,025 ... 2,048; that instance 4.33 -> 3.26 ms with three pairs per CU instead of two).
constexpr int kWideFoldBlockSmall = 256;
template <int kWideFoldBlock, uint32_t kWideFoldLds, uint32_t kGridBuckets, uint32_t kClass, int kMinWaves>
__global__ __launch_bounds__(kWideFoldBlock, kMinWaves) void k_wide_fold_wg(WideArgs a) {
    const uint32_t npairs = a.redo[a.npairs + 1 + kClass];          // this class's pairs, their descriptors
    const WidePairAddr* const desc = a.desc + (size_t)kClass * a.npairs;
    constexpr uint32_t kGridMax = kWideFoldLds < 4096 ? kWideFoldLds : 4096;
    __shared__ uint32_t s_flag, s_contrib, s_surv, s_roots;
    __shared__ unsigned long long s_best;
    __shared__ uint32_t s_hv[kWideFoldLds + 16];
    __shared__ uint16_t s_ov[kWideFoldLds + 16];
    // position grid of a long plain-chain list: per coordinate, the list positions counting-sorted by bucket floor(pos / 4) (hashed into
    // NB buckets).  s_cnt[b] packs both coordinates' counters (H low half, V high half: a list has at most 4,096 products); after the
    // scan and the scatter it holds the bucket ends, s_sorted {pos << 16 | list index} bucket by bucket, all of H, then all of V.
    // The products within k of a position stand in the buckets floor((pos - k) / 4) .. floor((pos + k) / 4), one contiguous range of
    // s_sorted, so "the first later product within k" is a scan of that range (about 1.1 x the products really within k) instead of
    // a walk over all later products (quadratic in the list).
    __shared__ __attribute__((aligned(16))) uint32_t s_cnt[kGridBuckets];
    constexpr uint32_t kVisit = BELLA_WIDE_VISIT;
    __shared__ uint32_t s_sorted[2 * kGridMax + 2 * kVisit];
    __shared__ uint32_t s_scr[kWideFoldBlock / 64];
    constexpr uint32_t kPend = 8;                                  // folded pairs waiting for their records to be written
    __shared__ FoldResult s_pfr[kPend];
    __shared__ uint32_t s_pr[kPend];
    uint32_t npend = 0;
    const uint32_t tid = threadIdx.x;
    constexpr uint32_t kRoot = 0x8000u;
    // Two pairs ahead: the descriptors (length, start) of the pair after the next one are in flight while the next pair's list is
    // loaded into registers (lists that fit the LDS staging) under the current pair's fold.
    constexpr uint32_t kPre = kWideFoldLds / kWideFoldBlock;
    static_assert(kWideFoldLds % kWideFoldBlock == 0, "a whole number of staged products per thread");
#ifdef BELLA_WF_CLOCK
    __shared__ unsigned long long c_acc[8];
    if (tid == 0) for (int i = 0; i < 8; ++i) c_acc[i] = 0;
    unsigned long long c_prev = clock64();
#define WFCLK(i) do { if (tid == 0) { const unsigned long long c_now = clock64(); c_acc[i] += c_now - c_prev; c_prev = c_now; } } while (0)
#else
#define WFCLK(i) do {} while (0)
#endif
    uint2 pre[kPre];
    WidePairAddr d1{};                                             // of the pair r
    uint32_t d2raw = 0;                                            // of the pair r + gridDim.x (mm = 0 beyond the last pair)
    if (blockIdx.x < npairs) d1 = desc[blockIdx.x];
    if (blockIdx.x + gridDim.x < npairs) d2raw = desc_load_raw(desc + blockIdx.x + gridDim.x);
    if (d1.mm <= kWideFoldLds) {
#pragma unroll
        for (uint32_t u = 0; u < kPre; ++u) { const uint32_t y = tid + u * kWideFoldBlock; if (y < d1.mm) pre[u] = a.plist[d1.lo + y]; }
    }
    for (uint32_t r = blockIdx.x; r < npairs; r += gridDim.x) {
        const uint32_t mm = d1.mm, rg = d1.pad;                    // (rg: the pair's index among all pairs)
        const uint64_t lo = d1.lo;
        uint2* w = a.plist + lo;
        const bool staged = mm <= kWideFoldLds;
        // a list that fits is staged in LDS (positions and overlap estimates): one streaming read of the pair's list
        if (staged) {
#pragma unroll
            for (uint32_t u = 0; u < kPre; ++u) {
                const uint32_t y = tid + u * kWideFoldBlock;
                if (u * kWideFoldBlock < mm) {                     // (wave-uniform: whole groups beyond the list are skipped by a scalar branch)
                    if (y < mm) { s_hv[y] = pre[u].x; s_ov[y] = (uint16_t)pre[u].y; }
                }
            }
        }
        d1 = desc_decode(d2raw);                                   // the next pair: its list now, the descriptor of the one after it
        if (d1.mm <= kWideFoldLds && r + gridDim.x < npairs) {
#pragma unroll
            for (uint32_t u = 0; u < kPre; ++u) {
                const uint32_t y = tid + u * kWideFoldBlock;
                if (u * kWideFoldBlock < d1.mm) { if (y < d1.mm) pre[u] = a.plist[d1.lo + y]; }
            }
        }
        d2raw = 0;
        if ((uint64_t)r + 2ull * gridDim.x < npairs) d2raw = desc_load_raw(desc + r + 2 * gridDim.x);
        if (mm >= kRoot) {                                         // parent links are u16
            if (tid == 0) a.redo[atomicAdd(&a.redo[a.npairs], 1u)] = rg;
            continue;
        }
        uint16_t* Par = a.sort_scratch + lo;
        if (tid == 0) { s_flag = 0; s_contrib = 0; s_surv = 0; s_roots = 0; s_best = 0; }
        const bool grid = mm > kGridMin && mm <= kGridMax && mm <= kWideFoldLds;    // (used if the chain turns out plain)
        constexpr uint32_t sh = 2;                                 // bucket width 4
        uint32_t NB = kWideFoldBlock;                              // buckets: >= 8 per product, a whole number per thread
        while (NB < 8 * mm && NB < kGridBuckets) NB <<= 1;
        if (grid) {
            if (NB == kGridBuckets) {
#pragma unroll
                for (uint32_t u = 0; u < kGridBuckets / kWideFoldBlock; u += 4) *(uint4*)&s_cnt[4 * tid + u * kWideFoldBlock] = make_uint4(0u, 0u, 0u, 0u);
            } else {
                for (uint32_t x = tid; x < NB; x += kWideFoldBlock) s_cnt[x] = 0;
            }
        }
        __syncthreads();
        WFCLK(0);
        // is it a plain chain (every product's parent is its successor)?  Only otherwise the parent links are stored and the flag
        // halves of the list words cleared (they become the support counters).  The grid is filled in the same sweep.
        for (uint32_t y = tid; y < mm; y += kWideFoldBlock) {
            if (y + 1 < mm) {
                const int ov0 = staged ? (int)s_ov[y] : (int)(w[y].y & 0xFFFFu), ov1 = staged ? (int)s_ov[y + 1] : (int)(w[y + 1].y & 0xFFFFu);
                if (!(iabs_(ov1 - ov0) < a.binSize)) s_flag = 1;
            }
            if (grid) {
                const uint32_t hv = s_hv[y];
                atomicAdd(&s_cnt[((hv & 0xFFFFu) >> sh) & (NB - 1)], 1u);
                atomicAdd(&s_cnt[((hv >> 16) >> sh) & (NB - 1)], 0x10000u);
            }
        }
        __syncthreads();
        WFCLK(1);
        if (s_flag) {
            for (uint32_t y = tid; y < mm; y += kWideFoldBlock) {
                const uint32_t ovw = staged ? (uint32_t)s_ov[y] : w[y].y & 0xFFFFu;
                w[y].y = ovw;
                uint32_t par = kRoot;
                for (uint32_t t = y + 1; t < mm; ++t)
                    if (iabs_((int)(staged ? (uint32_t)s_ov[t] : w[t].y & 0xFFFFu) - (int)ovw) < a.binSize) { par = t; break; }
                if (par == kRoot) atomicAdd(&s_roots, 1u);
                Par[y] = (uint16_t)par;
            }
            __syncthreads();
        }
        const bool plain = s_flag == 0;
        uint32_t contrib = 0, surv = 0;
        if (plain && grid) {
            const uint32_t kk = (uint32_t)a.k;
            {   // bucket counts -> bucket starts (one scan, both halves at once), then the scatter turns the starts into ends
                constexpr uint32_t kPer = kGridBuckets / kWideFoldBlock;
                static_assert(kPer % 4 == 0, "whole 16-byte words per thread");
                const uint32_t per = NB / kWideFoldBlock;          // 1 .. kPer consecutive buckets per thread
                uint32_t loc[kPer], sum = 0;
                if (per == kPer) {                                 // the usual case: 16-byte LDS accesses, no bank conflicts
#pragma unroll
                    for (uint32_t u = 0; u < kPer; u += 4) {
                        const uint4 q = *(const uint4*)&s_cnt[tid * kPer + u];
                        loc[u] = q.x; loc[u + 1] = q.y; loc[u + 2] = q.z; loc[u + 3] = q.w;
                    }
#pragma unroll
                    for (uint32_t u = 0; u < kPer; ++u) sum += loc[u];
                } else {
#pragma unroll
                    for (uint32_t u = 0; u < kPer; ++u) { loc[u] = u < per ? s_cnt[tid * per + u] : 0u; sum += loc[u]; }
                }
                uint32_t tot;
                uint32_t ex = block_excl_scan<kWideFoldBlock / 64>(sum, s_scr, &tot);
                if (per == kPer) {
#pragma unroll
                    for (uint32_t u = 0; u < kPer; u += 4) {
                        uint4 q;
                        q.x = ex; q.y = q.x + loc[u]; q.z = q.y + loc[u + 1]; q.w = q.z + loc[u + 2];
                        ex = q.w + loc[u + 3];
                        *(uint4*)&s_cnt[tid * kPer + u] = q;
                    }
                } else {
#pragma unroll
                    for (uint32_t u = 0; u < kPer; ++u) { if (u < per) s_cnt[tid * per + u] = ex; ex += loc[u]; }
                }
            }
            __syncthreads();
            WFCLK(2);
            for (uint32_t y = tid; y < mm; y += kWideFoldBlock) {
                const uint32_t hv = s_hv[y];
                s_sorted[atomicAdd(&s_cnt[((hv & 0xFFFFu) >> sh) & (NB - 1)], 1u) & 0xFFFFu] = (hv << 16) | y;
                s_sorted[mm + kVisit + (atomicAdd(&s_cnt[((hv >> 16) >> sh) & (NB - 1)], 0x10000u) >> 16)] = (hv & 0xFFFF0000u) | y;
            }
            // kVisit entries that match nothing behind either coordinate's part: the scans below read whole groups of kVisit
            if (tid < 2 * kVisit) s_sorted[tid < kVisit ? mm + tid : 2 * mm + tid] = 0xFFFFFFFFu;
            __syncthreads();
            WFCLK(3);
            typedef unsigned short us2 __attribute__((ext_vector_type(2)));
            const uint32_t lim = (2 * kk) << 16 | 0xFFFFu;         // high half (position - (pos - k)) <= 2k
            for (uint32_t y = tid; y + 1 < mm; y += kWideFoldBlock) {
                const uint32_t hv = s_hv[y];
                // distance (in list positions, minus one) to the first later product within k in either coordinate: the running minimum
                // sits in the low half of tt, both tests of an entry {pos << 16 | index} come out of one packed subtraction
                us2 tt = {(unsigned short)(mm - y - 1), 0};
#pragma unroll
                for (uint32_t cdn = 0; cdn < 2; ++cdn) {
                    const uint32_t pos = cdn ? hv >> 16 : hv & 0xFFFFu;
                    const uint32_t i0 = ((pos > kk ? pos - kk : 0u) >> sh) & (NB - 1), i1 = ((pos + kk) >> sh) & (NB - 1);
                    const uint32_t c0 = i0 ? s_cnt[i0 - 1] : 0u, c1 = s_cnt[i1];
                    const uint32_t sec = cdn ? mm + kVisit : 0u;   // where the coordinate's part of s_sorted begins
                    uint32_t xs = sec + (cdn ? c0 >> 16 : c0 & 0xFFFFu), xe = sec + (cdn ? c1 >> 16 : c1 & 0xFFFFu);
                    uint32_t xs2 = 0, xe2 = 0;
                    if (i0 > i1) { xs2 = sec; xe2 = xe; xe = sec + mm; }     // the bucket range wraps
                    const us2 ref = {(unsigned short)(y + 1), (unsigned short)(pos - kk)};
                    for (;;) {
                        // whole groups of kVisit entries: what stands behind xe lies in later buckets (positions beyond pos + k, or a
                        // multiple of 4 NB away) or is one of the kVisit fillers behind the part -- it fails the position test
                        for (uint32_t x = xs; x < xe; x += kVisit) {
                            uint32_t v[kVisit];
#pragma unroll
                            for (uint32_t u = 0; u < kVisit; ++u) v[u] = s_sorted[x + u];
#pragma unroll
                            for (uint32_t u = 0; u < kVisit; ++u) {
                                const us2 d = __builtin_bit_cast(us2, v[u]) - ref;     // {index - (y + 1), position - (pos - k)} mod 2^16
                                const uint32_t dw = __builtin_bit_cast(uint32_t, d);
                                const uint32_t cand = dw <= lim ? dw : 0xFFFFFFFFu;    // within k: the index distance competes (earlier products: >= 61,440)
                                tt = __builtin_elementwise_min(tt, __builtin_bit_cast(us2, cand));
                            }
                        }
                        if (xs2 == xe2) break;
                        xs = xs2; xe = xe2; xs2 = xe2 = 0;
                    }
                }
                const uint32_t dist = tt.x;                        // t - y - 1
                contrib += dist;
                surv += dist == mm - y - 1 ? 1u : 0u;
            }
        } else if (plain) {
            // plain chain: every position is compared with the later products until one is within k of it (16 per round trip)
            typedef unsigned short us2 __attribute__((ext_vector_type(2)));
            const us2 kk2 = {(unsigned short)a.k, (unsigned short)a.k};
            const us2 lim2 = {(unsigned short)(2 * a.k + 1), (unsigned short)(2 * a.k + 1)};
            const uint32_t lim = __builtin_bit_cast(uint32_t, lim2);
            constexpr uint32_t W = BELLA_WIDE_FOLD_W;
            for (uint32_t y = tid; y + 1 < mm; y += kWideFoldBlock) {
                const us2 xk = __builtin_bit_cast(us2, staged ? s_hv[y] : w[y].x) + kk2;
                uint32_t t = y + 1;
                uint32_t q[W];
                bool hit = false;
                while (t < mm) {
                    us2 acc = lim2;
#pragma unroll
                    for (uint32_t u = 0; u < W; ++u) {
                        const uint32_t idx = t + u < mm ? t + u : mm - 1;      // (copies of the last product: a hit there is the last one's)
                        q[u] = staged ? s_hv[idx] : w[idx].x;
                        acc = __builtin_elementwise_min(acc, (us2)(xk - __builtin_bit_cast(us2, q[u])));
                    }
                    if (__builtin_bit_cast(uint32_t, acc) != lim) { hit = true; break; }
                    t += W;
                }
                if (hit) {
                    uint32_t f = W - 1;
#pragma unroll
                    for (int u = (int)W - 2; u >= 0; --u) {
                        const us2 dd = __builtin_elementwise_min(lim2, (us2)(xk - __builtin_bit_cast(us2, q[u])));
                        f = __builtin_bit_cast(uint32_t, dd) != lim ? (uint32_t)u : f;
                    }
                    t += f;
                }
                t = t < mm ? t : mm;
                contrib += t - y - 1;
                surv += t == mm ? 1u : 0u;
            }
        } else {
            for (uint32_t y = tid; y + 1 < mm; y += kWideFoldBlock) {
                const uint32_t x = staged ? s_hv[y] : w[y].x;
                uint32_t root = y, t = Par[y];
                bool dead = false;
                while (!(t & kRoot)) {
                    if (!far_apart(x, staged ? s_hv[t] : w[t].x, a.k)) { dead = true; break; }
                    contrib++; root = t; t = Par[t];
                }
                if (!dead) atomicAdd(&w[root].y, 0x10000u);
            }
        }
        WFCLK(4);
        {   // one LDS atomic per wavefront, not per thread (512 same-address atomics queue up behind each other)
            const uint32_t cw = wave_incl_scan(contrib), sw = wave_incl_scan(surv);
            if (lane_id() == 63) { if (cw) atomicAdd(&s_contrib, cw); if (sw) atomicAdd(&s_surv, sw); }
        }
        __syncthreads();
        WFCLK(5);
        FoldResult fr;
        fr.many_bins = 0;
        fr.count = (uint16_t)(mm + s_contrib);
        uint32_t win = mm - 1;
        if (plain) {
            fr.nbins = 1; fr.support = (uint16_t)(s_surv + 1);
        } else {
            const uint32_t nroots = s_roots;
            if (nroots > 16) {                                     // std::sort leaves the insertion-sort regime (common.h:145)
                if (tid == 0) a.redo[atomicAdd(&a.redo[a.npairs], 1u)] = rg;
                __syncthreads();
                continue;
            }
            // first maximum in descending product order = the largest (support, product index)
            unsigned long long best = 0;
            for (uint32_t l = tid; l < mm; l += kWideFoldBlock) {
                if (!(Par[l] & kRoot)) continue;
                const uint32_t cnt = (w[l].y >> 16) + (l + 1 == mm ? 1u : 0u);   // the last product counts itself
                const unsigned long long key = ((unsigned long long)cnt << 32) | l;
                best = key > best ? key : best;
            }
            if (best) atomicMax(&s_best, best);
            __syncthreads();
            win = (uint32_t)s_best;
            fr.nbins = (uint16_t)nroots; fr.support = (uint16_t)(s_best >> 32);
        }
        // the record is not written here: reading the two seed k-mers and storing costs two trips to HBM that the whole workgroup
        // would wait for (and the store would hold up wavefront 0 at its next wait for loads); up to kPend results collect in LDS and
        // wavefront 0 writes them lane-parallel
        if (tid == 0) {
            if (staged) { fr.seed = s_hv[win]; fr.binov = s_ov[win]; }
            else { const uint2 e = w[win]; fr.seed = e.x; fr.binov = (uint16_t)(e.y & 0xFFFFu); }
            s_pfr[npend] = fr; s_pr[npend] = r;
        }
        ++npend;
        if (npend == kPend) {
            if (tid < kPend) wide_write_pair(a, desc[s_pr[tid]], s_pfr[tid]);
            npend = 0;
        }
        WFCLK(6);
        __syncthreads();
        WFCLK(7);
    }
    if (tid < npend) wide_write_pair(a, desc[s_pr[tid]], s_pfr[tid]);
#ifdef BELLA_WF_CLOCK
    if (tid == 0) for (int i = 0; i < 8; ++i) atomicAdd(&a.clk[i], c_acc[i]);
#endif
}

// the serial fold of the pairs k_wide_fold_wg left over: one lane per pair, serial statement of the semiring on the pair's own
// list (in place), libstdc++'s sort order for > 16 bins
__global__ __launch_bounds__(64) void k_wide_fold(WideArgs a) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= a.redo[a.npairs]) return;
    const uint32_t r = a.redo[x];
    const uint64_t lo = a.R_start[r];
    const uint32_t mm = a.R_len[r];
    uint32_t* w = (uint32_t*)(a.plist + lo);
    for (uint32_t t = 0; t < mm; ++t) w[2 * t + 1] &= 0xFFFFu;       // drop the flag / support halves
    struct S2 { uint32_t* base; __device__ uint32_t& operator[](uint32_t e) const { return base[2u * e]; } };
    FoldResult fr;
    fold_pair(S2{w}, S2{w + 1}, mm, a.k, a.binSize, a.sort_scratch + lo, fr);
    wide_write_pair(a, wide_pair_addr(a, r), fr);
}

// column ids of a tier's descriptor list (the columns above the LDS tiers join the wide columns)
__global__ void k_desc_cols(const uint4* desc, uint32_t n, uint32_t* out) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x < n) out[x] = desc[x].x;
}

}  // namespace bella
