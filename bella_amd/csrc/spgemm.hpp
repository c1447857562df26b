// spgemm.hpp -- C = A*A^T (strict lower triangle) under BELLA's position-binning semiring, gfx950.
//
// Replaces estimateFLOP (include/overlap.hpp:157-202), estimateNNZ_Hash (:205-276) and LocalSpGEMM
// (:281-363) with multiop/chainop (include/chain.hpp:74-150) of the reference.
//
// k_spgemm_rows_*  one 512-thread workgroup per output column i (= read i), bulk-synchronous phases in LDS
//                  (or in a global workspace for columns whose product list does not fit the LDS tiers):
//   X  expand   stream the column's B' entries (8 B each, coalesced); every entry carries a direct pointer to the
//               suffix "reads > i" of its k-mer's list in A' (8 B entries, contiguous): no column-pointer
//               indirection.  Products land in LDS in the reference's order (B slot order, then ascending read
//               id) with their u16 overlap estimate; their key (row id) goes into a grouping hash table T1.
//   O  order    emulate the reference's per-column open-addressing table (size 2^n >= max(16,nnz), hash key*107,
//               linear probe) to obtain its SLOT ORDER: keys are inserted in parallel with atomicMin on
//               (first-product-index, key): an entry with an earlier first occurrence displaces a later one,
//               which resumes probing -- the fixed point is exactly the layout sequential insertion produces.
//               A scan over the table gives every key its output rank and its list start.
//   S  scatter  wavefront 0 appends product indices to the per-pair lists, 64 products at a time: lists are ordered
//               across chunks, chunk-mates may be swapped.
//   R  rank     every product finds its exact rank inside its pair's list from its chunk-mates; the product-order
//               arrays are overlaid by the rank-order lists; single-product pairs (92 % at 100k reads) are
//               finished here: 16-byte record straight to the output.
//   P  fold     chainop in closed form (see the comment at phase P): parent links + independent walks, every
//               product of every pair is a lane; no per-pair state.
//   E  emit     one lane per pair writes the record (choose(): first maximum among <= 16 bins).
// k_fold_overflow  pairs that end with > 16 bins (std::sort's introsort regime): serial fold in HBM (core.hpp).
// (Columns with >= 65,536 products take the sort-based path of wide.hpp.)
//
// Data layout in HBM (built by assemble.hpp):
//   Bent[e] = { a_ptr, posV | cnt << 16 (14 bit) | pal << 30 | ori << 31 }  e in B' order (MergeDuplicates slot order)
//   rowF[i] = products of column i (the sum of its entries' counts: estimateFLOP, written once at layout time)
//   Aent[x] = { read | ori << 31, posH | readlen << 16 }  k-mer lists, ascending read id, stored in order of first
//                                                          appearance in B' (streaming for the owner row)
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "../../include/bella_hip.h"
#include "core.hpp"
#include "util.hpp"

namespace bella {

constexpr uint32_t kSU = 4;                               // chunks in flight per round of phase S
constexpr uint32_t kScatterChunk = 64;                    // phase S appends one wavefront of products at a time
#ifndef BELLA_WALK_W
#define BELLA_WALK_W 8
#endif
#ifndef BELLA_GATHER_BATCH
#define BELLA_GATHER_BATCH(blk) 2u                          // (round 4, expanding pass at 100k reads: 1 / 2 / 3 / 4 / 8 loads in flight per lane = 4.06 / 4.00 / 4.04 / 4.07 / 4.19 ms)
#endif
#ifndef BELLA_SCATTER_CHUNKS
#define BELLA_SCATTER_CHUNKS 8
#endif
#ifndef BELLA_SCATTER_WAVES
#define BELLA_SCATTER_WAVES 1
#endif
#ifndef BELLA_WALK_W_GLOBAL
#define BELLA_WALK_W_GLOBAL 32
#endif
#ifndef BELLA_ROW_BLOCK
#define BELLA_ROW_BLOCK 512
#endif
constexpr int kRowBlock = BELLA_ROW_BLOCK;                // threads per output column in the row kernels (1024 in the classes that hold
                                                          // one or two columns per CU: the CU's 32 wavefront slots stay filled)
// control block (u32 words) zeroed before every pass
constexpr uint32_t kCtlOverflow = 17;     // overflow list length
constexpr uint32_t kCtlWork2 = 18;        // k_fold_overflow counter
constexpr uint32_t kCtlStatus = 19;       // bit0: a pair ended with > 16 bins and no scratch was given
constexpr uint32_t kCtlRetry = 21;        // columns whose key table overflowed in an LDS tier (rerun on the global path)
constexpr uint32_t kCtlOrderBig = 22;     // columns with more than 1,024 pairs, left to k_order_block (order.hpp)
constexpr uint32_t kCtlTierCnt = 32;      // [16] columns per tier, last used entry = wide columns
constexpr uint32_t kCtlTotals = 48;       // u64[4]: nnz(C), products, products of the last tier's columns, of the wide columns (8-byte aligned)
constexpr uint32_t kCtlWords = 64;

struct SpgemmArgs {
    const uint32_t* rowlist;     // columns of this launch (retry launch) ...
    const uint4* rowdesc;        // ... or their descriptors {column, first B' entry, entries | read length << 16, -} (tier launches):
                                 // one load instead of a chain of three before the first B' entry can be fetched
    uint32_t nrows;
    uint32_t tier_lo, tier_hi;   // class launches: the tiers tier_hi .. tier_lo (big columns first); their lists are nreads apart
    uint32_t tcount[4];          // columns of the tiers tier_lo .. tier_lo + 3 (the host knows them at launch: no load before the descriptor)
    uint32_t tcount_valid;       // 0: the class spans more than four tiers, the counts are read from the control block
    uint32_t nreads;
    const uint32_t* Bptr;
    const uint2* Bent;
    uint32_t inl;                // the layout has B' entries in the INLINE form (util.hpp)
    const uint2* Aent;
    const uint2* Aent2;          // ready-made products (assemble.hpp: k_layout_rowlists), nullptr: expand B' x A' in the pass
    const uint16_t* Aov;         // the partner read's length of every product
    const uint64_t* Arow;        // [nreads + 1] first product of every column
    const uint64_t* roff;
    const uint32_t* packed;
    const uint64_t* flopptr;     // where a column's temporary output / product lists start: flops[i] slots, handed out by k_tier_lists
    const uint32_t* flops;       // products per column (estimateFLOP)
    bella_pair* tmp_pairs;
    bella_pair_ext* tmp_ext;
    uint32_t* nnzC;
    uint2* plist;                // [F] per-pair product lists {posH | posV << 16, overlap estimate}, product order
    uint4* overflow;             // descriptors {cid, key, start | m << 16, rank} of the pairs left to k_fold_overflow
    uint32_t* ctl;
    uint32_t* retry;             // [nreads] columns to rerun on the global path
    const uint32_t* nrows_dev;   // if set, the row count of this launch lives on the device
    uint8_t* ws;
    uint64_t ws_stride;
    uint32_t cap;
    uint32_t dcap;               // LDS tiers: pairs (distinct keys) the tier holds: cap/4 or cap/2
    int k;
    int binSize;
#ifdef BELLA_DEV_PROF
    unsigned long long* prof;    // development builds only: 10 counters per launch (phase cycles of wavefront 0, columns)
    int stop_phase;              // development builds only: leave the column after this phase (instruction counts per phase)
#endif
    int inject_unordered;        // tests: pretend every fifth column's lists came out of order in phase S (exercises the fallback)
};

constexpr uint32_t kRowScratchBytes = 256;                   // block scan scratch, counters
// cap = products held, dcap = distinct keys (pairs) held.  LDS tiers budget dcap = cap/2 or cap/4: a column with more pairs than
// that is rerun on the global path.
// overlay: the product-order arrays (A_hv, A_gov) are reused for the rank-order lists (LDS tiers: the values travel through
// registers between two barriers); without it (global path, any size) the lists get their own 8*cap bytes.
__host__ __device__ inline size_t row_mem_bytes(uint32_t cap, uint32_t dcap, bool overlay) {
    return kRowScratchBytes + (size_t)8 * cap + (size_t)16 * dcap + (overlay ? 2 : 4) * (size_t)((cap + 3) & ~3u) + 2 * (size_t)((dcap + 1) & ~1u) +
           (overlay ? 0 : (size_t)8 * cap + 2 * (size_t)((cap + 3) & ~3u));
}

struct RowMem {
    uint32_t* scr;      // 64 words: [0..15] scan, [48] chunk counter of the single-product sweep, [49] fail, [50] not-plain
    uint32_t* A_hv;     // [cap]  posH | posV << 16, product order
    uint32_t* A_gov;    // [cap]  T1 slot << 16 | overlap estimate ; LDS tiers (T1 slot < 2^13): flags << 30 on top
    uint32_t* T1key;    // [dcap]
    uint32_t* T1cnt;    // [dcap]  X: products of the key in product ranges 0 | 1 << 16 ; C..S: their scatter cursors ; then products | count << 16
    uint32_t* T1first;  // [dcap]  X: products in ranges 2 | 3 << 16 ; C..S: their scatter cursors ; then END of the pair's list | output index << 16
    uint32_t* Gaux;     // [dcap]  C..S: products | output index << 16 ; then plain chain: surviving positions (diagnostics), else (bins - 1) << 16
    uint16_t* S_p;      // [cap]   per-pair lists of product indices ; from phase P on: Par (u16 [cap]; global path u32 [cap])
    uint32_t* M;        // [dcap / 2] the multi-product pairs, dense: T1 slot | first product << 16 (its slot-order priority, k_order_*): the
                        //        per-pair passes after phase S (final words, emit) run over this list, not over the table's slots
    uint8_t* A_fl;      // [cap]  flags: bit0 oriented (checkstrand), bit1 palindromic k-mer   (global path only)
    uint8_t* L_fl;      // [cap]  the same in rank order                                        (global path only)
    uint32_t* L_hv;     // [cap]  rank-order lists: posH | posV << 16           (== A_hv when overlaid)
    uint32_t* L_gov;    // [cap]  rank-order lists: T1 slot << 16 | estimate     (== A_gov when overlaid)
    uint32_t cap, dcap;
};

__device__ __forceinline__ RowMem carve(uint8_t* base, uint32_t cap, uint32_t dcap, bool overlay) {
    RowMem m;
    m.scr = (uint32_t*)base;
    uint32_t* w = (uint32_t*)(base + kRowScratchBytes);
    m.A_hv = w;            w += cap;
    m.A_gov = w;           w += cap;
    m.T1key = w;           w += dcap;
    m.T1cnt = w;           w += dcap;
    m.T1first = w;         w += dcap;
    m.Gaux = w;            w += dcap;
    m.S_p = (uint16_t*)w;  w += (overlay ? 1 : 2) * (((cap + 3) & ~3u) / 2);
    m.M = w;               w += (dcap + 1) / 2;
    if (overlay) { m.L_hv = m.A_hv; m.L_gov = m.A_gov; m.A_fl = nullptr; m.L_fl = nullptr; }
    else { m.L_hv = w; w += cap; m.L_gov = w; w += cap; m.A_fl = (uint8_t*)w; w += (cap + 3) / 4; m.L_fl = (uint8_t*)w; }
    m.cap = cap;
    m.dcap = dcap;
    return m;
}

// One output column.  The records leave in the order of the grouping table's slots with the pair's first product index in the cid
// field: the reference's slot order is a function of (key, first product) alone and is produced afterwards by k_order_* for many
// columns at a time (the insertion cascades are chains of dependent LDS round trips: in here they would hold a whole workgroup).
// Returns false if the key table overflowed or a list came out of order (no global side effect happened yet; the caller queues
// the column again).
// RL: the column's products come ready-made from the row lists (arow = Arow[i], Fdesc = flops[i]); b0 / n / lenV are unused then.
template <bool OVERLAY, uint32_t NX, int BLK = BELLA_ROW_BLOCK, bool RL = false>
__device__ __forceinline__ bool process_row(const SpgemmArgs& a, const uint32_t i, const uint32_t b0, const uint32_t n, const uint32_t lenV,
                                            const RowMem& m, const uint64_t arow = 0, const uint32_t Fdesc = 0) {
    constexpr int kRowBlock = BLK;                            // threads of this workgroup
    constexpr int kRowWaves = BLK / 64;
    const uint32_t tid = threadIdx.x;
#ifdef BELLA_DEV_PROF                                          // development builds only (tools/prof_build.sh): wavefront 0's clock per phase
    long long tc_ = clock64();
#define BELLA_BPROF(n) { const long long t2_ = clock64(); if (tid == 0 && a.prof) atomicAdd(a.prof + (n), (unsigned long long)(t2_ - tc_)); \
                         if (a.stop_phase == (n)) { if (tid == 0) a.nnzC[i] = 0; return true; } tc_ = clock64(); }
#else
#define BELLA_BPROF(n)
#endif
    const uint32_t H1 = m.dcap;
    // where the column's records go: asked for HERE, as a scalar load (the column is wave-uniform), so that the trip is long over when
    // phase C first needs it -- issued where it is used (round 5) every wavefront of every column sat out one full memory latency there
    const uint64_t obase = a.flopptr[__builtin_amdgcn_readfirstlane((int)i)];
    uint32_t* s_chunk = m.scr + 48;
    uint32_t* s_fail = m.scr + 49;
    uint32_t* s_np = m.scr + 50;                              // LDS tiers: some pair of the column is not a plain chain
    uint32_t* s_nm = m.scr + 51;                              // multi-product pairs of the column (length of the dense list M)
    const uint32_t Mcap = (m.dcap + 1) / 2;
    const uint32_t k = (uint32_t)a.k;
    constexpr uint32_t GMASK = OVERLAY ? 0x1FFFu : 0xFFFFu;   // LDS tiers (<= 2752 slots): flags and the LAST mark ride on top of the T1 slot
    constexpr uint32_t kLastBit = 1u << 29;                   // L_gov (LDS tiers) / bit 2 of L_fl (global path): last product of its pair's list

    // the B' entries of the first round travel while the tables are initialised (their load is the second of two dependent HBM
    // round trips at the head of every column: descriptor, then entries)
#ifndef BELLA_RMAX_SMALL
#define BELLA_RMAX_SMALL 8
#endif
    // B' entries per thread and round (registers: 1024-thread workgroups run at 64 VGPRs).  A read of the PacBio-like sets has ~2,050
    // entries: a workgroup of 256 threads covers them in one round with ten entries per thread and needs a second round -- one more
    // dependent trip to HBM, one more scan -- with eight
    constexpr uint32_t RMAX = kRowBlock >= 1024 ? 4 : kRowBlock >= 512 ? 8 : BELLA_RMAX_SMALL;
    uint2 be0[RMAX];
    if (!RL) {
        const uint32_t nn = n < RMAX * kRowBlock ? n : RMAX * kRowBlock;
        const uint32_t R = (nn + kRowBlock - 1) / kRowBlock;
        const uint32_t j0 = tid * R;
#pragma unroll
        for (uint32_t u = 0; u < RMAX; ++u) {
            be0[u] = make_uint2(0u, 0u);
            if (u < R && j0 + u < nn) be0[u] = a.Bent[b0 + j0 + u];
        }
    }
    for (uint32_t s = tid; s < H1; s += kRowBlock) { m.T1key[s] = kEmpty; m.T1first[s] = 0; m.T1cnt[s] = 0; }
    if (tid == 0) { *s_chunk = 0; *s_fail = 0; *s_np = 0; *s_nm = 0; }
    __syncthreads();
#ifdef BELLA_DEV_PROF                                          // the head of the column: descriptor + first B' entries arrived, tables initialised
    { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const long long t2_ = clock64(); if (tid == 0 && a.prof) atomicAdd(a.prof + 8, (unsigned long long)(t2_ - tc_)); tc_ = t2_; }
#endif

    // ---- X: expand products in reference order, group keys --------------------------------------------
    // X1/X2: each thread owns a CONTIGUOUS run of B' entries (so one block scan orders all products); it only records,
    // per product, where its A' entry lives and the B' side word -- LDS writes, no dependent global loads.
    uint32_t running = RL ? Fdesc : 0u;
    if (!RL)
    for (uint32_t jb = 0; jb < n; jb += RMAX * kRowBlock) {
        const uint32_t nn = n - jb < RMAX * kRowBlock ? n - jb : RMAX * kRowBlock;
        const uint32_t R = (nn + kRowBlock - 1) / kRowBlock;
        const uint32_t j0 = tid * R;
        uint2 be[RMAX];
        uint32_t csum = 0;
#pragma unroll
        for (uint32_t u = 0; u < RMAX; ++u) {
            be[u] = be0[u];
            if (jb) { be[u] = make_uint2(0u, 0u); if (u < R && j0 + u < nn) be[u] = a.Bent[b0 + jb + j0 + u]; }
            csum += bent_count(be[u], a.inl);
        }
        uint32_t tot;
        uint32_t p = running + block_excl_scan<kRowWaves>(csum, m.scr, &tot);
#pragma unroll
        for (uint32_t u = 0; u < RMAX; ++u) {
            const uint32_t cnt = bent_count(be[u], a.inl);            // (an INLINE entry leaves itself: bit 31 of A_hv marks it)
            for (uint32_t t = 0; t < cnt; ++t) { m.A_hv[p] = be[u].x + t; m.A_gov[p] = be[u].y; ++p; }
        }
        running += tot;
    }
    const uint32_t F = running;
    // The products are cut into four ranges of whole 64-product chunks: a key counts its products per range (four u16 fields in
    // T1cnt / T1first), so that in phase S four wavefronts append to the per-pair lists independently, each in product order.
    const uint32_t RB = ((F + 255u) >> 8) << 6;
    __syncthreads();
    BELLA_BPROF(0)
    // X3: product-parallel gather of the A' entries (balanced: every lane has work; four independent loads in flight).  Per
    // product ONE compare-and-swap (claim or find the key's slot) and one add (the key's count in the product's range).
    constexpr uint32_t XB = BELLA_GATHER_BATCH(kRowBlock);    // A' loads in flight per lane
    for (uint32_t base = 0; base < F; base += XB * kRowBlock) {
        uint2 ae[XB];
        uint32_t bw[XB], ax[XB];
        bool inl[XB];
#pragma unroll
        for (uint32_t u = 0; u < XB; ++u) {
            const uint32_t p = base + u * kRowBlock + tid;
            ae[u] = make_uint2(0u, 0u);
            bw[u] = 0; ax[u] = 0;
            inl[u] = false;
            if (p < F) {
                if (RL) { ae[u] = a.Aent2[arow + p]; bw[u] = a.Aov[arow + p]; }
                else {
                    const uint32_t at = m.A_hv[p];
                    bw[u] = m.A_gov[p];
                    inl[u] = a.inl && (at >> 31);
                    ax[u] = at;
                    if (inl[u]) {                              // no gather: the partner's length only -- the two offsets travel like a gathered entry
                        const uint32_t* ro = (const uint32_t*)a.roff + 2u * (at & 0x3FFFFFFFu);    // (and are subtracted where it would be used)
                        ae[u] = make_uint2(ro[0], ro[2]);
                    } else ae[u] = a.Aent[at];
                }
            }
        }
        // Decode all products of the batch, then the FIRST probe of every product back to back (two compare-and-swaps in flight per
        // lane), then the rest of each insertion.  (Round 5 measured ONE probe loop for the whole batch: +35 %, spills.)
        uint32_t key[XB], hv[XB], ovf[XB], hh[XB], old[XB];
#pragma unroll
        for (uint32_t u = 0; u < XB; ++u) {
            const uint32_t p = base + u * kRowBlock + tid;
            key[u] = 0; hv[u] = 0; ovf[u] = 0; hh[u] = 0; old[u] = kEmpty;
            if (p >= F) continue;
            uint32_t ov, pal;
            bool oriented;
            if (RL) {                                          // {partner | pal << 30 | oriented << 31, posH | posV << 16}, partner's length
                key[u] = ae[u].x & 0x3FFFFFFFu; pal = (ae[u].x >> 30) & 1u; oriented = (ae[u].x >> 31) != 0; hv[u] = ae[u].y;
                ov = (uint32_t)overlap_estimate(hv[u] & 0xFFFFu, hv[u] >> 16, bw[u], lenV, oriented, k) & 0xFFFFu;
            } else if (inl[u]) {                               // the entry itself: partner | oriented << 30 | 1 << 31, posV | posH << 16
                key[u] = ax[u] & 0x3FFFFFFFu;
                const uint32_t posH = bw[u] >> 16, posV = bw[u] & 0xFFFFu;
                pal = 0;
                oriented = ((ax[u] >> 30) & 1u) != 0;
                ov = (uint32_t)overlap_estimate(posH, posV, ae[u].y - ae[u].x, lenV, oriented, k) & 0xFFFFu;
                hv[u] = posH | (posV << 16);
            } else {
                key[u] = ae[u].x & 0x7FFFFFFFu;
                const uint32_t posH = ae[u].y & 0xFFFFu, lenH = ae[u].y >> 16;
                const uint32_t posV = bw[u] & 0xFFFFu;
                pal = (bw[u] >> 30) & 1u;
                oriented = (ae[u].x >> 31) == (bw[u] >> 31);
                ov = (uint32_t)overlap_estimate(posH, posV, lenH, lenV, oriented, k) & 0xFFFFu;
                hv[u] = posH | (posV << 16);
            }
            ovf[u] = ov | (((oriented ? 1u : 0u) | (pal << 1)) << 30);
            hh[u] = hash_range(key[u], H1);
        }
#pragma unroll
        for (uint32_t u = 0; u < XB; ++u)
            if (base + u * kRowBlock + tid < F) old[u] = atomicCAS(&m.T1key[hh[u]], kEmpty, key[u]);
#pragma unroll
        for (uint32_t u = 0; u < XB; ++u) {
            const uint32_t p = base + u * kRowBlock + tid;
            if (p >= F) continue;
            uint32_t h = hh[u], o = old[u];
            uint32_t probes = 0;
            while (o != kEmpty && o != key[u]) {
                if (++probes == H1) break;
                h = (h + 1 == H1) ? 0 : h + 1;
                o = atomicCAS(&m.T1key[h], kEmpty, key[u]);
            }
            if (probes == H1) { *s_fail = 1; continue; }     // more pairs than this tier's key table holds
            const uint32_t q = (p >= RB ? 1u : 0u) + (p >= 2u * RB ? 1u : 0u) + (p >= 3u * RB ? 1u : 0u);
            atomicAdd((q & 2u) ? &m.T1first[h] : &m.T1cnt[h], (q & 1u) ? 0x10000u : 1u);
            m.A_hv[p] = hv[u];
            if (OVERLAY) m.A_gov[p] = (ovf[u] & 0xC000FFFFu) | (h << 16);
            else { m.A_gov[p] = (h << 16) | (ovf[u] & 0xFFFFu); m.A_fl[p] = (uint8_t)(ovf[u] >> 30); }
        }
    }
    __syncthreads();
    BELLA_BPROF(1)
    if (*s_fail) return false;

    // ---- C: every pair gets its output index (table-slot order), the multi-product pairs their list and four scatter cursors ----
    uint32_t Fm, d;                                           // products of the multi-product pairs = total list length; pairs
    {
        const uint32_t c = (H1 + kRowBlock - 1) / kRowBlock;
        const uint32_t lo = tid * c < H1 ? tid * c : H1;
        const uint32_t hi = lo + c < H1 ? lo + c : H1;
        uint32_t occ = 0, csum = 0;
        for (uint32_t s = lo; s < hi; ++s) {
            const uint32_t w0 = m.T1cnt[s], w1 = m.T1first[s];
            const uint32_t mm = (w0 & 0xFFFFu) + (w0 >> 16) + (w1 & 0xFFFFu) + (w1 >> 16);
            occ += mm ? 1u : 0u;
            csum += mm > 1 ? mm : 0u;
        }
        uint32_t tot;
        const uint32_t ex = block_excl_scan<kRowWaves>((occ << 16) | csum, m.scr, &tot);
        Fm = tot & 0xFFFFu;
        d = tot >> 16;
        uint32_t rank = ex >> 16, st = ex & 0xFFFFu;
        for (uint32_t s = lo; s < hi; ++s) {
            const uint32_t w0 = m.T1cnt[s], w1 = m.T1first[s];
            const uint32_t c0 = w0 & 0xFFFFu, c1 = w0 >> 16, c2 = w1 & 0xFFFFu, c3 = w1 >> 16;
            const uint32_t mm = c0 + c1 + c2 + c3;
            if (!mm) continue;
            m.Gaux[s] = mm | (rank << 16);
            if (mm > 1) {
                m.T1cnt[s] = st | ((st + c0) << 16);
                m.T1first[s] = (st + c0 + c1) | ((st + c0 + c1 + c2) << 16);
                st += mm;
                const uint32_t j = atomicAdd(s_nm, 1u);       // (a few per wavefront instruction: one pair in twenty-five has more than one product)
                if (j < Mcap) m.M[j] = s;
            }
            rank++;
        }
    }
    __syncthreads();
    BELLA_BPROF(2)

    // ---- S: product indices into per-pair lists.  Range q is appended by ONE wavefront, 64 products at a time in product order
    // (LDS atomics of one wavefront execute in program order; within one instruction the same-address atomics are applied in lane
    // order -- observed, not promised: phase R verifies), from the cursor C gave the key for that range: a list is in product
    // order.  Meanwhile the other wavefronts (all of them, once the scatter is done) sweep the products for the pairs with a single
    // product and write their records (multiop only: count 1, one bin, seed = the k-mer; 47 % of the pairs at 10k reads, 92 % at 100k).
    uint16_t* S_p = m.S_p;
    {
        constexpr uint32_t kSA = BELLA_SCATTER_CHUNKS;
        constexpr uint32_t kOwners = kRowWaves < 4 ? kRowWaves : 4;
        const uint32_t w = wave_id(), lane = lane_id();
        if (w < kOwners && Fm) {
            for (uint32_t q = w; q < 4; q += kOwners) {
                const uint32_t p0 = q * RB, p1 = (q + 1) * RB < F ? (q + 1) * RB : F;
                uint32_t* const curs = (q & 2u) ? m.T1first : m.T1cnt;
                const uint32_t inc = (q & 1u) ? 0x10000u : 1u, sh = (q & 1u) * 16u;
                for (uint32_t base = p0; base < p1; base += kSA * kScatterChunk) {
                    uint32_t g[kSA], old[kSA];
                    bool mine[kSA];
#pragma unroll
                    for (uint32_t u = 0; u < kSA; ++u) {
                        const uint32_t p = base + u * kScatterChunk + lane;
                        g[u] = m.A_gov[p < p1 ? p : p1 - 1];
                    }
#pragma unroll
                    for (uint32_t u = 0; u < kSA; ++u) g[u] = (g[u] >> 16) & GMASK;
#pragma unroll
                    for (uint32_t u = 0; u < kSA; ++u) {
                        const uint32_t p = base + u * kScatterChunk + lane;
                        mine[u] = p < p1 && ((const uint16_t*)m.Gaux)[2u * g[u]] != 1u;   // single-product pairs have no list
                    }
                    // (old stays unset for the other lanes on purpose: with a default value the compiler folds the first use into the
                    // predicated block and waits there)
#pragma unroll
                    for (uint32_t u = 0; u < kSA; ++u)
                        if (mine[u]) old[u] = atomicAdd(&curs[g[u]], inc);
#pragma unroll
                    for (uint32_t u = 0; u < kSA; ++u)
                        if (mine[u]) S_p[(old[u] >> sh) & 0xFFFFu] = (uint16_t)(base + u * kScatterChunk + lane);
                }
            }
        }
        {
            constexpr uint32_t kSweep = 4 * 64;               // products per grab
            for (;;) {
                uint32_t c0 = 0;
                if (lane == 0) c0 = atomicAdd(s_chunk, kSweep);
                c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)c0);
                if (c0 >= F) break;
#pragma unroll
                for (uint32_t u = 0; u < kSweep / 64; ++u) {
                    const uint32_t p = c0 + u * 64 + lane;
                    if (p >= F) continue;
                    const uint32_t gov = m.A_gov[p];
                    const uint32_t g = (gov >> 16) & GMASK;
                    const uint32_t ga = m.Gaux[g];
                    if ((ga & 0xFFFFu) != 1u) continue;
                    const uint32_t hv = m.A_hv[p];
                    const uint32_t fl = OVERLAY ? gov >> 30 : (uint32_t)m.A_fl[p];
                    // the 16-byte record {rid, cid = first product, count = 1, seedH, seedV, flags} as ONE store (field by field the
                    // compiler writes a bella_pair with four: 184 M store requests per pass at 100k reads instead of 46 M)
                    const uint32_t flags = (fl & 1u) | ((((fl & 1u) ^ 1u) | (fl >> 1)) << 1);       // bit1: revcomp(seedH)==seedV
                    *(uint4*)(a.tmp_pairs + obase + (ga >> 16)) = make_uint4(m.T1key[g], p, 1u | ((hv & 0xFFFFu) << 16), (hv >> 16) | (flags << 16));
                    if (a.tmp_ext) {
                        bella_pair_ext ex2;
                        ex2.nbins = 1; ex2.support = 1; ex2.binov = (uint16_t)(gov & 0xFFFFu); ex2.pad = 0;
                        a.tmp_ext[obase + (ga >> 16)] = ex2;
                    }
                }
            }
        }
    }
    __syncthreads();
    BELLA_BPROF(3)

    // ---- C2 + R: the per-pair words take their final meaning (END of list | output index, products | count, first product); every
    // product's exact rank inside its pair's list (global path: list position corrected by the chunk-mates on the wrong side; LDS
    // tiers: checked) and the lists in rank order: L_hv / L_gov. -------------------------------------------------------------
    const uint32_t nm = *s_nm;
    // More multi-product pairs than the dense list holds (dcap / 2: columns whose pairs mostly have several products -- low-error or
    // deep-coverage input): the two per-pair passes run over the table's slots instead, as they did before the list existed, and the
    // list's storage holds the pairs' first products by SLOT (dcap halves of a word: a product index has 16 bits).  Round 5 sent such a
    // column to the global-workspace rerun.  (The slot-scan loops are loops of their own and touch LDS only: a global load in phase E --
    // the first version carried the first product through the record's cid field -- costs every column a wait for the record stores
    // of phase S, 7 % of the pass, whether the loop runs or not.)
    const bool dense = nm <= Mcap;
    auto c2_one = [&](const uint32_t s, const uint32_t j, const bool by_slot) {
        const uint32_t ga = m.Gaux[s];
        const uint32_t mm = ga & 0xFFFFu;
        const uint32_t end = m.T1first[s] >> 16;              // range 3's cursor ran to the end of the list
        uint32_t fp = S_p[end - mm];
        if (!OVERLAY) {                                       // chunk-mates may be swapped here (phase R repairs): the smallest of the first chunk
            const uint32_t ch = fp / kScatterChunk;
            for (uint32_t y = end - mm + 1; y < end; ++y) {
                const uint32_t o = S_p[y];
                if (o / kScatterChunk != ch) break;
                fp = o < fp ? o : fp;
            }
        }
        if (by_slot) ((uint16_t*)m.M)[s] = (uint16_t)fp;
        else m.M[j] = s | (fp << 16);
        m.T1first[s] = end | (ga & 0xFFFF0000u);
        m.T1cnt[s] = mm | (mm << 16);
        m.Gaux[s] = 0;
    };
    if (dense) {
        for (uint32_t j = tid; j < nm; j += kRowBlock) c2_one(m.M[j], j, false);
    } else {
        for (uint32_t s = tid; s < H1; s += kRowBlock)
            if (m.T1key[s] != kEmpty && (m.Gaux[s] & 0xFFFFu) >= 2u) c2_one(s, 0u, true);
    }
    if (!OVERLAY) __syncthreads();
    // NX = list positions per thread in the LDS tiers (cap <= NX * kRowBlock)
    uint32_t hvv[NX], govv[NX];
    auto rank_one = [&](uint32_t x, uint32_t& dst, uint32_t& hvq, uint32_t& govq, uint32_t& flq) {
        const uint32_t p = S_p[x];
        const uint32_t gov = m.A_gov[p];
        const uint32_t g = (gov >> 16) & GMASK;
        const uint32_t end = m.T1first[g] & 0xFFFFu;
        const uint32_t hv = m.A_hv[p];
        uint32_t fl = (uint32_t)m.A_fl[p];
        const uint32_t mm = m.T1cnt[g] & 0xFFFFu;
        const uint32_t st = end - mm;
        const uint32_t ch = p / kScatterChunk;
        uint32_t rk = x - st;
        for (uint32_t y = x; y > st; --y) {                   // chunk-mates on the left that belong after p
            const uint32_t o = S_p[y - 1];
            if (o / kScatterChunk != ch) break;
            rk -= (o > p);
        }
        for (uint32_t y = x + 1; y < end; ++y) {              // chunk-mates on the right that belong before p
            const uint32_t o = S_p[y];
            if (o / kScatterChunk != ch) break;
            rk += (o < p);
        }
        dst = st + rk; hvq = hv; govq = gov; flq = fl | (dst + 1 == end ? 4u : 0u);
    };
    if (OVERLAY) {
        // LDS tiers: verify instead of repair.  On gfx950 the same-address LDS atomics of one wavefront instruction are applied in
        // lane order (0 of 1.2e7 products ever needed the repair), but that is not an architectural promise: a list is in product
        // order iff every entry is below its right neighbour; if one is not, the column is redone on the global path, which repairs.
        // Straight-line stages over groups of four of the thread's NX list positions.  A list ends where the next position's
        // product belongs to another key slot.
        constexpr uint32_t kRB = 4;
#pragma unroll
        for (uint32_t u0 = 0; u0 < NX; u0 += kRB) {
            uint32_t pv[kRB], pn[kRB], gn[kRB];
            bool bad[kRB];
#pragma unroll
            for (uint32_t v = 0; v < kRB; ++v) {
                const uint32_t x = tid + (u0 + v) * kRowBlock;
                const bool ok = u0 + v < NX && x < Fm;
                const uint32_t xc = ok ? x : 0u, xn = ok && x + 1 < Fm ? x + 1 : xc;
                pv[v] = S_p[xc]; pn[v] = S_p[xn];
            }
#pragma unroll
            for (uint32_t v = 0; v < kRB; ++v) {
                const uint32_t x = tid + (u0 + v) * kRowBlock;
                const bool ok = u0 + v < NX && x < Fm;
                pv[v] = ok ? pv[v] : 0u; pn[v] = ok ? pn[v] : 0u;
                bad[v] = pn[v] < pv[v];
            }
#pragma unroll
            for (uint32_t v = 0; v < kRB; ++v)
                if (u0 + v < NX) { govv[u0 + v] = m.A_gov[pv[v]]; hvv[u0 + v] = m.A_hv[pv[v]]; gn[v] = m.A_gov[pn[v]]; }
#pragma unroll
            for (uint32_t v = 0; v < kRB; ++v) {
                if (u0 + v >= NX) continue;
                const uint32_t x = tid + (u0 + v) * kRowBlock;
                if (x < Fm) {
                    const bool last = x + 1 == Fm || ((gn[v] >> 16) & GMASK) != ((govv[u0 + v] >> 16) & GMASK);
                    if (!last && bad[v]) *s_fail = 1;
                    // the successor's overlap estimate is at hand: is it this product's parent (phase P)?  If that holds for every
                    // product of the column, all its pairs are plain chains and the parent phase is left out
                    if (!last && !(iabs_((int)(gn[v] & 0xFFFFu) - (int)(govv[u0 + v] & 0xFFFFu)) < a.binSize)) *s_np = 1;
                    if (last) govv[u0 + v] |= kLastBit;
                }
            }
        }
        if (a.inject_unordered && i % 5u == 2u && tid == 0) *s_fail = 1;
        __syncthreads();                                      // every A_hv / A_gov read is done: reuse them as L_hv / L_gov
        if (*s_fail) return false;                            // a list out of order (see above): only single-product records were written
#pragma unroll
        for (uint32_t u = 0; u < NX; ++u) {
            const uint32_t x = tid + u * kRowBlock;
            if (x < Fm) { m.L_hv[x] = hvv[u]; m.L_gov[x] = govv[u]; }
        }
    } else {
        for (uint32_t x = tid; x < Fm; x += kRowBlock) {
            uint32_t dst, hvq, govq, flq;
            rank_one(x, dst, hvq, govq, flq);
            m.L_hv[dst] = hvq; m.L_gov[dst] = govq; m.L_fl[dst] = (uint8_t)flq;
        }
    }
    __syncthreads();
    BELLA_BPROF(4)

    // ---- P: the fold, in parallel.  chainop (chain.hpp:100-150) on a pair's products in order has a closed form:
    //  * every product t opens a bin with overlap ov_t and itself as first position; a bin lives, unchanged, until the first
    //    later product whose estimate is within binSize of ITS overlap (chain.hpp:114) dissolves it into that product's bin:
    //        parent(l) = min { t > l : |ov_l - ov_t| < binSize }          (none: the bin survives = a final bin, a "root")
    //  * a position s is only ever compared with the products on its ancestor path s -> parent(s) -> parent(parent(s)) ...;
    //    at each of them it is dropped if within k in either coordinate (chain.hpp:88-97,121) and otherwise moves along:
    //        count = m + sum_s #(ancestors passed alive)   (mod 2^16)     (chain.hpp:104,140)
    //        support(root) = #(positions that reach it alive, the root's own product included)
    //  * the final bins in the reference's order are the roots by descending product index (a new bin is inserted in front,
    //    orphans keep their order: chain.hpp:137-149), each headed by its own product: choose()'s seed (common.h:162-170).
    // Every step is a "first later match" search: no serial dependency between the products of a pair.  For the usual
    // single-bin pair parent(l) = l+1 and the walk is a forward scan.
    // Bookkeeping: bit 31 of T1key marks a pair whose path is not the plain chain l -> l+1 (then Gaux >> 16 counts its roots
    // other than the last product, and Sup[] collects the support of each root).
    // Par[y]: parent of position y (index inside its pair's list) or kRoot | support.  LDS tiers: u16 (lists <= 4096 long)
    typedef typename std::conditional<OVERLAY, uint16_t, uint32_t>::type par_t;
    constexpr uint32_t kRoot = OVERLAY ? 0x8000u : 0x80000000u;
    par_t* Par = (par_t*)m.S_p;                               // S_p is dead
    // (Par holds ABSOLUTE list positions; a list ends at the entry carrying the LAST mark: no per-pair look-up on the fast path)
    auto is_last = [&](uint32_t y, uint32_t gov) -> bool { return OVERLAY ? (gov & kLastBit) != 0 : (m.L_fl[y] & 4u) != 0; };
    const bool need_parents = !OVERLAY || *s_np != 0;         // (wave-uniform: read after the barrier that follows phase R)
    if (need_parents)
    for (uint32_t y = tid; y < Fm; y += kRowBlock) {
        const uint32_t gov = m.L_gov[y];
        uint32_t gt = y + 1 < Fm ? m.L_gov[y + 1] : 0u;       // the usual parent: in flight together with the entry itself
        uint32_t par = kRoot;
        if (!is_last(y, gov)) {
            const int ovs = (int)(gov & 0xFFFFu);
            for (uint32_t t = y + 1;; ++t) {
                if (iabs_((int)(gt & 0xFFFFu) - ovs) < a.binSize) { par = t; break; }
                if (is_last(t, gt)) break;
                gt = m.L_gov[t + 1];
            }
            if (par != y + 1) {
                const uint32_t g = (gov >> 16) & GMASK;
                atomicOr(&m.T1key[g], 0x80000000u);
                if (par == kRoot) atomicAdd(&m.Gaux[g], 0x10000u);
            }
        }
        Par[y] = (par_t)par;
    }
    if (need_parents) __syncthreads();
    BELLA_BPROF(5)
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const us2 kk2 = {(unsigned short)a.k, (unsigned short)a.k};
    const us2 lim2 = {(unsigned short)(2 * a.k + 1), (unsigned short)(2 * a.k + 1)};
    const uint32_t lim = __builtin_bit_cast(uint32_t, lim2);
    for (uint32_t y = tid; y < Fm; y += kRowBlock) {
        const uint32_t gov = m.L_gov[y];
        const uint32_t x = m.L_hv[y];
        if (is_last(y, gov)) continue;                        // nothing after it: no comparison, no contribution
        const uint32_t g = (gov >> 16) & GMASK;
        const uint32_t mm = m.T1first[g] & 0xFFFFu;           // END of the list: the walk runs on absolute positions s = y .. mm
        const uint32_t* lst = m.L_hv;
        const uint32_t s = y;
        if (!need_parents || !(m.T1key[g] >> 31)) {           // (all-plain column: no look-up)
            // plain chain: the position is compared with every later product until one is within k of it.
            // within k  <=>  (x + k - q) mod 2^16 <= 2k  in either half (k-mer starts are <= 65535 - k): eight products per test
            const us2 xk = __builtin_bit_cast(us2, x) + kk2;
            // The last group of a list reads up to W - 1 words beyond its end (the next list, or the words behind the lists: always
            // inside the column's arrays); a "hit" found there lies at or beyond the end and is cut off below -- no masked tail.
            uint32_t t = s + 1;
            constexpr uint32_t W = OVERLAY ? BELLA_WALK_W : BELLA_WALK_W_GLOBAL;   // products per test (global path: latency per round trip, not issue, is what counts)
            uint32_t q[W];
            bool hit = false;
            while (t < mm) {
                us2 acc = lim2;
#pragma unroll
                for (uint32_t u = 0; u < W; ++u) {
                    q[u] = lst[t + u];
                    acc = __builtin_elementwise_min(acc, (us2)(xk - __builtin_bit_cast(us2, q[u])));
                }
                if (__builtin_bit_cast(uint32_t, acc) != lim) { hit = true; break; }
                t += W;
            }
            if (hit) {                                        // first product of the group that is within k
                uint32_t f = W - 1;
#pragma unroll
                for (int u = (int)W - 2; u >= 0; --u) {
                    const us2 dd = __builtin_elementwise_min(lim2, (us2)(xk - __builtin_bit_cast(us2, q[u])));
                    f = __builtin_bit_cast(uint32_t, dd) != lim ? (uint32_t)u : f;
                }
                t += f;
            }
            t = t < mm ? t : mm;                              // ran off the end, or the first product within k lies beyond it
            const uint32_t contrib = t - s - 1;
            if (contrib) atomicAdd(&m.T1cnt[g], (contrib & 0xFFFFu) << 16);   // the cursor half already holds m
            if (a.tmp_ext && t == mm) atomicAdd(&m.Gaux[g], 1u);
        } else {
            const par_t* pr = Par;
            uint32_t root = s, contrib = 0, t = pr[s];
            bool dead = false;
            while (!(t & kRoot)) {
                if (!far_apart(x, lst[t], a.k)) { dead = true; break; }
                contrib++; root = t; t = pr[t];
            }
            if (contrib) atomicAdd(&m.T1cnt[g], (contrib & 0xFFFFu) << 16);
            if (!dead) {                                      // the root's support (a u16 counter is half of an LDS word)
                if (OVERLAY) atomicAdd((uint32_t*)Par + (root >> 1), (root & 1u) ? 0x10000u : 1u);
                else atomicAdd((uint32_t*)Par + root, 1u);
            }
        }
    }
    __syncthreads();
    BELLA_BPROF(6)

    // ---- E: one record per multi-product pair, at the pair's output index (the single-product pairs left in phase S) -------------
    auto e_one = [&](const uint32_t g, const uint32_t firstp) {
        const uint32_t cw = m.T1cnt[g];
        const uint32_t mm = cw & 0xFFFFu;
        const uint32_t aux = m.Gaux[g];
        const uint32_t tf = m.T1first[g];
        const uint32_t r = tf >> 16;                          // output index
        const uint32_t st = (tf & 0xFFFFu) - mm;
        const uint32_t keyw = m.T1key[g];
        // plain chain: one bin, headed by the last product (which phase P skips: it always survives)
        uint32_t win = mm - 1, sup = (aux & 0xFFFFu) + 1, nroots = 1;
        if (keyw >> 31) {
            nroots = (aux >> 16) + 1;
            if (nroots > 16) {
                // std::sort is not stable past 16 elements (common.h:145): hand the pair's list to the serial fold, which
                // reproduces libstdc++'s order exactly (it keeps the cid field: the first product index)
                for (uint32_t t = 0; t < mm; ++t) a.plist[obase + st + t] = make_uint2(m.L_hv[st + t], m.L_gov[st + t] & 0xFFFFu);
                a.tmp_pairs[obase + r].cid = firstp;
                a.overflow[atomicAdd(&a.ctl[kCtlOverflow], 1u)] = make_uint4(i, keyw & 0x7FFFFFFFu, st | (mm << 16), r);
                return;
            }
            // bins in the reference's order = roots by descending product index; std::sort by support (desc) on <= 16 bins is
            // an insertion sort: the first maximum wins
            sup = 0;
            for (uint32_t l = mm; l-- > 0;) {
                const uint32_t pw = Par[st + l];
                if (!(pw & kRoot)) continue;
                const uint32_t cnt = (pw & (kRoot - 1)) + (l + 1 == mm ? 1u : 0u);   // the last product counts itself
                if (cnt > sup) { sup = cnt; win = l; }
            }
        }
        const uint32_t hv = m.L_hv[st + win];
        const uint32_t fl = OVERLAY ? m.L_gov[st + win] >> 30 : (uint32_t)m.L_fl[st + win] & 3u;
        const uint32_t flags = (fl & 1u) | ((((fl & 1u) ^ 1u) | (fl >> 1)) << 1);
        *(uint4*)(a.tmp_pairs + obase + r) = make_uint4(keyw & 0x7FFFFFFFu, firstp, (cw >> 16) | ((hv & 0xFFFFu) << 16), (hv >> 16) | (flags << 16));
        if (a.tmp_ext) {
            bella_pair_ext ex;
            ex.nbins = (uint16_t)nroots; ex.support = (uint16_t)sup; ex.binov = (uint16_t)(m.L_gov[st + win] & 0xFFFFu); ex.pad = 0;
            a.tmp_ext[obase + r] = ex;
        }
    };
    if (dense) {
        for (uint32_t j = tid; j < nm; j += kRowBlock) { const uint32_t mj = m.M[j]; e_one(mj & 0xFFFFu, mj >> 16); }
    } else {
        // (slot scan: a single-product pair still holds its X-phase counts in T1cnt -- 0 or 1 in the low half; an empty slot 0)
        for (uint32_t g = tid; g < H1; g += kRowBlock)
            if (m.T1key[g] != kEmpty && (m.T1cnt[g] & 0xFFFFu) >= 2u) e_one(g, ((const uint16_t*)m.M)[g]);
    }
    if (tid == 0) a.nnzC[i] = d;
    BELLA_BPROF(7)
#ifdef BELLA_DEV_PROF
    if (tid == 0 && a.prof) atomicAdd(a.prof + 9, 1ull);
#endif
#undef BELLA_BPROF
    return true;
}

// LDS tiers: one column per workgroup, dynamic LDS = row_mem_bytes(cap, dcap); NX * 512 >= cap (8: the tiers up to 4096 products,
// 16: the two big tiers that keep HiFi-like columns with long lists off the global path)
template <uint32_t NX, int BLK = BELLA_ROW_BLOCK, bool RL = false>
__global__ __launch_bounds__(BLK, (BLK == 64 ? (NX <= 11 ? 4 : 2) : BLK == 1024 ? (NX <= 4 ? 8 : 4) : BLK <= 256 ? 8 : (NX <= 8 ? 6 : 2))) void k_spgemm_rows_lds(SpgemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // the launch covers the tiers of one LDS class, largest columns first: workgroup x -> (tier, place in the tier's list)
    uint32_t x = blockIdx.x, t = a.tier_hi;
    if (a.tcount_valid) {
#pragma unroll
        for (int k2 = 3; k2 >= 1; --k2)
            if (a.tier_lo + (uint32_t)k2 == t && x >= a.tcount[k2]) { x -= a.tcount[k2]; --t; }
        const uint32_t rel = t - a.tier_lo;
        const uint32_t ct = rel == 0 ? a.tcount[0] : rel == 1 ? a.tcount[1] : rel == 2 ? a.tcount[2] : a.tcount[3];
        if (x >= ct) return;
    } else {
        const uint32_t* tc = a.ctl + kCtlTierCnt;
        while (t > a.tier_lo && x >= tc[t]) { x -= tc[t]; --t; }
        if (x >= tc[t]) return;
    }
    const uint4 ds = a.rowdesc[(size_t)t * a.nreads + x];
    const uint32_t i = ds.x;
    const RowMem m = carve(smem, a.cap, a.dcap, true);
    // row lists: the descriptor carries the column's first product (48 bits: y, low half of z) and its product count (w)
    const bool ok = RL ? process_row<true, NX, BLK, true>(a, i, 0u, 0u, ds.z >> 16, m, (uint64_t)ds.y | ((uint64_t)(ds.z & 0xFFFFu) << 32), ds.w)
                       : process_row<true, NX, BLK, false>(a, i, ds.y, ds.z & 0xFFFFu, ds.z >> 16, m);
    if (!ok && threadIdx.x == 0) a.retry[atomicAdd(&a.ctl[kCtlRetry], 1u)] = i;
}

// Global-workspace path: columns with more products than the largest LDS tier (< 65536), and -- second launch, list and
// length produced on the device -- columns whose key table overflowed in an LDS tier.  Persistent workgroups.
#ifndef BELLA_GLOBAL_BLOCK
#define BELLA_GLOBAL_BLOCK 1024
#endif
constexpr int kGlobalBlock = BELLA_GLOBAL_BLOCK;          // threads per column on the global-workspace path
template <bool RL>
__global__ __launch_bounds__(kGlobalBlock) void k_spgemm_rows_global(SpgemmArgs a) {
    uint8_t* ws = a.ws + (uint64_t)blockIdx.x * a.ws_stride;
    const uint32_t nrows = a.nrows_dev ? *a.nrows_dev : a.nrows;
    for (uint32_t x = blockIdx.x; x < nrows; x += gridDim.x) {
        const uint32_t i = a.rowdesc ? a.rowdesc[x].x : a.rowlist[x];
        uint32_t f = a.flops[i];
        if (f < 16u) f = 16u;
        const RowMem m = carve(ws, f, f, false);
        const uint32_t b0 = a.Bptr[i];
        if (RL) (void)process_row<false, 8, kGlobalBlock, true>(a, i, 0u, 0u, (uint32_t)(a.roff[i + 1] - a.roff[i]), m, a.Arow[i], a.flops[i]);
        else (void)process_row<false, 8, kGlobalBlock, false>(a, i, b0, a.Bptr[i + 1] - b0, (uint32_t)(a.roff[i + 1] - a.roff[i]), m);
        __syncthreads();
    }
}

// ---- the fold ---------------------------------------------------------------------------------------------------
struct FoldArgs {
    uint32_t* ctl;
    const uint4* overflow;       // descriptors of the pairs with more than 16 final bins
    const uint64_t* flopptr;
    uint2* plist;
    const uint64_t* roff;
    const uint32_t* packed;
    bella_pair* tmp_pairs;
    bella_pair_ext* tmp_ext;
    uint16_t* sort_scratch;      // [F] (k_fold_overflow only)
    int k;
    int binSize;
};

struct Stride2Ptr {                  // word e of an interleaved {x,y} list: base[2e]
    uint32_t* base;
    __device__ __forceinline__ uint32_t& operator[](uint32_t e) const { return base[2u * e]; }
};

__device__ __forceinline__ void write_pair(const FoldArgs& a, const uint4 ds, const FoldResult& fr) {
    const uint32_t cid = ds.x, key = ds.y;
    const uint32_t k = (uint32_t)a.k;
    const uint32_t seedH = fr.seed & 0xFFFFu, seedV = fr.seed >> 16;
    const uint64_t leH = kmer_le(a.packed, a.roff[key] + seedH, k);
    const uint64_t leV = kmer_le(a.packed, a.roff[cid] + seedV, k);
    const uint32_t flags = (leH == leV ? 1u : 0u) | (kmer_rc_from_le(leH, k) == kmer_fw_from_le(leV, k) ? 2u : 0u);
    const uint64_t o = a.flopptr[cid] + ds.w;
    bella_pair pr;
    pr.rid = key; pr.cid = a.tmp_pairs[o].cid; pr.count = fr.count; pr.seedH = (uint16_t)seedH; pr.seedV = (uint16_t)seedV;   // (cid: the first product index, for k_order_*)
    pr.flags = (uint16_t)flags;
    a.tmp_pairs[o] = pr;
    if (a.tmp_ext) {
        bella_pair_ext ex;
        ex.nbins = fr.nbins; ex.support = fr.support; ex.binov = fr.binov; ex.pad = 0;
        a.tmp_ext[o] = ex;
    }
}

// pairs that end with more than 16 bins (choose()'s std::sort leaves the insertion-sort regime, common.h:145): the serial
// fold, in place in HBM on the pair's own list, with libstdc++'s introsort restated (core.hpp).  Rare by construction.
__global__ __launch_bounds__(64) void k_fold_overflow(FoldArgs a) {
    const uint32_t n = a.ctl[kCtlOverflow];
    for (;;) {
        const uint32_t idx = atomicAdd(&a.ctl[kCtlWork2], 1u);
        if (idx >= n) break;
        const uint4 ds = a.overflow[idx];
        const uint32_t mm = ds.z >> 16;
        const uint64_t lo = a.flopptr[ds.x] + (ds.z & 0xFFFFu);
        uint32_t* w = (uint32_t*)(a.plist + lo);
        FoldResult fr;
        fold_pair(Stride2Ptr{w}, Stride2Ptr{w + 1}, mm, a.k, a.binSize, a.sort_scratch ? a.sort_scratch + lo : (uint16_t*)nullptr, fr);
        if (fr.many_bins) atomicOr(&a.ctl[kCtlStatus], 1u);
        write_pair(a, ds, fr);
    }
}

// pairs/products of a sample of columns (a property of the operands, taken once at assembly): picks the key-table budget of
// the LDS tiers (cap/4 when at most ~1/5 of a column's products open a new pair, else cap/2).  One workgroup per sampled
// column, the distinct row ids counted with a bitmap in LDS.  out[0] = max over the sample of 1024*d/F.
__global__ __launch_bounds__(kBlock) void k_sample_pair_ratio(const uint32_t* Bptr, const uint2* Bent, const uint2* Aent, uint32_t nreads,
                                                              uint32_t first, uint32_t stride, uint32_t* out, uint32_t inl) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t* bits = (uint32_t*)smem;
    __shared__ uint32_t s_d, s_f;
    const uint32_t i = first + blockIdx.x * stride;              // (a column the context owns: B' holds its entries)
    if (i >= nreads) return;
    const uint32_t words = (nreads + 31) / 32;
    for (uint32_t w = threadIdx.x; w < words; w += kBlock) bits[w] = 0;
    if (threadIdx.x == 0) { s_d = 0; s_f = 0; }
    __syncthreads();
    uint32_t d = 0, f = 0;
    for (uint32_t e = Bptr[i] + threadIdx.x; e < Bptr[i + 1]; e += kBlock) {
        const uint2 be = Bent[e];
        const uint32_t cnt = bent_count(be, inl);
        f += cnt;
        for (uint32_t t = 0; t < cnt; ++t) {
            const uint32_t key = bent_partner(be, t, Aent, inl);
            const uint32_t bit = 1u << (key & 31);
            if (!(atomicOr(&bits[key >> 5], bit) & bit)) d++;
        }
    }
    atomicAdd(&s_d, d);
    atomicAdd(&s_f, f);
    __syncthreads();
    if (threadIdx.x == 0 && s_f >= 64) atomicMax(out, (uint32_t)(((uint64_t)s_d << 10) / s_f));
    // out[2], out[3]: pairs and products / 64 over the whole sample (the average list length: long-list inputs get row lists, bella_hip.hip)
    if (threadIdx.x == 0) { atomicAdd(out + 2, s_d); atomicAdd(out + 3, (s_f >> 6) < (1u << 20) ? (s_f >> 6) : (1u << 20)); }
}

// ---- the symbolic phase alone: estimateFLOP + estimateNNZ_Hash (overlap.hpp:157-202, 205-276) -------------------------------------
// Distinct row ids per output column, nothing else: no product lists, no fold, no records, no product-sized buffer.  The reference
// inserts a column's keys into a hash table sized by its products; here the table is a BITMAP over the reads (one bit per possible
// row id), in LDS when nreads bits fit (1.2 M reads) and in a per-workgroup slice of global memory above that: one atomic OR per
// product, the pair count is the number of bits that were newly set.  Persistent workgroups over the columns of the context
// (partition + stage); the bitmap is cleared after every column -- all of it when the column has more products than the bitmap has
// words, else only the words its products touched (the products are read again: they are in the caches).
constexpr int kCountBlock = 512;
struct CountArgs {
    const uint32_t* Bptr;        // row pointers of B' (the context's own: Bloc when partitioned)
    const uint2* Bent;
    const uint2* Aent;
    const uint2* Aent2;          // row lists (nullptr: expand B' x A')
    const uint64_t* Arow;
    uint32_t inl;                // B' entries in the INLINE form (util.hpp)
    uint32_t i0, stride, nown;
    uint32_t words;              // bitmap words = ceil(nreads / 32)
    uint32_t* gmap;              // global bitmaps, `words` per workgroup (nullptr: LDS)
    uint32_t* nnzC;              // [nreads + 1] out: pairs of the owned columns (the others are not touched)
    uint32_t* flops;             // [nreads + 1] out: products
    unsigned long long* totals;  // [2]: nnz(C), products (added)
};
template <bool LDSMAP>
__global__ __launch_bounds__(kCountBlock) void k_count_pairs(CountArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ uint32_t s_acc[2];
    uint32_t* bits = LDSMAP ? (uint32_t*)smem : a.gmap + (size_t)blockIdx.x * a.words;
    const uint32_t tid = threadIdx.x;
    for (uint32_t w = tid; w < a.words; w += kCountBlock) bits[w] = 0;
    if (tid < 2) s_acc[tid] = 0;
    __syncthreads();
    unsigned long long totP = 0, totF = 0;
    for (uint32_t j = blockIdx.x; j < a.nown; j += gridDim.x) {
        const uint32_t i = a.i0 + j * a.stride;
        uint32_t d = 0, f = 0;
        // one sweep over the column's products: mode 0 sets bits and counts, mode 1 clears the touched words
        auto sweep = [&](const int mode) {
            if (a.Aent2) {
                const uint64_t p0 = a.Arow[i], p1 = a.Arow[i + 1];
                for (uint64_t p = p0 + tid; p < p1; p += kCountBlock) {
                    const uint32_t key = a.Aent2[p].x & 0x3FFFFFFFu;
                    if (mode == 0) { const uint32_t bit = 1u << (key & 31u); if (!(atomicOr(&bits[key >> 5], bit) & bit)) d++; f++; }
                    else bits[key >> 5] = 0;
                }
            } else {
                for (uint32_t e = a.Bptr[i] + tid; e < a.Bptr[i + 1]; e += kCountBlock) {
                    const uint2 be = a.Bent[e];
                    const uint32_t cnt = bent_count(be, a.inl);
                    for (uint32_t t = 0; t < cnt; ++t) {
                        const uint32_t key = bent_partner(be, t, a.Aent, a.inl);
                        if (mode == 0) { const uint32_t bit = 1u << (key & 31u); if (!(atomicOr(&bits[key >> 5], bit) & bit)) d++; }
                        else bits[key >> 5] = 0;
                    }
                    if (mode == 0) f += cnt;
                }
            }
        };
        sweep(0);
#pragma unroll
        for (int dlt = 32; dlt > 0; dlt >>= 1) { d += __shfl_xor(d, dlt, 64); f += __shfl_xor(f, dlt, 64); }
        if (lane_id() == 0) { if (d) atomicAdd(&s_acc[0], d); if (f) atomicAdd(&s_acc[1], f); }
        __syncthreads();                                       // every bit of the column is set and counted
        const uint32_t dcol = s_acc[0], fcol = s_acc[1];
        if (fcol >= a.words) { for (uint32_t w = tid; w < a.words; w += kCountBlock) bits[w] = 0; }
        else sweep(1);
        __syncthreads();
        if (tid == 0) { a.nnzC[i] = dcol; a.flops[i] = fcol; totP += dcol; totF += fcol; s_acc[0] = 0; s_acc[1] = 0; }
        __syncthreads();
    }
    if (tid == 0) { if (totP) atomicAdd(a.totals, totP); if (totF) atomicAdd(a.totals + 1, totF); }
}

// estimateFLOP (overlap.hpp:157-202, lowtri): products of column i = sum of the suffix counts of its entries.
// One wavefront per column of this context (partition i % stride == first, stage [lo, hi)): column j is i0 + j * stride.  The
// entries of the other columns are zero: the host clears the arrays when the operands, the partition or the stage change.
// the same with row lists (assemble.hpp): the products of column i are Aent2[Arow[i] .. Arow[i + 1]) -- their number is a difference
// of two row pointers, no count stream to read.  One thread per column.
__global__ __launch_bounds__(kBlock) void k_row_flops_rl(const uint64_t* Arow, uint32_t i0, uint32_t stride, uint32_t nown, uint32_t* flops,
                                                         uint32_t* nnzC, uint32_t* ctl) {
    if (blockIdx.x == 0 && threadIdx.x < kCtlWords) ctl[threadIdx.x] = 0;   // the pass's control block (first kernel of the pass)
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nown) return;
    const uint32_t i = i0 + j * stride;
    nnzC[i] = 0;
    flops[i] = (uint32_t)(Arow[i + 1] - Arow[i]);
}
// The default layout keeps the products of every row (rowF, written once at layout time: assemble.hpp k_layout_live): one thread per column.
__global__ __launch_bounds__(kBlock) void k_row_flops(const uint32_t* rowF, uint32_t i0, uint32_t stride, uint32_t nown, uint32_t* flops,
                                                      uint32_t* nnzC, uint32_t* ctl) {
    if (blockIdx.x == 0 && threadIdx.x < kCtlWords) ctl[threadIdx.x] = 0;   // the pass's control block (first kernel of the pass)
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nown) return;
    const uint32_t i = i0 + j * stride;
    nnzC[i] = 0;                                               // this pass's pair counts start from zero
    flops[i] = rowF[i];
}

// tier lists: column i with f products goes to the first tier whose cap >= f (caps ascending; last tier = global path).
// One atomic per wavefront and tier; the order inside a list only affects scheduling.
__global__ __launch_bounds__(kBlock) void k_tier_lists(const uint32_t* flops, uint32_t nreads, uint32_t i0, uint32_t stride, uint32_t nown,
                                                       const uint32_t* caps, uint32_t ntiers,
                                                       const uint32_t* Bptr, const uint64_t* roff, uint4* desc, uint32_t* widelist,
                                                       uint32_t* counts, unsigned long long* total, uint64_t* obase, const uint64_t* Arow) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;      // the j-th column of this context
    const uint32_t i = i0 + j * stride;
    const uint32_t f = j < nown ? flops[i] : 0u;
    uint32_t tier = 0xFFFFFFFFu;
    uint4 ds = make_uint4(i, 0u, 0u, 0u);
    if (f > 0) {
        tier = ntiers;                                     // >= 65536 products: the wide path (wide.hpp), list number ntiers
        for (uint32_t t = 0; t < ntiers; ++t)
            if (f <= caps[t]) { tier = t; break; }
        const uint32_t b0 = Bptr[i];
        ds.y = b0;
        ds.z = (Bptr[i + 1] - b0) | ((uint32_t)(roff[i + 1] - roff[i]) << 16);   // both < 65536 (checked at set_reads / assembly)
        if (Arow) {                                        // row lists: first product (48 bits) instead of the entries, and the product count
            const uint64_t ar = Arow[i];
            ds.y = (uint32_t)ar;
            ds.z = (uint32_t)((ar >> 32) & 0xFFFFu) | ((uint32_t)(roff[i + 1] - roff[i]) << 16);
            ds.w = f;
        }
    }
    // One global atomic per tier and WORKGROUP (a few hot counters serve all columns): the wavefronts first reserve their places
    // inside the workgroup in LDS.
    __shared__ uint32_t s_cnt[16], s_base[16];
    if (threadIdx.x < 16) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    {   // temporary output space of the column (f slots; prefixsum of overlap.hpp:110-146 in spirit: any disjoint placement does,
        // the final order comes from colptrC): one atomic per wavefront on the pass's product total, which sizes the buffers
        unsigned long long inc = f;
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1) {
            const unsigned long long t = __shfl_up(inc, dlt, 64);
            if ((int)lane_id() >= dlt) inc += t;
        }
        const unsigned long long tot = __shfl(inc, 63, 64);
        unsigned long long base = 0;
        if (lane_id() == 0 && tot) base = atomicAdd(total, tot);
        base = __shfl(base, 0, 64);
        if (j < nown) obase[i] = base + inc - f;
        // the products of the columns above the LDS tiers (total[1]: <= 65,535 products each, the last tier) and of the wide ones
        // (total[2]): what the sort-based path of wide.hpp will hold -- the host sizes its batch from it without asking again
        if (__ballot(tier + 1 >= ntiers && tier != 0xFFFFFFFFu)) {
            unsigned long long m2 = tier + 1 == ntiers ? f : 0u, w2 = tier == ntiers ? f : 0u;
#pragma unroll
            for (int dlt = 32; dlt > 0; dlt >>= 1) { m2 += __shfl_xor(m2, dlt, 64); w2 += __shfl_xor(w2, dlt, 64); }
            if (lane_id() == 0) { if (m2) atomicAdd(total + 1, m2); if (w2) atomicAdd(total + 2, w2); }
        }
    }
    uint32_t my = 0;                                       // place inside the workgroup's share of my tier's list
    for (uint32_t t = 0; t <= ntiers; ++t) {
        const unsigned long long mask = __ballot(tier == t);
        if (mask == 0) continue;
        uint32_t base = 0;
        if (lane_id() == 0) base = atomicAdd(&s_cnt[t], (uint32_t)__popcll(mask));
        base = __shfl(base, 0, 64);
        if (tier == t) my = base + (uint32_t)__popcll(mask & ((1ull << lane_id()) - 1ull));
    }
    __syncthreads();
    if (threadIdx.x <= ntiers && s_cnt[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(&counts[threadIdx.x], s_cnt[threadIdx.x]);
    __syncthreads();
    if (tier != 0xFFFFFFFFu) {
        const uint32_t o = s_base[tier] + my;
        if (tier < ntiers) desc[(uint64_t)tier * nreads + o] = ds;
        else widelist[o] = i;
    }
}

// Init-time self-test of the one property phase S leans on that the ISA does not promise: the same-address LDS atomics of ONE
// wavefront instruction are applied in lane order (the value an atomicAdd returns is the number of lower lanes on the same address).
// Several address patterns, full and partial exec masks.  out[0] = number of lanes that saw anything else.  The phase-R check would
// catch a violation column by column (and send the column to the repairing path); this selects that path up front instead.
__global__ __launch_bounds__(64) void k_lane_order_selftest(uint32_t* out) {
    __shared__ uint32_t ctr[64];
    const uint32_t lane = threadIdx.x;
    uint32_t bad = 0;
    for (uint32_t pat = 0; pat < 12; ++pat) {
        ctr[lane] = 0;
        __syncthreads();
        const uint32_t groups = 1u << (pat % 6);                                  // 1, 2, 4, 8, 16, 32 distinct addresses
        const uint32_t addr = pat < 6 ? lane % groups : (lane * 7u + pat) % groups;
        const bool on = pat % 3 != 2 || (lane * 2654435761u >> 28) > 5;            // every third pattern under a partial exec mask
        uint32_t got = 0, want = 0;
        if (on) got = atomicAdd(&ctr[addr], 1u);
        for (uint32_t l = 0; l < 64; ++l) {
            const uint32_t a2 = (uint32_t)__shfl((int)addr, (int)l, 64);
            const uint32_t o2 = (uint32_t)__shfl((int)(on ? 1 : 0), (int)l, 64);
            if (l < lane && o2 && a2 == addr) ++want;
        }
        if (on && got != want) ++bad;
        __syncthreads();
    }
    if (bad) atomicAdd(out, bad);
}

// colptrC = exclusive prefix sums (u64) of the nreads + 1 pair counts: ONE launch of one 1024-thread workgroup for passes of up to
// kScanSingleMax columns (the counts go through LDS: coalesced reads, every thread then scans a contiguous chunk; their total is
// < 2^32 at this size).  Above, the library scan (two launches, many workgroups) takes over.
constexpr uint32_t kScanSingleMax = 12288;
__global__ __launch_bounds__(1024) void k_scan_counts(const uint32_t* in, uint64_t* out, uint32_t n) {
    __shared__ uint32_t s_v[kScanSingleMax];
    __shared__ uint32_t s_w[16];
    const uint32_t tid = threadIdx.x;
    for (uint32_t x = tid; x < n; x += 1024) s_v[x] = in[x] & ~kOrderedBit;
    __syncthreads();
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t lo = tid * per < n ? tid * per : n, hi = lo + per < n ? lo + per : n;
    uint32_t sum = 0;
    for (uint32_t x = lo; x < hi; ++x) sum += s_v[x];
    uint32_t tot;
    uint64_t run = block_excl_scan<16>(sum, s_w, &tot);
    for (uint32_t x = lo; x < hi; ++x) { out[x] = run; run += s_v[x]; }
}

}  // namespace bella
