// spgemm.hpp -- C = A*A^T (strict lower triangle) under BELLA's position-binning semiring, gfx950.
//
// Replaces estimateFLOP (include/overlap.hpp:157-202), estimateNNZ_Hash (:205-276) and LocalSpGEMM
// (:281-363) with multiop/chainop (include/chain.hpp:74-150) of the reference.  One workgroup
// (4 wavefronts) owns one output column i (= read i) and runs five bulk-synchronous phases over LDS
// (or over a global workspace for columns whose product list does not fit the LDS tiers):
//
//   X  expand   stream the column's B' entries (8 B each, coalesced); every entry carries a direct
//               pointer to the suffix "reads > i" of its k-mer's list in A' (8 B entries, contiguous),
//               so there is no column-pointer indirection.  Products are written to LDS in the
//               reference's order (B slot order, then ascending read id) together with their u16
//               overlap estimate, and their key (row id) is inserted into a grouping hash table T1.
//   O  order    emulate the reference's per-column open-addressing table (size 2^n >= max(16,nnz),
//               hash key*107, linear probe) to obtain its SLOT ORDER: keys are inserted in parallel with
//               atomicMin on (first-product-index, key): an entry with an earlier first occurrence
//               displaces a later one, which resumes probing -- the fixed point is exactly the layout
//               sequential insertion produces.  A scan over the table gives every key its output rank.
//   S  scatter  counting-sort the products by key into contiguous per-pair lists (rank order).
//   F  fold     one lane per pair: restore product order inside the list, then fold it in place with
//               core.hpp's fold_pair (order-dependent semiring) and choose() the seed.
//   W  write    16-byte pair records at a per-column temporary offset (exclusive scan of flops);
//               k_compact_pairs packs them once nnz(C) per column is known.
//
// Data layout in HBM (built by assemble.hpp):
//   Bent[e] = { a_ptr, posV | cnt << 16 | ori << 31 }   e in B' order (column i, MergeDuplicates slot order)
//   Aent[x] = { read | ori << 31, posH | readlen << 16 } k-mer lists, ascending read id, stored in order of
//                                                          first appearance in B' (streaming for the owner row)
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/bella_hip.h"
#include "core.hpp"
#include "util.hpp"

namespace bella {

struct SpgemmArgs {
    const uint32_t* rowlist;
    uint32_t nrows;
    const uint32_t* Bptr;
    const uint2* Bent;
    const uint2* Aent;
    const uint64_t* roff;
    const uint32_t* packed;
    const uint64_t* flopptr;
    bella_pair* tmp_pairs;
    bella_pair_ext* tmp_ext;
    uint32_t* nnzC;
    uint16_t* sort_scratch;
    uint32_t* status;
    uint8_t* ws;
    uint64_t ws_stride;
    uint32_t cap;
    int k;
    int binSize;
    unsigned long long* phase;   // optional per-phase cycle counters (development aid, BELLA_HIP_PHASE_TIMERS=1)
};

constexpr uint32_t kRowScratchBytes = 64;                    // block scan scratch + counters
__host__ __device__ inline size_t row_mem_bytes(uint32_t cap) { return kRowScratchBytes + (size_t)30 * cap; }

struct RowMem {
    uint32_t* scr;      // 16 words
    uint32_t* A_hv;     // [cap]  posH | posV << 16, product order
    uint32_t* A_gov;    // [cap]  T1 slot << 16 | overlap estimate
    uint32_t* S_hv;     // [cap]  grouped by pair           (phase O: T2 overlays S_hv..S_pov, 2*cap words)
    uint32_t* S_pov;    // [cap]  product index << 16 | overlap estimate ; fold: bin metadata
    uint32_t* T1key;    // [cap]
    uint32_t* T1first;  // [cap]  first product index ; after phase O: list start | rank << 16
    uint32_t* T1cnt;    // [cap]  products | scatter cursor << 16
    uint16_t* G;        // [cap]  rank -> T1 slot
    uint32_t cap;
};

__device__ __forceinline__ RowMem carve(uint8_t* base, uint32_t cap) {
    RowMem m;
    m.scr = (uint32_t*)base;
    uint32_t* w = (uint32_t*)(base + kRowScratchBytes);
    m.A_hv = w;            w += cap;
    m.A_gov = w;           w += cap;
    m.S_hv = w;            w += cap;
    m.S_pov = w;           w += cap;
    m.T1key = w;           w += cap;
    m.T1first = w;         w += cap;
    m.T1cnt = w;           w += cap;
    m.G = (uint16_t*)w;
    m.cap = cap;
    return m;
}

__device__ __forceinline__ void process_row(const SpgemmArgs& a, const uint32_t i, const RowMem& m) {
    const uint32_t tid = threadIdx.x;
    const uint32_t H1 = m.cap;
    uint32_t* s_d = m.scr + 8;
    const uint32_t b0 = a.Bptr[i];
    const uint32_t n = a.Bptr[i + 1] - b0;
    const uint32_t lenV = (uint32_t)(a.roff[i + 1] - a.roff[i]);
    const uint32_t k = (uint32_t)a.k;

    for (uint32_t s = tid; s < H1; s += kBlock) { m.T1key[s] = kEmpty; m.T1first[s] = kEmpty; m.T1cnt[s] = 0; }
    if (tid == 0) *s_d = 0;
    __syncthreads();
    long long tc = 0;
    if (a.phase && tid == 0) tc = clock64();
#define BELLA_PHASE(n) if (a.phase && tid == 0) { const long long t2 = clock64(); atomicAdd(a.phase + (n), (unsigned long long)(t2 - tc)); tc = t2; }

    // ---- X: expand products in reference order, group keys --------------------------------------------
    uint32_t running = 0;
    for (uint32_t base = 0; base < n; base += kBlock) {
        const uint32_t j = base + tid;
        uint2 be = make_uint2(0u, 0u);
        if (j < n) be = a.Bent[b0 + j];
        const uint32_t cnt = (be.y >> 16) & 0x7FFFu;
        uint32_t tot;
        const uint32_t off = running + block_excl_scan(cnt, m.scr, &tot);
        const uint32_t oriB = be.y >> 31, posV = be.y & 0xFFFFu;
        for (uint32_t t = 0; t < cnt; ++t) {
            const uint2 ae = a.Aent[(uint64_t)be.x + t];
            const uint32_t key = ae.x & 0x7FFFFFFFu;
            const uint32_t posH = ae.y & 0xFFFFu, lenH = ae.y >> 16;
            const uint32_t ov = (uint32_t)overlap_estimate(posH, posV, lenH, lenV, (ae.x >> 31) == oriB, k) & 0xFFFFu;
            const uint32_t p = off + t;
            uint32_t h = hash_range(key, H1);
            uint32_t old;
            for (;;) {
                old = atomicCAS(&m.T1key[h], kEmpty, key);
                if (old == kEmpty || old == key) break;
                h = (h + 1 == H1) ? 0 : h + 1;
            }
            if (old == kEmpty) atomicAdd(s_d, 1u);
            atomicMin(&m.T1first[h], p);
            atomicAdd(&m.T1cnt[h], 1u);
            m.A_hv[p] = posH | (posV << 16);
            m.A_gov[p] = (h << 16) | ov;
        }
        running += tot;
    }
    const uint32_t F = running;
    __syncthreads();
    const uint32_t d = *s_d;
    BELLA_PHASE(0)

    // ---- O: the reference's slot order (overlap.hpp:289-361) -----------------------------------------
    const uint32_t ht = pow2_at_least(16u, d);
    uint32_t* T2 = m.S_hv;
    for (uint32_t s = tid; s < ht; s += kBlock) T2[s] = kEmpty;
    __syncthreads();
    for (uint32_t s = tid; s < H1; s += kBlock) {
        const uint32_t key = m.T1key[s];
        if (key == kEmpty) continue;
        uint32_t item = (m.T1first[s] << 16) | s;
        uint32_t h = (key * 107u) & (ht - 1);
        for (;;) {
            const uint32_t old = atomicMin(&T2[h], item);
            if (old == kEmpty) break;
            if (old > item) item = old;          // we took the slot; the displaced entry resumes probing
            h = (h + 1) & (ht - 1);
        }
    }
    __syncthreads();
    {
        const uint32_t c = (ht + kBlock - 1) / kBlock;
        const uint32_t lo = tid * c;
        const uint32_t hi = lo + c < ht ? lo + c : ht;
        uint32_t occ = 0, csum = 0;
        for (uint32_t s = lo; s < hi; ++s) {
            const uint32_t it = T2[s];
            if (it != kEmpty) { occ++; csum += m.T1cnt[it & 0xFFFFu] & 0xFFFFu; }
        }
        uint32_t tot;
        const uint32_t ex = block_excl_scan((occ << 16) | csum, m.scr, &tot);
        uint32_t rank = ex >> 16, st = ex & 0xFFFFu;
        for (uint32_t s = lo; s < hi; ++s) {
            const uint32_t it = T2[s];
            if (it == kEmpty) continue;
            const uint32_t g = it & 0xFFFFu;
            m.G[rank] = (uint16_t)g;
            m.T1first[g] = st | (rank << 16);
            st += m.T1cnt[g] & 0xFFFFu;
            rank++;
        }
    }
    __syncthreads();
    BELLA_PHASE(1)

    // ---- S: scatter products into per-pair lists -----------------------------------------------------
    for (uint32_t p = tid; p < F; p += kBlock) {
        const uint32_t gov = m.A_gov[p];
        const uint32_t g = gov >> 16;
        const uint32_t old = atomicAdd(&m.T1cnt[g], 0x10000u);
        const uint32_t slot = (m.T1first[g] & 0xFFFFu) + (old >> 16);
        m.S_hv[slot] = m.A_hv[p];
        m.S_pov[slot] = (p << 16) | (gov & 0xFFFFu);
    }
    __syncthreads();
    BELLA_PHASE(2)

    // ---- F/W: fold each pair, write its record -------------------------------------------------------
    const uint64_t obase = a.flopptr[i];
    const uint64_t goffV = a.roff[i];
    for (uint32_t r = tid; r < d; r += kBlock) {
        const uint32_t g = m.G[r];
        const uint32_t key = m.T1key[g];
        const uint32_t st = m.T1first[g] & 0xFFFFu;
        const uint32_t mm = m.T1cnt[g] & 0xFFFFu;
        uint32_t* P = m.S_hv + st;
        uint32_t* Bm = m.S_pov + st;
        FoldResult fr;
        if (mm == 1) {
            fr.count = 1; fr.nbins = 1; fr.support = 1; fr.binov = (uint16_t)(Bm[0] & 0xFFFFu); fr.seed = P[0]; fr.many_bins = 0;
        } else {
            sort_products_by_index(P, Bm, mm);
            fold_pair(P, Bm, mm, a.k, a.binSize, a.sort_scratch ? a.sort_scratch + obase + st : (uint16_t*)nullptr, fr);
        }
        if (fr.many_bins) atomicOr(a.status, 1u);
        const uint32_t seedH = fr.seed & 0xFFFFu, seedV = fr.seed >> 16;
        const uint64_t leH = kmer_le(a.packed, a.roff[key] + seedH, k);
        const uint64_t leV = kmer_le(a.packed, goffV + seedV, k);
        const uint32_t flags = (leH == leV ? 1u : 0u) | (kmer_rc_from_le(leH, k) == kmer_fw_from_le(leV, k) ? 2u : 0u);
        bella_pair pr;
        pr.rid = key; pr.cid = i; pr.count = fr.count; pr.seedH = (uint16_t)seedH; pr.seedV = (uint16_t)seedV;
        pr.flags = (uint16_t)flags;
        a.tmp_pairs[obase + r] = pr;
        if (a.tmp_ext) {
            bella_pair_ext ex;
            ex.nbins = fr.nbins; ex.support = fr.support; ex.binov = fr.binov; ex.pad = 0;
            a.tmp_ext[obase + r] = ex;
        }
    }
    if (tid == 0) a.nnzC[i] = d;
    if (a.phase) { __syncthreads(); BELLA_PHASE(3) }
#undef BELLA_PHASE
}

// LDS tiers: one column per workgroup, dynamic LDS = row_mem_bytes(cap)
__global__ __launch_bounds__(kBlock) void k_spgemm_rows_lds(SpgemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t i = a.rowlist[blockIdx.x];
    const RowMem m = carve(smem, a.cap);
    process_row(a, i, m);
}

// Global-workspace path: columns with cap < products < 65536; persistent workgroups, one workspace each.
__global__ __launch_bounds__(kBlock) void k_spgemm_rows_global(SpgemmArgs a) {
    uint8_t* ws = a.ws + (uint64_t)blockIdx.x * a.ws_stride;
    for (uint32_t x = blockIdx.x; x < a.nrows; x += gridDim.x) {
        const uint32_t i = a.rowlist[x];
        const uint32_t f = (uint32_t)(a.flopptr[i + 1] - a.flopptr[i]);
        const RowMem m = carve(ws, f < 16u ? 16u : f);   // T2 (>= 16 slots) overlays 2*cap words
        process_row(a, i, m);
        __syncthreads();
    }
}

// estimateFLOP (overlap.hpp:157-202, lowtri): products of column i = sum of the suffix counts of its entries.
// One wavefront per column.  Columns outside this context's partition get 0.
__global__ __launch_bounds__(kBlock) void k_row_flops(const uint32_t* Bptr, const uint2* Bent, uint32_t nreads,
                                                      uint32_t first, uint32_t stride, uint32_t* flops) {
    const uint32_t i = blockIdx.x * kWaves + wave_id();
    if (i >= nreads) return;
    uint32_t s = 0;
    if (i % stride == first) {
        const uint32_t b0 = Bptr[i], b1 = Bptr[i + 1];
        for (uint32_t e = b0 + lane_id(); e < b1; e += 64) s += (Bent[e].y >> 16) & 0x7FFFu;
    }
#pragma unroll
    for (int dlt = 32; dlt > 0; dlt >>= 1) s += __shfl_xor(s, dlt, 64);
    if (lane_id() == 0) flops[i] = s;
}

// tier flags: flag[t*nreads + i] = 1 iff column i belongs to tier t (caps ascending; last tier = global path)
__global__ void k_tier_flags(const uint32_t* flops, uint32_t nreads, const uint32_t* caps, uint32_t ntiers,
                             uint8_t* flag, uint32_t* status) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nreads) return;
    const uint32_t f = flops[i];
    uint32_t tier = 0xFFFFFFFFu;
    if (f > 0) {
        for (uint32_t t = 0; t < ntiers; ++t)
            if (f <= caps[t]) { tier = t; break; }
        if (tier == 0xFFFFFFFFu) atomicOr(status, 2u);     // >= 65536 products: wide path missing
    }
    for (uint32_t t = 0; t < ntiers; ++t) flag[(uint64_t)t * nreads + i] = (tier == t);
}

__global__ __launch_bounds__(kBlock) void k_compact_pairs(const uint64_t* flopptr, const uint64_t* colptrC,
                                                          const uint32_t* nnzC, uint32_t nreads,
                                                          const bella_pair* tmp_pairs, const bella_pair_ext* tmp_ext,
                                                          bella_pair* pairs, bella_pair_ext* ext) {
    const uint32_t i = blockIdx.x * kWaves + wave_id();
    if (i >= nreads) return;
    const uint32_t cnt = nnzC[i];
    const uint64_t src = flopptr[i], dst = colptrC[i];
    for (uint32_t r = lane_id(); r < cnt; r += 64) {
        pairs[dst + r] = tmp_pairs[src + r];
        if (ext) ext[dst + r] = tmp_ext[src + r];
    }
}

}  // namespace bella
