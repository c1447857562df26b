// order.hpp -- the reference's output order inside a column, for many columns at a time (gfx950).
//
// LocalSpGEMM emits a column's pairs in the SLOT ORDER of its per-column open-addressing table (include/overlap.hpp:289-304
// table of 2^n >= max(16, nnz) slots, :321-333 hash key*107 with linear probing, :343-361 compaction in slot order).  That order is
// a function of (key, first product index) of the column's pairs alone: keys enter the table in the order of their first
// products.  The row kernels (spgemm.hpp) therefore leave their records in the order of THEIR grouping table with the first product
// index in the cid field, and the kernels here compute the slot order and move every record to its final place -- the pass over
// the records that used to be a plain compaction (the reference's "combine step", overlap.hpp:732-745).
//
// The order is reproduced with parallel atomicMin insertion of (first product << 16 | record index): an entry with an earlier
// first product displaces a later one, which resumes probing; the fixed point is exactly the layout sequential insertion produces
// (tests/test_core_host.py proves it under random interleavings).  The insertion cascades are chains of dependent LDS round trips:
// one WAVEFRONT per column (tables of up to 1,024 slots) keeps 24 independent columns in flight on a CU, which is what hides them;
// larger columns take a workgroup each.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/bella_hip.h"
#include "core.hpp"
#include "slotorder.hpp"
#include "util.hpp"

namespace bella {

constexpr uint32_t kOrderWaveHt = 1024;                    // largest table a single wavefront orders (4 KB + 2 KB of LDS)
constexpr uint32_t kOrderLdsHt = 16384;                    // largest table a workgroup holds in LDS (64 KB + 32 KB)
constexpr uint32_t kOrderBlock = 256;
constexpr uint32_t kOrderBigGrid = 512;                    // persistent workgroups of k_order_block
constexpr uint32_t kOrderOneLaunchMax = 32768;             // columns of a pass up to which ONE wavefront-per-column launch takes all table sizes
constexpr uint64_t kOrderWsBytes = (uint64_t)6 * 65536;    // global tables of one workgroup for columns with more than 16,384 pairs

struct OrderArgs {
    const uint64_t* flopptr;     // where a column's records stand in tmp_pairs
    const uint64_t* colptrC;     // ... and where they go
    const uint32_t* nnzC;        // pairs per column (| kOrderedBit: already in slot order, cid already the column)
    const uint32_t* flops;       // products per column: the first product indices (insertion times) lie below it
    uint32_t nreads, i0, stride, nown;
    const bella_pair* tmp_pairs;
    const bella_pair_ext* tmp_ext;
    bella_pair* pairs;
    bella_pair_ext* ext;
    uint64_t* totals;            // totals[0] = nnz(C)
    uint32_t* nbig;              // columns left to k_order_block
    uint32_t* biglist;
    uint8_t* ws;                 // kOrderWsBytes per workgroup of k_order_block
};

__device__ __forceinline__ void order_copy_record(const OrderArgs& a, uint64_t from, uint64_t to, uint32_t cid) {
    uint4 rec = *(const uint4*)(a.tmp_pairs + from);
    asm volatile("" : "+v"(rec.y));                          // (keeps the load ONE 16-byte load: the compiler splits it around the word that is replaced)
    rec.y = cid;
    *(uint4*)(a.pairs + to) = rec;
    if (a.ext) a.ext[to] = a.tmp_ext[from];
}

// The two table scans of a wavefront-ordered column, lane-INTERLEAVED (lane l takes slots 64 b + l): a block of 64 slots is one
// conflict-free LDS access and one ballot, where the chunked form of slotorder.hpp (lane l takes ht / 64 consecutive slots) puts all
// lanes of a 32-lane group on two of the 32 banks (stride 16 words at ht = 1,024).  (Round 5: the kernel's time did not change -- 0.597 ms
// for the 50 M records of the 100k set either way: it moves 2.4 GB at 4 TB/s, the copy ceiling of these boxes being 5 TB/s.  Measured and
// not kept: the records read ONCE, whole, into registers and scattered to their ranks with 16-byte stores -- 1.6 GB instead of 2.4 GB,
// but 0.67 ms: scattered 16-byte stores cost more than scattered 16-byte loads.)
// nf[s] = first empty slot at or after s, cyclically; false: the table is full
__device__ __forceinline__ bool wave_next_free(const uint32_t* T2, uint16_t* nf, uint32_t ht) {
    const uint32_t lane = lane_id();
    const uint32_t nb = (ht + 63u) >> 6;
    uint32_t carry = 0xFFFFFFFFu;                             // first empty slot above the current block; after the first sweep: of the whole table
#pragma unroll 1
    for (int sweep = 0; sweep < 2; ++sweep) {
        for (uint32_t b = nb; b-- > 0;) {
            const uint32_t s = (b << 6) + lane;
            const bool empty = s < ht && T2[s] == kEmpty;
            const unsigned long long m = __ballot(empty);
            if (sweep == 1 && s < ht) {
                const unsigned long long up = m >> lane;      // empty slots of this block at or after mine
                nf[s] = (uint16_t)(up ? s + (uint32_t)__ffsll((long long)up) - 1u : carry);
            }
            if (m) carry = (b << 6) + (uint32_t)__ffsll((long long)m) - 1u;
        }
        if (carry == 0xFFFFFFFFu) return false;
    }
    return true;
}
// ord[rank] = record index of the rank-th occupied slot (the slot order itself)
__device__ __forceinline__ void wave_rank_slots(const uint32_t* T2, uint16_t* ord, uint32_t ht) {
    const uint32_t lane = lane_id();
    const uint32_t nb = (ht + 63u) >> 6;
    uint32_t base = 0;
    for (uint32_t b = 0; b < nb; ++b) {
        const uint32_t s = (b << 6) + lane;
        const uint32_t it = s < ht ? T2[s] : kEmpty;
        const unsigned long long m = __ballot(it != kEmpty);
        if (it != kEmpty) ord[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)(it & 0xFFFFu);
        base += (uint32_t)__popcll(m);
    }
}

// one wavefront per column: tables of HTLO < ht <= HT slots (HT <= kOrderWaveHt; two instances share the columns: the small one keeps
// fewer records per lane in registers and half the LDS, so more of its columns are in flight per CU).  The instance with HTLO == 0
// also copies the columns that are in order already and lists the ones above kOrderWaveHt for k_order_block.
template <uint32_t HT, uint32_t HTLO>
__global__ __launch_bounds__(kOrderBlock) void k_order_wave(OrderArgs a) {
    __shared__ uint32_t s_T2[kOrderBlock / 64][HT];
    __shared__ uint16_t s_ord[kOrderBlock / 64][HT];
    const uint32_t j0 = blockIdx.x * (kOrderBlock / 64) + wave_id();
    const uint32_t lane = lane_id();
    if (HTLO == 0 && j0 == 0 && lane == 0) a.totals[0] = a.colptrC[a.nreads];   // nnz(C): read back once with the control block
    if (j0 >= a.nown) return;
    const uint32_t i = a.i0 + j0 * a.stride;
    // everything the column needs from its descriptors in ONE round trip (each of these is wave-uniform, so the compiler waits for
    // it where it is loaded: loaded one after the other they are four dependent trips to HBM per column)
    uint32_t nz = a.nnzC[i];
    uint64_t src = a.flopptr[i], dst = a.colptrC[i];
    uint32_t tmax = a.flops[i];
    asm volatile("" : "+v"(nz), "+v"(src), "+v"(dst), "+v"(tmax));   // (all four are issued before the first is waited for)
    const uint32_t d = nz & ~kOrderedBit;
    if (!d) return;
    if (nz & kOrderedBit) {
        if (HTLO == 0) for (uint32_t r = lane; r < d; r += 64) order_copy_record(a, src + r, dst + r, i);
        return;
    }
    const uint32_t ht = pow2_at_least(16u, d);
    if (ht > kOrderWaveHt) {
        if (HTLO == 0 && lane == 0) a.biglist[atomicAdd(a.nbig, 1u)] = i;
        return;
    }
    if (ht > HT || ht <= HTLO) return;                       // the other instance's column
    uint32_t* T2 = s_T2[wave_id()];
    uint16_t* ord = s_ord[wave_id()];                        // next-free table during the rounds, then rank -> record
    for (uint32_t s = lane; s < ht; s += 64) T2[s] = kEmpty;
    constexpr uint32_t NI = HT / 64;                         // records per lane
    uint32_t key[NI], fp[NI];
#pragma unroll
    for (uint32_t u = 0; u < NI; ++u) {
        const uint32_t j = lane + 64 * u;
        key[u] = 0; fp[u] = 0xFFFFFFFFu;
        if (j < d) { const uint2 kf = *(const uint2*)(a.tmp_pairs + src + j); key[u] = kf.x; fp[u] = kf.y; }   // {key, first product}
    }
    group_sync<64>();
    uint32_t prev = 0;
    for (uint32_t rd = 0;; ++rd) {                            // rounds by insertion time (slotorder.hpp)
        const uint32_t bound = round_bound(rd, ht, d, tmax);
        if (rd) { (void)wave_next_free(T2, ord, ht); group_sync<64>(); }
#pragma unroll
        for (uint32_t u = 0; u < NI; ++u) {
            if (fp[u] < prev || fp[u] >= bound) continue;
            const uint32_t home = (key[u] * 107u) & (ht - 1), item = (fp[u] << 16) | (lane + 64 * u);
            if (rd) slot_insert<true>(T2, ord, ht - 1, home, item);
            else slot_insert<false>(T2, ord, ht - 1, home, item);
        }
        group_sync<64>();
        if (bound >= tmax) break;
        prev = bound;
    }
    wave_rank_slots(T2, ord, ht);
    group_sync<64>();
    {   // four records in flight per lane
        uint32_t r = lane;
        for (; r + 192 < d; r += 256) {
            uint4 rec[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) rec[u] = *(const uint4*)(a.tmp_pairs + src + ord[r + 64 * u]);
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) { asm volatile("" : "+v"(rec[u].y)); rec[u].y = i; *(uint4*)(a.pairs + dst + r + 64 * u) = rec[u]; }
            if (a.ext) {
#pragma unroll
                for (uint32_t u = 0; u < 4; ++u) a.ext[dst + r + 64 * u] = a.tmp_ext[src + ord[r + 64 * u]];
            }
        }
        for (; r < d; r += 64) order_copy_record(a, src + ord[r], dst + r, i);
    }
}

// one workgroup per column (persistent): the columns k_order_wave listed
__global__ __launch_bounds__(kOrderBlock) void k_order_block(OrderArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ uint32_t scr[16];
    const uint32_t nbig = *a.nbig;
    const uint32_t tid = threadIdx.x;
    for (uint32_t x = blockIdx.x; x < nbig; x += gridDim.x) {
        const uint32_t i = a.biglist[x];
        const uint32_t d = a.nnzC[i] & ~kOrderedBit;
        const uint64_t src = a.flopptr[i], dst = a.colptrC[i];
        const uint32_t ht = pow2_at_least(16u, d);
        uint32_t* T2;
        uint16_t* ord;
        if (ht <= kOrderLdsHt) { T2 = (uint32_t*)smem; ord = (uint16_t*)(smem + (size_t)4 * kOrderLdsHt); }
        else { uint8_t* w = a.ws + (uint64_t)blockIdx.x * kOrderWsBytes; T2 = (uint32_t*)w; ord = (uint16_t*)(w + (size_t)4 * 65536); }
        for (uint32_t s = tid; s < ht; s += kOrderBlock) T2[s] = kEmpty;
        __syncthreads();
        const uint32_t tmax = a.flops[i];
        uint32_t prev = 0;
        for (uint32_t rd = 0;; ++rd) {                        // rounds by insertion time (slotorder.hpp); ord doubles as the next-free table
            const uint32_t bound = round_bound(rd, ht, d, tmax);
            if (rd) { (void)build_next_free<(int)kOrderBlock>(T2, ord, ht, scr); __syncthreads(); }
            for (uint32_t j = tid; j < d; j += kOrderBlock) {
                const uint2 kf = *(const uint2*)(a.tmp_pairs + src + j);
                if (kf.y < prev || kf.y >= bound) continue;
                const uint32_t home = (kf.x * 107u) & (ht - 1), item = (kf.y << 16) | j;
                if (rd) slot_insert<true>(T2, ord, ht - 1, home, item);
                else slot_insert<false>(T2, ord, ht - 1, home, item);
            }
            __syncthreads();
            if (bound >= tmax) break;
            prev = bound;
        }
        {
            const uint32_t c = (ht + kOrderBlock - 1) / kOrderBlock;
            const uint32_t lo = tid * c < ht ? tid * c : ht, hi = lo + c < ht ? lo + c : ht;
            uint32_t occ = 0;
            for (uint32_t s = lo; s < hi; ++s) occ += T2[s] != kEmpty ? 1u : 0u;
            uint32_t tot;
            uint32_t rank = block_excl_scan<kOrderBlock / 64>(occ, scr, &tot);
            for (uint32_t s = lo; s < hi; ++s) {
                const uint32_t it = T2[s];
                if (it != kEmpty) ord[rank++] = (uint16_t)(it & 0xFFFFu);
            }
        }
        __syncthreads();
        for (uint32_t r = tid; r < d; r += kOrderBlock) order_copy_record(a, src + ord[r], dst + r, i);
        __syncthreads();
    }
}

}  // namespace bella
