// assemble.hpp -- on-device assembly of the operands (gfx950).
//
// Replaces, for this path, what src/main.cpp:476-489 does on the host of the reference:
//   CSC<..> transpmat(tuples, ...)   src/CSC.cpp:422-479 (stable bucket by read) + MergeDuplicates :301-420
//   spmat = transpmat.Transpose()    include/common/transpose.h:13-52
// and adds what the SpGEMM kernel wants precomputed per nonzero: the orientation bit of the k-mer
// occurrence (checkstrand, chain.hpp:35-44, becomes a 1-bit compare), the read length, and a direct
// pointer from every B' entry to the "reads > i" suffix of its k-mer list.
#pragma once
#include <hip/hip_runtime.h>
#include "core.hpp"
#include "util.hpp"

namespace bella {

// ASCII ACGT -> 2 bit, 16 bases per word (base t of the concatenation at bits 2(t%16) of word t/16)
__global__ void k_pack_reads(const uint8_t* bases, uint64_t total, uint32_t* packed, uint64_t nwords, uint32_t* status) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwords) return;
    uint32_t out = 0;
    bool bad = false;
    for (uint32_t t = 0; t < 16; ++t) {
        const uint64_t g = w * 16 + t;
        if (g >= total) break;
        const uint32_t c = bases[g];
        uint32_t code;
        switch (c) {
            case 'A': code = 0; break;
            case 'C': code = 1; break;
            case 'G': code = 2; break;
            case 'T': code = 3; break;
            default: code = 0; bad = true;
        }
        out |= code << (2 * t);
    }
    packed[w] = out;
    if (bad) atomicOr(status, 4u);
}

// tstart[r] = first tuple of read r (tuples grouped by non-decreasing read id); tstart[nreads] = ntuples
// (reads are numbered from `first`: a panel of the matrix holds the rows first .. first+nreads-1)
__global__ void k_tuple_bounds(const uint32_t* t_read, uint64_t ntuples, uint32_t first, uint32_t nreads, uint64_t* tstart,
                               uint32_t* status) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > ntuples) return;
    const int64_t prev = t == 0 ? -1 : (int64_t)t_read[t - 1] - (int64_t)first;
    const int64_t cur = t == ntuples ? (int64_t)nreads : (int64_t)t_read[t] - (int64_t)first;
    if (cur < prev || prev < -1 || (t < ntuples && cur >= (int64_t)nreads)) { atomicOr(status, 8u); return; }
    for (int64_t r = prev + 1; r <= cur; ++r) tstart[r] = t;
}

struct AsmArgs {
    const uint32_t* t_kmer;
    const uint16_t* t_pos;
    const uint64_t* tstart;
    uint32_t nreads;
    uint32_t* Bk_tmp;       // [ntuples] row r written at tstart[r]
    uint16_t* Bpos_tmp;
    uint32_t* rowcnt;       // distinct k-mers per read
    uint8_t* ws;            // global tables for reads with more than kAsmLdsSlots tuples
    uint64_t ws_stride;
    uint32_t* status;
};
constexpr uint32_t kAsmLdsSlots = 4096;
constexpr size_t kAsmLdsBytes = 64 + (size_t)16 * kAsmLdsSlots;

// One read: de-duplicate (keep FIRST occurrence index for the slot order, LAST occurrence's position as the
// value: CSC.cpp:344 with main.cpp:477-480's lambda), then emulate the insertion order into the reference's
// table of size ht = 2^n >= max(16, #tuples) (CSC.cpp:322-326) with the atomicMin displacement scheme
// (see spgemm.hpp phase O), and emit the slots in order (CSC.cpp:358-373).
__device__ __forceinline__ void asm_row(const AsmArgs& a, uint32_t r, uint32_t* scr, uint32_t* K, uint32_t* first,
                                        uint32_t* last, uint32_t* T2, uint32_t ht) {
    const uint32_t tid = threadIdx.x;
    const uint64_t ts = a.tstart[r];
    const uint32_t n = (uint32_t)(a.tstart[r + 1] - ts);
    uint32_t* s_d = scr + 8;
    for (uint32_t s = tid; s < ht; s += kBlock) { K[s] = kEmpty; first[s] = kEmpty; last[s] = 0; T2[s] = kEmpty; }
    if (tid == 0) *s_d = 0;
    __syncthreads();
    for (uint32_t t = tid; t < n; t += kBlock) {
        const uint32_t key = a.t_kmer[ts + t];
        uint32_t h = (key * 107u) & (ht - 1);
        uint32_t old;
        for (;;) {
            old = atomicCAS(&K[h], kEmpty, key);
            if (old == kEmpty || old == key) break;
            h = (h + 1) & (ht - 1);
        }
        if (old == kEmpty) atomicAdd(s_d, 1u);
        atomicMin(&first[h], t);
        atomicMax(&last[h], t);
    }
    __syncthreads();
    const uint32_t d = *s_d;
    for (uint32_t s = tid; s < ht; s += kBlock) {
        const uint32_t key = K[s];
        if (key == kEmpty) continue;
        uint32_t item = (first[s] << 16) | s;
        uint32_t h = (key * 107u) & (ht - 1);
        for (;;) {
            const uint32_t old = atomicMin(&T2[h], item);
            if (old == kEmpty) break;
            if (old > item) item = old;
            h = (h + 1) & (ht - 1);
        }
    }
    __syncthreads();
    const uint32_t c = (ht + kBlock - 1) / kBlock;
    const uint32_t lo = tid * c;
    const uint32_t hi = lo + c < ht ? lo + c : ht;
    uint32_t occ = 0;
    for (uint32_t s = lo; s < hi; ++s) occ += (T2[s] != kEmpty);
    uint32_t tot;
    uint32_t rank = block_excl_scan(occ, scr, &tot);
    for (uint32_t s = lo; s < hi; ++s) {
        const uint32_t it = T2[s];
        if (it == kEmpty) continue;
        const uint32_t g = it & 0xFFFFu;
        a.Bk_tmp[ts + rank] = K[g];
        a.Bpos_tmp[ts + rank] = a.t_pos[ts + last[g]];
        rank++;
    }
    if (tid == 0) a.rowcnt[r] = d;
    __syncthreads();
}

__global__ __launch_bounds__(kBlock) void k_asm_rows(AsmArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t* scr = (uint32_t*)smem;
    uint32_t* L = (uint32_t*)(smem + 64);
    uint32_t* W = (uint32_t*)(a.ws + (uint64_t)blockIdx.x * a.ws_stride);
    for (uint32_t r = blockIdx.x; r < a.nreads; r += gridDim.x) {
        const uint32_t n = (uint32_t)(a.tstart[r + 1] - a.tstart[r]);
        if (n == 0) { if (threadIdx.x == 0) a.rowcnt[r] = 0; continue; }
        if (n >= 65536u) { if (threadIdx.x == 0) { atomicOr(a.status, 16u); a.rowcnt[r] = 0; } continue; }
        const uint32_t ht = pow2_at_least(16u, n);
        if (ht <= kAsmLdsSlots) asm_row(a, r, scr, L, L + ht, L + 2 * ht, L + 3 * ht, ht);
        else asm_row(a, r, scr, W, W + ht, W + 2 * ht, W + 3 * ht, ht);
    }
}

// pack the per-read rows (written at tuple offsets) to the final CSR of B; also the row id of every entry
__global__ __launch_bounds__(kBlock) void k_compact_B(const uint64_t* tstart, const uint32_t* Bptr, uint32_t nreads,
                                                      const uint32_t* Bk_tmp, const uint16_t* Bpos_tmp, uint32_t* Bk,
                                                      uint16_t* Bpos) {
    const uint32_t r = blockIdx.x * kWaves + wave_id();
    if (r >= nreads) return;
    const uint64_t src = tstart[r];
    const uint32_t dst = Bptr[r], cnt = Bptr[r + 1] - dst;
    for (uint32_t x = lane_id(); x < cnt; x += 64) { Bk[dst + x] = Bk_tmp[src + x]; Bpos[dst + x] = Bpos_tmp[src + x]; }
}

__global__ __launch_bounds__(kBlock) void k_entry_rows(const uint32_t* Bptr, uint32_t nreads, uint32_t* Brow) {
    const uint32_t r = blockIdx.x * kWaves + wave_id();
    if (r >= nreads) return;
    for (uint32_t e = Bptr[r] + lane_id(); e < Bptr[r + 1]; e += 64) Brow[e] = r;
}

// per entry: degree histogram, smallest read of the k-mer, orientation bit (occurrence is not the canonical
// form: Kmer::rep() = min(kmer, twin), kmercode/Kmer.cpp:314-317; palindromes get 0) and palindrome bit
__global__ void k_kmer_stats(const uint32_t* Bk, const uint16_t* Bpos, const uint32_t* Brow, uint64_t nnz,
                             const uint32_t* packed, const uint64_t* roff, uint32_t k, uint32_t nkmers, uint32_t* deg,
                             uint32_t* minread, uint8_t* ori, uint32_t* status) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const uint32_t km = Bk[e], r = Brow[e];
    if (km >= nkmers) { atomicOr(status, 32u); return; }
    if ((uint64_t)Bpos[e] + k > roff[r + 1] - roff[r]) { atomicOr(status, 128u); return; }   // the k-mer would run past the end of its read
    const uint64_t le = kmer_le(packed, roff[r] + Bpos[e], k);
    const uint64_t fw = kmer_fw_from_le(le, k), rc = kmer_rc_from_le(le, k);
    ori[e] = (uint8_t)((fw > rc ? 1u : 0u) | (fw == rc ? 2u : 0u));   // bit0: not the canonical form, bit1: palindrome
    atomicAdd(&deg[km], 1u);
    atomicMin(&minread[km], r);
}

// weight of entry e in the "first appearance" layout of A': the owner (smallest read) reserves the whole list
__global__ void k_first_weight(const uint32_t* Bk, const uint32_t* Brow, uint64_t nnz, uint32_t nkmers, const uint32_t* deg,
                               const uint32_t* minread, uint32_t* w, uint32_t* status) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const uint32_t km = Bk[e];
    if (km >= nkmers) { w[e] = 0; return; }                     // flagged by k_kmer_stats; the host stops after this kernel
    if (deg[km] > 16383u) atomicOr(status, 64u);                // Bent's product count field holds 14 bits
    w[e] = (Brow[e] == minread[km]) ? deg[km] : 0u;
}

__global__ void k_col_starts(const uint32_t* Bk, const uint32_t* Brow, uint64_t nnz, const uint32_t* minread,
                             const uint32_t* wscan, uint32_t* colstart) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const uint32_t km = Bk[e];
    if (Brow[e] == minread[km]) colstart[km] = wscan[e];
}

__global__ void k_fill_A(const uint32_t* Bk, const uint32_t* Brow, uint64_t nnz, const uint32_t* colstart,
                         uint32_t* fill, uint2* Atmp) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const uint32_t km = Bk[e];
    const uint32_t slot = colstart[km] + atomicAdd(&fill[km], 1u);
    Atmp[slot] = make_uint2(Brow[e], (uint32_t)e);
}

// one thread per k-mer: order its list by read id (the reference's 1-thread Transpose order,
// transpose.h:26-50) and emit the final A' and B' entries
__global__ void k_finalize_cols(uint32_t nkmers, const uint32_t* deg, const uint32_t* colstart, uint2* Atmp,
                                const uint16_t* Bpos, const uint8_t* ori, const uint64_t* roff, uint2* Aent,
                                uint2* Bent, uint16_t* Bcnt, uint32_t* status) {
    const uint32_t km = blockIdx.x * blockDim.x + threadIdx.x;
    if (km >= nkmers) return;
    const uint32_t dg = deg[km];
    if (dg == 0) return;
    if (dg > 16383u) { atomicOr(status, 64u); return; }
    const uint32_t cs = colstart[km];
    for (uint32_t x = 1; x < dg; ++x) {
        const uint2 v = Atmp[cs + x];
        uint32_t y = x;
        while (y > 0 && Atmp[cs + y - 1].x > v.x) { Atmp[cs + y] = Atmp[cs + y - 1]; --y; }
        if (y != x) Atmp[cs + y] = v;
    }
    for (uint32_t x = 0; x < dg; ++x) {
        const uint2 v = Atmp[cs + x];
        const uint32_t r = v.x, e = v.y;
        const uint32_t o = ori[e] & 1u, pal = (ori[e] >> 1) & 1u, pos = Bpos[e];
        const uint32_t len = (uint32_t)(roff[r + 1] - roff[r]);
        Aent[cs + x] = make_uint2(r | (o << 31), pos | (len << 16));
        Bent[e] = make_uint2(cs + x + 1, pos | ((dg - 1 - x) << 16) | (pal << 30) | (o << 31));
        Bcnt[e] = (uint16_t)(dg - 1 - x);             // the products of the entry once more, compact: estimateFLOP streams 2 B per nonzero
    }
}

}  // namespace bella
