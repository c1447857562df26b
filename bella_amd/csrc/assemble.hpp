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
#include "slotorder.hpp"
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
    const uint32_t* list;   // the reads of this launch (k_asm_classify)
    uint32_t nlist;
#ifdef BELLA_ASM_CLOCK
    unsigned long long* clk;
#endif
};

// reads by table size: class c holds the reads with kAsmClassHt[c-1] < ht <= kAsmClassHt[c]; the last class takes global tables.
// (Exact grids per class: an empty workgroup that reserves tens of KB of LDS still waits for the LDS.)
constexpr uint32_t kAsmClasses = 5;
__device__ __forceinline__ uint32_t asm_class_of(uint32_t ht) { return ht <= 1024u ? 0u : ht <= 2048u ? 1u : ht <= 4096u ? 2u : ht <= 8192u ? 3u : 4u; }
__global__ void k_asm_classify(const uint64_t* tstart, uint32_t nreads, uint32_t* lists, uint32_t* counts, uint32_t* rowcnt, uint32_t* status) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t c = 0xFFFFFFFFu;
    if (r < nreads) {
        const uint64_t n = tstart[r + 1] - tstart[r];
        if (n == 0 || n >= 65536ull) { rowcnt[r] = 0; if (n) atomicOr(status, 16u); }
        else c = asm_class_of(pow2_at_least(16u, (uint32_t)n));
    }
    // one atomic per wavefront and class (five counters serve all reads: one atomic per read queues up behind the others)
    for (uint32_t q = 0; q < kAsmClasses; ++q) {
        const unsigned long long mask = __ballot(c == q);
        if (mask == 0) continue;
        uint32_t base = 0;
        if (lane_id() == (uint32_t)(__ffsll((long long)mask) - 1)) base = atomicAdd(&counts[q], (uint32_t)__popcll(mask));
        base = __shfl(base, __ffsll((long long)mask) - 1, 64);
        if (c == q) lists[(size_t)q * nreads + base + (uint32_t)__popcll(mask & ((1ull << lane_id()) - 1ull))] = r;
    }
}
constexpr uint32_t kAsmLdsSlots = 8192;                    // largest table held in LDS (one 1024-thread workgroup per CU)
constexpr uint32_t kAsmScratchBytes = 128;
// the de-duplication table holds 1.4 slots per tuple (any size: multiplicative hash + range reduction)
__host__ __device__ inline uint32_t asm_dedup_slots(uint32_t n) { return (uint32_t)(((uint64_t)n * 7 + 4) / 5) | 1u; }
__host__ __device__ inline size_t asm_lds_bytes(uint32_t ht) { return kAsmScratchBytes + (size_t)8 * (asm_dedup_slots(ht) + 1) + (size_t)6 * ht; }

// One read: de-duplicate (keep the FIRST occurrence index for the slot order, the LAST occurrence's position as the
// value: CSC.cpp:344 with main.cpp:477-480's lambda), then reproduce the insertion order into the reference's
// table of size ht = 2^n >= max(16, #tuples) (CSC.cpp:322-326) -- slotorder.hpp: rounds by insertion time -- and emit the slots in
// order (CSC.cpp:358-373).
// Kk / Kf: de-duplication table of KS slots (k-mer id; first occurrence << 16 | last occurrence, tuple indices inside the read);
// T2: the slot-order table (first occurrence in both halves of the word); nf: its next-free table.
// One compare-and-swap per tuple: the thread that claims a slot stores Kf with a plain store; a duplicate k-mer inside a read
// (rare: repeats) only raises a flag, and a flagged read takes one more pass that folds every tuple into Kf with a CAS loop.
template <int BLK>
__device__ __forceinline__ void asm_row(const AsmArgs& a, uint32_t r, uint32_t* scr, uint32_t* Kk, uint32_t* Kf, uint32_t KS, uint32_t* T2,
                                        uint16_t* nf, uint32_t ht) {
    constexpr int NW = BLK / 64;
    const uint32_t tid = threadIdx.x;
#ifdef BELLA_ASM_CLOCK
    unsigned long long c_prev = clock64();
#define ACLK(i) do { if (tid == 0) { const unsigned long long c_now = clock64(); atomicAdd(&a.clk[i], c_now - c_prev); c_prev = c_now; } } while (0)
#else
#define ACLK(i) do {} while (0)
#endif
    const uint64_t ts = a.tstart[r];
    const uint32_t n = (uint32_t)(a.tstart[r + 1] - ts);
    uint32_t* s_d = scr + 24;
    uint32_t* s_dup = scr + 25;
    // the first kPre * BLK k-mer ids (every tuple of a read whose table is in LDS) travel while the tables are initialised
    constexpr uint32_t kPre = 4;
    uint32_t k0[kPre];
#pragma unroll
    for (uint32_t u = 0; u < kPre; ++u) { const uint32_t t = tid + u * BLK; k0[u] = t < n ? a.t_kmer[ts + t] : 0u; }
    for (uint32_t s = tid; s < KS; s += BLK) Kk[s] = kEmpty;
    for (uint32_t s = tid; s < ht; s += BLK) T2[s] = kEmpty;
    if (tid == 0) { *s_d = 0; *s_dup = 0; }
    __syncthreads();
    ACLK(0);
    uint32_t mine = 0;
    auto insert = [&](uint32_t key, uint32_t t) {
        uint32_t h = hash_range(key, KS);
        uint32_t old;
        for (;;) {
            old = atomicCAS(&Kk[h], kEmpty, key);
            if (old == kEmpty || old == key) break;
            h = h + 1 == KS ? 0 : h + 1;
        }
        if (old == kEmpty) { Kf[h] = (t << 16) | t; ++mine; }
        else *s_dup = 1;
    };
#pragma unroll
    for (uint32_t u = 0; u < kPre; ++u) { const uint32_t t = tid + u * BLK; if (t < n) insert(k0[u], t); }
    for (uint32_t t = tid + kPre * BLK; t < n; t += BLK) insert(a.t_kmer[ts + t], t);
    {   // one LDS atomic per wavefront (BLK same-address atomics would queue up behind each other)
        const uint32_t mw = wave_incl_scan(mine);
        if (lane_id() == 63 && mw) atomicAdd(s_d, mw);
    }
    __syncthreads();
    ACLK(1);
    const bool dup = *s_dup != 0;
    if (dup) {                                               // some k-mer occurs twice in this read: first = min, last = max over ALL tuples
        for (uint32_t t = tid; t < n; t += BLK) {
            const uint32_t key = a.t_kmer[ts + t];
            uint32_t h = hash_range(key, KS);
            while (Kk[h] != key) h = h + 1 == KS ? 0 : h + 1;
            uint32_t cur = Kf[h];
            for (;;) {
                const uint32_t f = cur >> 16, l = cur & 0xFFFFu;
                const uint32_t want = ((t < f ? t : f) << 16) | (t > l ? t : l);
                if (want == cur) break;
                const uint32_t got = atomicCAS(&Kf[h], cur, want);
                if (got == cur) break;
                cur = got;
            }
        }
        __syncthreads();
    }
    ACLK(2);
    const uint32_t d = *s_d;
    // the distinct k-mers enter the reference's table in the order of their first occurrences, in rounds (slotorder.hpp)
    uint32_t prev = 0;
    for (uint32_t rd = 0;; ++rd) {
        const uint32_t bound = round_bound(rd, ht, d, n);
        if (rd) { (void)build_next_free<BLK>(T2, nf, ht, scr); __syncthreads(); }
        if (!dup && rd == 0) {                                // every tuple is the first occurrence of its k-mer: straight from the tuples,
#pragma unroll
            for (uint32_t u = 0; u < kPre; ++u) {             // whose ids are still in registers
                const uint32_t t = tid + u * BLK;
                if (t < bound) slot_insert<false>(T2, nf, ht - 1, (k0[u] * 107u) & (ht - 1), (t << 16) | t);
            }
            for (uint32_t t = tid + kPre * BLK; t < bound; t += BLK)
                slot_insert<false>(T2, nf, ht - 1, (a.t_kmer[ts + t] * 107u) & (ht - 1), (t << 16) | t);
        } else if (!dup) {
            for (uint32_t t = prev + tid; t < bound; t += BLK)
                slot_insert<true>(T2, nf, ht - 1, (a.t_kmer[ts + t] * 107u) & (ht - 1), (t << 16) | t);
        } else {
            for (uint32_t s = tid; s < KS; s += BLK) {
                const uint32_t key = Kk[s];
                if (key == kEmpty) continue;
                const uint32_t first = Kf[s] >> 16;
                if (first < prev || first >= bound) continue;
                const uint32_t home = (key * 107u) & (ht - 1);
                if (rd) slot_insert<true>(T2, nf, ht - 1, home, (first << 16) | first);
                else slot_insert<false>(T2, nf, ht - 1, home, (first << 16) | first);
            }
        }
        __syncthreads();
        if (bound >= n) break;
        prev = bound;
    }
    ACLK(3);
    const uint32_t c = (ht + BLK - 1) / BLK;
    const uint32_t lo = tid * c < ht ? tid * c : ht;
    const uint32_t hi = lo + c < ht ? lo + c : ht;
    uint32_t occ = 0;
    for (uint32_t s = lo; s < hi; ++s) occ += (T2[s] != kEmpty);
    uint32_t tot;
    uint32_t rank = block_excl_scan<NW>(occ, scr, &tot);
    ACLK(4);
    for (uint32_t s = lo; s < hi; ++s) {
        const uint32_t it = T2[s];
        if (it == kEmpty) continue;
        const uint32_t f = it & 0xFFFFu;
        const uint32_t key = a.t_kmer[ts + f];
        uint32_t last = f;
        if (dup) {
            uint32_t h = hash_range(key, KS);
            while (Kk[h] != key) h = h + 1 == KS ? 0 : h + 1;
            last = Kf[h] & 0xFFFFu;
        }
        a.Bk_tmp[ts + rank] = key;
        a.Bpos_tmp[ts + rank] = a.t_pos[ts + last];
        rank++;
    }
    if (tid == 0) a.rowcnt[r] = d;
    __syncthreads();
    ACLK(5);
}

// LDS classes by table size: one workgroup per read of the class's list, the workgroup size grows with the table so that a CU's
// wavefront slots stay filled
template <int BLK>
__global__ __launch_bounds__(BLK) void k_asm_rows_lds(AsmArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t r = a.list[blockIdx.x];
    const uint32_t n = (uint32_t)(a.tstart[r + 1] - a.tstart[r]);
    const uint32_t ht = pow2_at_least(16u, n);
    const uint32_t KS = asm_dedup_slots(ht);                 // what the LDS of a table of ht slots holds: 1.4 ... 2.8 slots per tuple, short probes
    uint32_t* scr = (uint32_t*)smem;
    uint32_t* T2 = (uint32_t*)(smem + kAsmScratchBytes);
    uint32_t* Kk = T2 + ht;
    uint32_t* Kf = Kk + KS;
    uint16_t* nf = (uint16_t*)(Kf + KS);
    asm_row<BLK>(a, r, scr, Kk, Kf, KS, T2, nf, ht);
}

// reads whose table does not fit LDS (> 8192 tuples): tables in a global workspace, persistent workgroups
__global__ __launch_bounds__(1024) void k_asm_rows_global(AsmArgs a) {
    __shared__ uint32_t scr[kAsmScratchBytes / 4];
    uint32_t* W = (uint32_t*)(a.ws + (uint64_t)blockIdx.x * a.ws_stride);
    for (uint32_t x = blockIdx.x; x < a.nlist; x += gridDim.x) {
        const uint32_t r = a.list[x];
        const uint32_t n = (uint32_t)(a.tstart[r + 1] - a.tstart[r]);
        const uint32_t ht = pow2_at_least(16u, n);
        const uint32_t KS = asm_dedup_slots(n);
        asm_row<1024>(a, r, scr, W + ht, W + ht + KS, KS, W, (uint16_t*)(W + ht + 2 * KS), ht);
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

// ---- device layout (B' / A') from the CSR of B: one stable radix sort + segmented passes ------------------------------------
// The k-mer lists of A' are the runs of equal k-mer id in the entries of B sorted by k-mer id; the sort is stable and the entries
// of B are grouped by ascending read, so every run comes out in ascending read order = the reference's 1-thread Transpose
// (transpose.h:26-50).  The sort carries everything the emit pass needs per entry in a 64-bit value, so that pass gathers nothing
// by entry index: hi = read | ori << 31 (= Aent.x), lo = pos | place in the read's row << 16.

// per entry e of B (one wavefront per read row): sort key (k-mer id), sort value, validation; clears the scatter target of
// k_layout_heads.  ori: the occurrence is not the canonical form (Kmer::rep() = min(kmer, twin), kmercode/Kmer.cpp:314-317).
__global__ __launch_bounds__(kBlock) void k_layout_prep(const uint32_t* Bptr, const uint32_t* Bk, const uint16_t* Bpos, uint32_t nreads,
                                                        const uint32_t* packed, const uint64_t* roff, uint32_t k, uint32_t nkmers,
                                                        uint32_t* key, uint64_t* val, uint32_t* w, uint32_t* status) {
    const uint32_t r = blockIdx.x * kWaves + wave_id();
    if (r >= nreads) return;
    const uint64_t base = roff[r];
    const uint32_t len = (uint32_t)(roff[r + 1] - base);
    const uint32_t b0 = Bptr[r], b1 = Bptr[r + 1];
    for (uint32_t e = b0 + lane_id(); e < b1; e += 64) {
        const uint32_t km = Bk[e], pos = Bpos[e];
        uint32_t ori = 0, pal = 0;
        if (km >= nkmers) atomicOr(status, 32u);
        if (pos + k > len) atomicOr(status, 128u);                       // the k-mer would run past the end of its read
        else {
            const uint64_t le = kmer_le(packed, base + pos, k);
            const uint64_t fw = kmer_fw_from_le(le, k), rc = kmer_rc_from_le(le, k);
            ori = fw > rc ? 1u : 0u;
            pal = fw == rc ? 1u : 0u;                                    // (a property of the k-mer: the same for all its occurrences)
        }
        key[e] = km;
        // read ids below 2^30 leave bit 30 of the high word to the palindrome flag: the emit pass then reads no sequence at all
        val[e] = ((uint64_t)(r | (ori << 31) | (nreads <= (1u << 30) ? pal << 30 : 0u)) << 32) | (pos | ((e - b0) << 16));
        if (w) w[e] = 0;                                                     // (only the first-appearance layout scans w)
    }
}

// per run head of the sorted entries: its length (the k-mer's degree) lands at the entry of its first (= smallest) read; the
// exclusive scan of that array over the entries of B is the "first appearance" layout of A': the owner row streams its lists.
__global__ void k_layout_heads(const uint32_t* skey, const uint64_t* sval, uint64_t nnz, const uint32_t* Bptr, uint32_t rmask, uint32_t* w,
                               uint32_t* status) {
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= nnz) return;
    const uint32_t km = skey[x];
    if (x && skey[x - 1] == km) return;
    uint32_t dg = 1;
    while (x + dg < nnz && dg <= 16384u && skey[x + dg] == km) ++dg;
    if (dg > 16383u) { atomicOr(status, 64u); dg = 0; }                  // Bent's product count field holds 14 bits
    const uint64_t v = sval[x];
    w[Bptr[(uint32_t)(v >> 32) & rmask] + ((uint32_t)v >> 16)] = dg;
}

// first / one-past-last position of the run of equal keys around x (runs are short: a few steps; long ones by bisection)
__device__ __forceinline__ void run_bounds(const uint32_t* skey, uint64_t nnz, uint64_t x, uint64_t& lo, uint64_t& hi) {
    const uint32_t km = skey[x];
    uint64_t a = x;
    uint32_t j = 0;
    while (a > 0 && j < 16 && skey[a - 1] == km) { --a; ++j; }
    if (j == 16 && a > 0 && skey[a - 1] == km) {                         // lower bound in [0, a)
        uint64_t l = 0, h = a;
        while (l < h) { const uint64_t m = (l + h) >> 1; if (skey[m] < km) l = m + 1; else h = m; }
        a = l;
    }
    uint64_t b = x + 1;
    j = 0;
    while (b < nnz && j < 16 && skey[b] == km) { ++b; ++j; }
    if (j == 16 && b < nnz && skey[b] == km) {                           // upper bound in (b, nnz]
        uint64_t l = b, h = nnz;
        while (l < h) { const uint64_t m = (l + h) >> 1; if (skey[m] <= km) l = m + 1; else h = m; }
        b = l;
    }
    lo = a; hi = b;
}

// per sorted entry: its place in A' and the final A' and B' entries (coalesced reads; A' written run by run).  The B' entry belongs at
// the entry's own index e -- a random 8-byte write per nonzero if done from here -- so it leaves as (e, entry) in sorted order; one
// radix pass on the top bits of e and k_layout_place then write B' region by region (the writes of a region meet in the caches).
// by_kmer (the row-list layout, k_layout_rowlists below): the lists of A' stay in k-mer order = the sorted order itself -- no list
// starts to scatter, scan and look up (k_layout_heads is not run), A' is written in place.
// per read: its first B' entry and its length side by side -- the two things k_layout_emit looks up per ENTRY by read id (one 8-byte
// load instead of three from two arrays: that kernel runs at the L2's request rate, not at HBM's)
__global__ void k_layout_rinfo(const uint32_t* Bloc, const uint64_t* roff, uint32_t nreads, uint2* rinfo) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < nreads) rinfo[r] = make_uint2(Bloc[r], (uint32_t)(roff[r + 1] - roff[r]));
}
constexpr int kLayoutEmitPartBlock = 1024;      // workgroup of the partitioned launch (own_stride > 1); at most this otherwise
constexpr uint32_t kEmitHalo = 32;              // keys on either side of the workgroup's entries in its LDS window
constexpr uint32_t kOwnerMax = 64;              // ranks of a communicator the shared formation of A' serves (build_layout, dist)
__global__ void k_layout_emit(const uint32_t* skey, const uint64_t* sval, uint64_t nnz, const uint32_t* Bptr, const uint32_t* Bloc, const uint32_t* wscan,
                              const uint32_t* packed, const uint64_t* roff, uint32_t k, uint32_t rmask, uint2* Aent, uint32_t* ekey, uint64_t* eval,
                              uint32_t by_kmer, uint32_t own_first, uint32_t own_stride, uint32_t* counter, uint32_t* status, uint32_t inl,
                              const uint2* rinfo, uint32_t abase = 0, unsigned long long* cursor = nullptr) {
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool mine = false;
    uint32_t dstkey = 0, owner = 0;
    uint64_t dstval = 0;
    // the workgroup's keys and 32 on either side in LDS: a run's bounds are a few LDS reads (they were a chain of dependent global
    // loads per entry: 4.05 ms of this kernel's time at 100k reads was that chain); runs that leave the window take the global search
    __shared__ uint32_t s_key[kLayoutEmitPartBlock + 2 * kEmitHalo];
    const uint64_t x0 = (uint64_t)blockIdx.x * blockDim.x;
    for (uint32_t i = threadIdx.x; i < blockDim.x + 2u * kEmitHalo; i += blockDim.x) {
        const uint64_t g = x0 + i;                                           // (window position i = entry x0 - halo + i)
        s_key[i] = g >= kEmitHalo && g - kEmitHalo < nnz ? skey[g - kEmitHalo] : 0xFFFFFFFFu;   // (no k-mer id: ids are < nkmers < 2^32 - 16)
    }
    __syncthreads();
    if (x < nnz) {
        uint64_t lo, hi;
        {
            const uint32_t me = threadIdx.x + kEmitHalo, km = s_key[me], wend = blockDim.x + 2u * kEmitHalo;
            uint32_t a = me, b = me + 1;
            while (a > 0 && s_key[a - 1] == km) --a;
            while (b < wend && s_key[b] == km) ++b;
            if ((a == 0 && x0 > kEmitHalo) || (b == wend && x0 + blockDim.x + kEmitHalo < nnz)) run_bounds(skey, nnz, x, lo, hi);   // the run may go on outside
            else { lo = x0 + a - kEmitHalo; hi = x0 + b - kEmitHalo; }
        }
        const uint32_t dg = (uint32_t)(hi - lo), rk = (uint32_t)(x - lo);
        const bool need_first = !by_kmer || rmask != 0x3FFFFFFFu;            // (the default layout never looks at the run's first entry)
        const uint64_t vf = need_first ? sval[lo] : 0ull, v = sval[x];
        const uint32_t rf = (uint32_t)(vf >> 32) & rmask;
        if (dg > 16383u && rk == 0) atomicOr(status, 64u);                   // Bent's product count field holds 14 bits
        // by_kmer: the list starts where its run starts; else at the scanned degree of the run's first entry (k_layout_heads: indexed
        // by that entry's index in the WHOLE matrix)
        const uint32_t cs = by_kmer ? (uint32_t)lo : wscan[Bptr[rf] + ((uint32_t)vf >> 16)];
        const uint32_t hiw = (uint32_t)(v >> 32), r = hiw & rmask, pos = (uint32_t)v & 0xFFFFu;
        uint32_t pal;
        if (rmask == 0x3FFFFFFFu) pal = (hiw >> 30) & 1u;
        else {                                                                // (2^30 reads or more: from the sequence of the first occurrence)
            const uint64_t le = kmer_le(packed, roff[rf] + ((uint32_t)vf & 0xFFFFu), k);
            pal = kmer_fw_from_le(le, k) == kmer_rc_from_le(le, k) ? 1u : 0u;
        }
        const uint32_t ori = pal ? 0u : hiw >> 31;                            // (palindromes count as canonical)
        const uint2 ri = rinfo[r];                                            // {first B' entry of read r, its length}
        const uint32_t len = ri.y;
        // (abase: the sorted entries are one rank's SLICE of the k-mer id space -- build_layout's shared formation of A' --: the list
        // positions of the slice start at abase in the whole A')
        Aent[abase + cs + rk] = make_uint2(r | (ori << 31), pos | (len << 16));
        owner = r % own_stride;
        mine = own_stride == 1u || owner == own_first;                        // B' entries only for the columns this context owns
        dstkey = ri.x + ((uint32_t)v >> 16);
        dstval = (uint64_t)(abase + cs + rk + 1) | ((uint64_t)(pos | ((dg - 1 - rk) << 16) | (pal << 30) | (ori << 31)) << 32);
        if (inl && rk + 2u == dg && !pal) {                                   // exactly one later read (the last but one of any list): the entry carries it (util.hpp)
            const uint64_t v1 = sval[x + 1];
            const uint32_t hi1 = (uint32_t)(v1 >> 32);
            dstval = (uint64_t)((hi1 & rmask) | ((hi1 >> 31) == ori ? 1u << 30 : 0u) | (1u << 31)) | ((uint64_t)(pos | (((uint32_t)v1 & 0xFFFFu) << 16)) << 32);
        }
    }
    if (cursor) {                                                             // shared formation: every entry leaves, into the range of the rank that owns its row
        __shared__ uint32_t s_oc[kOwnerMax];                                   // (one atomic per owner and workgroup on the ranges' cursors)
        __shared__ unsigned long long s_ob[kOwnerMax];
        if (threadIdx.x < kOwnerMax) s_oc[threadIdx.x] = 0;
        __syncthreads();
        uint32_t at = 0;
        if (x < nnz) at = atomicAdd(&s_oc[owner], 1u);
        __syncthreads();
        if (threadIdx.x < own_stride && s_oc[threadIdx.x]) s_ob[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], (unsigned long long)s_oc[threadIdx.x]);
        __syncthreads();
        if (x < nnz) { const unsigned long long d = s_ob[owner] + at; ekey[d] = dstkey; eval[d] = dstval; }
        return;
    }
    if (own_stride == 1u) {                                                   // every entry has a B' entry: it leaves at its sorted place
        if (x < nnz) { ekey[x] = dstkey; eval[x] = dstval; }
        return;
    }
    // partitioned: the owned entries leave compacted, ONE atomic per workgroup of 1,024 entries on the shared counter (a counter takes
    // about 90 atomics per microsecond: one per wavefront would be 3 M of them at 100k reads); their order does not matter, the
    // partition pass and k_layout_place put every entry at its own index
    __shared__ uint32_t scr[16];
    __shared__ uint32_t s_base;
    uint32_t tot;
    const uint32_t ex = block_excl_scan<16>(mine ? 1u : 0u, scr, &tot);
    if (threadIdx.x == 0) s_base = tot ? atomicAdd(counter, tot) : 0u;
    __syncthreads();
    if (mine) { const uint32_t o = s_base + ex; ekey[o] = dstkey; eval[o] = dstval; }
}

// ---- the formation of A' shared over the ranks of a communicator (bella_hip.hip: build_layout, dist) --------------------------------
// Rank g of N keeps the entries of B whose k-mer id lies in [klo, khi) -- the g-th N-th of the id space --, sorts THEM (1/N of the sort),
// emits its slice of A' and the B' entries of ALL rows for those k-mers; the slices and the entries then travel (one grouped all-gather,
// one grouped all-to-all).  The compaction of a row's in-range entries is stable (the sort is, too: a list of A' stays in ascending read
// order).
__global__ __launch_bounds__(kBlock) void k_range_rowcount(const uint32_t* Bptr, const uint32_t* Bk, uint32_t nreads, uint32_t klo, uint32_t khi,
                                                           uint32_t* cnt, uint32_t nkmers, uint32_t* status) {
    const uint32_t r = blockIdx.x * kWaves + wave_id();
    if (r > nreads) return;
    uint32_t n = 0;
    if (r < nreads)
        for (uint32_t e = Bptr[r] + lane_id(); e < Bptr[r + 1]; e += 64) {
            const uint32_t km = Bk[e];
            n += km >= klo && km < khi ? 1u : 0u;
            if (km >= nkmers) atomicOr(status, 32u);                     // (every rank sees every entry here: the bad input is reported, not a slice mismatch)
        }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) n += __shfl_xor(n, d, 64);
    if (lane_id() == 0) cnt[r] = n;
}
// k_layout_prep for the in-range entries only, written compacted at rowbase[r] + (rank inside the row)
__global__ __launch_bounds__(kBlock) void k_layout_prep_range(const uint32_t* Bptr, const uint32_t* Bk, const uint16_t* Bpos, uint32_t nreads,
                                                              const uint32_t* packed, const uint64_t* roff, uint32_t k, uint32_t nkmers,
                                                              uint32_t klo, uint32_t khi, const uint32_t* rowbase, uint32_t* key, uint64_t* val,
                                                              uint32_t* status) {
    const uint32_t r = blockIdx.x * kWaves + wave_id();
    if (r >= nreads) return;
    const uint64_t base = roff[r];
    const uint32_t len = (uint32_t)(roff[r + 1] - base);
    const uint32_t b0 = Bptr[r], b1 = Bptr[r + 1];
    uint32_t out = rowbase[r];
    for (uint32_t e0 = b0; e0 < b1; e0 += 64) {
        const uint32_t e = e0 + lane_id();
        uint32_t km = 0, pos = 0;
        bool in = false;
        if (e < b1) {
            km = Bk[e]; pos = Bpos[e];
            if (km >= nkmers) atomicOr(status, 32u);
            if (pos + k > len) atomicOr(status, 128u);
            in = km >= klo && km < khi;
        }
        const unsigned long long m = __ballot(in);
        if (in) {
            uint32_t ori = 0, pal = 0;
            if (pos + k <= len) {
                const uint64_t le = kmer_le(packed, base + pos, k);
                const uint64_t fw = kmer_fw_from_le(le, k), rc = kmer_rc_from_le(le, k);
                ori = fw > rc ? 1u : 0u;
                pal = fw == rc ? 1u : 0u;
            }
            const uint32_t o = out + (uint32_t)__popcll(m & ((1ull << lane_id()) - 1ull));
            key[o] = km;
            val[o] = ((uint64_t)(r | (ori << 31) | (nreads <= (1u << 30) ? pal << 30 : 0u)) << 32) | (pos | ((e - b0) << 16));
        }
        out += (uint32_t)__popcll(m);
    }
}
// the rows' lengths in OWNER-major order: P[(r % N) * M + r / N] = length of row r (M = ceil(nreads / N)); the exclusive scan of P,
// minus the scan at the owner's first place, is where row r starts in ITS owner's compacted B'
__global__ void k_owner_lengths(const uint32_t* Bptr, uint32_t nreads, uint32_t N, uint32_t M, uint32_t* P) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x > N * M) return;
    const uint32_t o = x / M, j = x % M;
    const uint64_t r = (uint64_t)j * N + o;
    P[x] = x < N * M && r < nreads ? Bptr[r + 1] - Bptr[r] : 0u;
}
__global__ void k_layout_rinfo_owner(const uint32_t* scanP, const uint64_t* roff, uint32_t nreads, uint32_t N, uint32_t M, uint2* rinfo) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nreads) return;
    const uint32_t o = r % N, j = r / N;
    rinfo[r] = make_uint2(scanP[o * M + j] - scanP[o * M], (uint32_t)(roff[r + 1] - roff[r]));
}
// B' entries by owner: how many entries of this rank's slice belong to the rows of each rank (from the per-row counts: the owner of row r
// is r % N); k_layout_emit then hands every entry a place in its owner's range (order inside a range: any -- k_layout_place puts
// every entry at its own index)
__global__ __launch_bounds__(256) void k_owner_hist_rows(const uint32_t* cnt, uint32_t nreads, uint32_t N, unsigned long long* hist) {
    __shared__ uint32_t s_h[kOwnerMax];
    if (threadIdx.x < kOwnerMax) s_h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < nreads) { const uint32_t n = cnt[r]; if (n) atomicAdd(&s_h[r % N], n); }    // (256 rows of < 2^16 entries: no overflow)
    __syncthreads();
    if (threadIdx.x < N && s_h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)s_h[threadIdx.x]);
}

// XCD-aware: consecutive workgroups go to the eight XCDs in turn, each with an L2 of its own.  Walking the sorted entries in launch
// order, all eight would write into the SAME region of B' at any time and each L2 would hold, and evict, its own partial copy of every
// line; here XCD j takes the j-th eighth of the entries (32 regions of its own), so that the writes to a line meet in one L2.
__global__ void k_layout_place(const uint32_t* ekey, const uint64_t* eval, uint64_t n, uint2* Bent) {
    const uint32_t per = gridDim.x >> 3;                                      // (the grid is a multiple of eight workgroups)
    const uint64_t chunk = (uint64_t)(blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    const uint64_t x = chunk * blockDim.x + threadIdx.x;
    if (x >= n) return;
    const uint64_t v = eval[x];
    Bent[ekey[x]] = make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}

// rows of B' of a partitioned context: column i keeps its entries when i % stride == first, else none (the exclusive scan of these
// lengths is the context's own row pointer array, Bloc)
__global__ void k_layout_own_lengths(const uint32_t* Bptr, uint32_t nreads, uint32_t own_first, uint32_t own_stride, uint32_t* len) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > nreads) return;
    len[i] = i < nreads && i % own_stride == own_first ? Bptr[i + 1] - Bptr[i] : 0u;
}

// ---- B' without the entries that have no later read (round 6) ---------------------------------------------------------------------
// An entry of row i whose k-mer occurs in no read > i has no product in the strict lower triangle (overlap.hpp:157-202 counts none for
// it, LocalSpGEMM :281-363 skips its whole list): 38 % of the entries at 30x coverage, nearly all entries of the last columns.  Every
// pass streamed them, decoded them and scanned their zero counts.  They are dropped here, once: one wavefront per row counts its live
// entries, a scan gives the new row pointers, a second pass moves the live entries -- in their order, so the product order of a column
// (entry by entry, then down the list) is what it was.  BELLA_TUNE_COMPACT_B 1 keeps every entry (tests, A/B).
__global__ __launch_bounds__(kBlock) void k_layout_live(const uint32_t* Bloc, const uint2* Bent, uint32_t nreads, uint32_t inl, uint32_t* len,
                                                        uint32_t* rowF) {
    const uint32_t i = blockIdx.x * kWaves + wave_id();
    if (i > nreads) return;
    uint32_t n = 0, f = 0;
    if (i < nreads)
        for (uint32_t e = Bloc[i] + lane_id(); e < Bloc[i + 1]; e += 64) { const uint32_t cnt = bent_count(Bent[e], inl); n += cnt ? 1u : 0u; f += cnt; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { n += __shfl_xor(n, d, 64); f += __shfl_xor(f, d, 64); }
    if (lane_id() == 0) {
        if (len) len[i] = n;                                          // (len[nreads] = 0: the scan's last element)
        // the row's PRODUCTS (estimateFLOP, overlap.hpp:157-202: the sum of the entries' later-read counts) -- a property of the operands
        // like a row pointer: the passes read four bytes per column instead of streaming a count per entry (round 5: 2 B per nonzero per pass)
        rowF[i] = f;
    }
}
__global__ __launch_bounds__(kBlock) void k_layout_compact(const uint32_t* Bloc, const uint32_t* Bnew, const uint2* Bent, uint32_t nreads, uint32_t inl,
                                                           uint2* out) {
    const uint32_t i = blockIdx.x * kWaves + wave_id();
    if (i >= nreads) return;
    const uint32_t b0 = Bloc[i], b1 = Bloc[i + 1];
    uint32_t o = Bnew[i];
    for (uint32_t e0 = b0; e0 < b1; e0 += 64) {
        const uint32_t e = e0 + lane_id();
        uint2 be = make_uint2(0u, 0u);
        if (e < b1) be = Bent[e];
        const bool live = e < b1 && bent_count(be, inl) != 0;
        const unsigned long long m = __ballot(live);
        if (live) out[o + (uint32_t)__popcll(m & ((1ull << lane_id()) - 1ull))] = be;
        o += (uint32_t)__popcll(m);
    }
}

// ---- row lists: the products of every column, ready-made, in product order -----------------------------------------------------
// The SpGEMM of column i reads, for each entry of row i, the tail of that k-mer's list in A' (the later reads): with A' alone that is
// one random 8..56-byte read per entry and pass, plus the expansion of the entries into products and the overlap estimate of every
// product.  None of it depends on the pass: here the products are written ONCE, at assembly time, in the order a column generates
// them (entry by entry, then down the list): Aent2[Arow[i] + p] = {partner read | palindrome << 30 | same orientation << 31,
// posH | posV << 16}, Aov[...] = the partner's read length -- the two operand entries of the product side by side; the multiply of
// the semiring (the overlap estimate, chain.hpp:47-71) stays in the pass.  The numeric phase streams them -- coalesced, no index per
// product, no B' entries, no expansion.  10 bytes per product of the whole SpGEMM (1.8 GB at 100k reads); needs read ids < 2^30.
// one workgroup per row (grid-stride), its entries in rounds of 1024: block scan of the counts, then groups of L lanes copy the tail
// of one entry's list each (L = 1, 4 or 16 by the round's products per entry: short lists -- PacBio-like input, less than one later
// read per entry -- one lane per entry, four independent loads in flight; long lists -- HiFi-like input, tens of later reads -- a
// group per entry, so that a store instruction writes runs of L neighbouring products).  The tails are contiguous in A', the
// products of neighbouring entries are neighbours in the output.  cols == nullptr: all rows, row i starts at starts[i] (= Arow);
// else the rows cols[0 .. nrows), row cols[x] starts at starts[x] (the wide columns of a batch expanded into a temporary list, wide.hpp).
constexpr int kRowListBlock = 1024;
__global__ __launch_bounds__(kRowListBlock) void k_layout_rowlists(const uint32_t* Bptr, const uint2* Bent, const uint2* Aent, const uint64_t* starts,
                                                                   const uint64_t* roff, const uint32_t* cols, uint32_t nrows, uint2* Aent2, uint16_t* Aov,
                                                                   uint32_t inl) {
    __shared__ uint32_t scr[kRowListBlock / 64];
    __shared__ uint32_t s_off[kRowListBlock];
    __shared__ uint2 s_be[kRowListBlock];
    for (uint32_t x = blockIdx.x; x < nrows; x += gridDim.x) {
        const uint32_t i = cols ? cols[x] : x;
        const uint32_t b0 = Bptr[i], n = Bptr[i + 1] - b0;
        const uint64_t o = starts[x];
        uint64_t running = 0;
        for (uint32_t jb = 0; jb < n; jb += kRowListBlock) {
            const uint32_t j = jb + threadIdx.x;
            const uint32_t nround = n - jb < (uint32_t)kRowListBlock ? n - jb : (uint32_t)kRowListBlock;
            uint2 be = make_uint2(0u, 0u);
            if (j < n) be = Bent[b0 + j];
            const uint32_t cnt = bent_count(be, inl);
            uint32_t tot;
            const uint32_t ex = block_excl_scan<kRowListBlock / 64>(cnt, scr, &tot);
            const uint32_t L = tot <= 2u * nround ? 1u : tot <= 12u * nround ? 4u : 16u;   // (uniform over the workgroup)
            auto copy_tail = [&](const uint2 eb, const uint32_t ecnt, const uint64_t dst, const uint32_t t_first, const uint32_t t_step) {
                if (bent_is_inline(eb, inl)) {                                // the one product is in the entry
                    if (t_first == 0u) {
                        const BProduct pr = bent_product(eb, 0u, Aent, roff, inl);
                        Aent2[dst] = make_uint2(pr.key | (pr.oriented ? 0x80000000u : 0u), pr.posH | (pr.posV << 16));
                        Aov[dst] = (uint16_t)pr.lenH;
                    }
                    return;
                }
                const uint32_t posV = eb.y & 0xFFFFu, pal = (eb.y >> 30) & 1u, oriB = eb.y >> 31;
                for (uint32_t t0 = t_first; t0 < ecnt; t0 += 4 * t_step) {
                    uint2 ae[4];
#pragma unroll
                    for (uint32_t u = 0; u < 4; ++u) { ae[u] = make_uint2(0u, 0u); if (t0 + u * t_step < ecnt) ae[u] = Aent[(uint64_t)eb.x + t0 + u * t_step]; }
#pragma unroll
                    for (uint32_t u = 0; u < 4; ++u) {
                        const uint32_t t = t0 + u * t_step;
                        if (t >= ecnt) continue;
                        const uint32_t posH = ae[u].y & 0xFFFFu, lenH = ae[u].y >> 16;
                        const bool oriented = (ae[u].x >> 31) == oriB;
                        Aent2[dst + t] = make_uint2((ae[u].x & 0x3FFFFFFFu) | (pal << 30) | (oriented ? 0x80000000u : 0u), posH | (posV << 16));
                        Aov[dst + t] = (uint16_t)lenH;
                    }
                }
            };
            if (L == 1u) {
                copy_tail(be, cnt, o + running + ex, 0u, 1u);
            } else {
                s_off[threadIdx.x] = ex; s_be[threadIdx.x] = be;
                __syncthreads();
                for (uint32_t e = threadIdx.x / L; e < nround; e += kRowListBlock / L) {
                    const uint2 eb = s_be[e];
                    copy_tail(eb, bent_count(eb, inl), o + running + s_off[e], threadIdx.x % L, L);
                }
                __syncthreads();
            }
            running += tot;
        }
    }
}

}  // namespace bella
