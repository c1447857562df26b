// util.hpp -- block/wave primitives shared by the gfx950 kernels (wave = 64 lanes, block = 256 threads).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bella {

constexpr int kBlock = 256;        // 4 wavefronts
constexpr int kWaves = kBlock / 64;
constexpr uint32_t kOrderedBit = 0x80000000u;   // nnzC[i]: the column's records already stand in the reference's slot order (wide.hpp)

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint32_t wave_id() { return threadIdx.x >> 6; }

// Inclusive scan over the 64 lanes of a wavefront with DPP moves (no LDS traffic, six VALU adds): row_shr 1/2/4/8 scan the four
// rows of 16 lanes (lanes shifted in from outside a row read 0), row_bcast:15 adds a row's total to the next row (rows 1 and 3),
// row_bcast:31 adds the total of the lower half to rows 2 and 3.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
    return v;
}

// Exclusive scan over the threads of a block (NW wavefronts, NW <= 16); *total receives the block sum.
// `scr` must hold NW u32 in LDS.  Contains two barriers; safe to call back to back.  The wavefront totals are combined by a
// second DPP scan inside one row of 16 lanes (every wavefront does it for itself: one LDS read instead of a loop over NW).
template <int NW = kWaves>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* scr, uint32_t* total) {
    static_assert(NW <= 16, "one DPP row holds the wavefront totals");
    const uint32_t inc = wave_incl_scan(v);
    if (lane_id() == 63) scr[wave_id()] = inc;
    __syncthreads();
    uint32_t s = lane_id() < (uint32_t)NW ? scr[lane_id()] : 0u;
    if (NW > 1) s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x111, 0xF, 0xF, false);
    if (NW > 2) s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x112, 0xF, 0xF, false);
    if (NW > 4) s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x114, 0xF, 0xF, false);
    if (NW > 8) s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x118, 0xF, 0xF, false);
    const uint32_t w = wave_id();                              // wave-uniform: lanes w-1 and NW-1 of the row
    const uint32_t sum = (uint32_t)__builtin_amdgcn_readlane((int)s, NW - 1);
    const uint32_t upto = (uint32_t)__builtin_amdgcn_readlane((int)s, (int)__builtin_amdgcn_readfirstlane((int)(w ? w - 1 : 0)));
    const uint32_t base = w ? upto : 0u;
    __syncthreads();
    *total = sum;
    return base + inc - v;
}

__device__ __forceinline__ uint32_t pow2_at_least(uint32_t minsz, uint32_t n) {
    uint32_t s = minsz;
    while (s < n) s <<= 1;
    return s;
}

// ---- B' entries (assemble.hpp: k_layout_emit) -----------------------------------------------------------------------------------------
// Plain form: {index of the first LATER read in the k-mer's list of A', posV | later reads << 16 | palindrome << 30 | orientation << 31}.
// INLINE form (layouts built with it: read ids < 2^30, nnz(A) < 2^31, so that bit 31 of the index is free): an entry whose k-mer
// has exactly ONE later read -- the last but one entry of every list: 58 % of the entries with products at 30x are the first of a list
// of two -- carries that read instead of pointing at it:
// {partner | same orientation << 30 | 1 << 31, posV | partner's posH << 16} (never a palindrome: those keep the plain form).  The
// pass then reads nothing from A' for it: every avoided gather is a 128-byte line from HBM for eight useful bytes.  The partner's
// read length, the one thing its A' entry also held, comes from the read offsets (cache-resident).
__device__ __forceinline__ bool bent_is_inline(uint2 be, uint32_t inl) { return inl && (be.x >> 31); }
__device__ __forceinline__ uint32_t bent_count(uint2 be, uint32_t inl) { return bent_is_inline(be, inl) ? 1u : (be.y >> 16) & 0x3FFFu; }
struct BProduct { uint32_t key, posH, posV, lenH, pal; bool oriented; };
// product t of entry be (t < bent_count)
__device__ __forceinline__ BProduct bent_product(uint2 be, uint32_t t, const uint2* Aent, const uint64_t* roff, uint32_t inl) {
    BProduct r;
    if (bent_is_inline(be, inl)) {
        r.key = be.x & 0x3FFFFFFFu; r.oriented = ((be.x >> 30) & 1u) != 0; r.pal = 0; r.posV = be.y & 0xFFFFu; r.posH = be.y >> 16;
        r.lenH = (uint32_t)(roff[r.key + 1] - roff[r.key]);
    } else {
        const uint2 ae = Aent[(uint64_t)be.x + t];
        r.key = ae.x & 0x7FFFFFFFu; r.posH = ae.y & 0xFFFFu; r.lenH = ae.y >> 16; r.posV = be.y & 0xFFFFu; r.pal = (be.y >> 30) & 1u;
        r.oriented = (ae.x >> 31) == (be.y >> 31);
    }
    return r;
}
// the partner read of product t alone (symbolic phase: no positions)
__device__ __forceinline__ uint32_t bent_partner(uint2 be, uint32_t t, const uint2* Aent, uint32_t inl) {
    return bent_is_inline(be, inl) ? be.x & 0x3FFFFFFFu : Aent[(uint64_t)be.x + t].x & 0x7FFFFFFFu;
}

// any-size table index for our own (non-reference) hash tables: multiplicative hash + range reduction
__device__ __forceinline__ uint32_t hash_range(uint32_t key, uint32_t size) {
    const uint32_t h = key * 2654435761u;
    return (uint32_t)(((uint64_t)h * size) >> 32);
}

}  // namespace bella
