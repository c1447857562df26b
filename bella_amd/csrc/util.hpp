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

// any-size table index for our own (non-reference) hash tables: multiplicative hash + range reduction
__device__ __forceinline__ uint32_t hash_range(uint32_t key, uint32_t size) {
    const uint32_t h = key * 2654435761u;
    return (uint32_t)(((uint64_t)h * size) >> 32);
}

}  // namespace bella
