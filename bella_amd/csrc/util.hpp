// util.hpp -- block/wave primitives shared by the gfx950 kernels (wave = 64 lanes, block = 256 threads).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bella {

constexpr int kBlock = 256;        // 4 wavefronts
constexpr int kWaves = kBlock / 64;

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint32_t wave_id() { return threadIdx.x >> 6; }

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if ((int)lane_id() >= d) v += t;
    }
    return v;
}

// Exclusive scan over the 256 threads of a block; *total receives the block sum.
// `scr` must hold NW u32 in LDS (NW = wavefronts per block).  Contains two barriers; safe to call back to back.
template <int NW = kWaves>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* scr, uint32_t* total) {
    const uint32_t inc = wave_incl_scan(v);
    if (lane_id() == 63) scr[wave_id()] = inc;
    __syncthreads();
    uint32_t base = 0, sum = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const uint32_t s = scr[w];
        if (w < (int)wave_id()) base += s;
        sum += s;
    }
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
