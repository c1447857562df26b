// slotorder.hpp -- the layout sequential linear-probing insertion produces, computed in parallel (gfx950).
//
// The reference fills per-column / per-read open-addressing tables of 2^n >= max(16, items) slots, hash key * 107, linear probing,
// one item after the other (src/CSC.cpp:316-375 MergeDuplicates; include/overlap.hpp:289-361 LocalSpGEMM), and reads them out in
// slot order: that order is part of the output.  With every item carrying its insertion time, the layout is the fixed point of
// parallel atomicMin insertion of (time << 16 | id): an earlier item displaces a later one, which resumes probing
// (tests/test_core_host.py proves the fixed point under random interleavings).  At the load factors the reference's tables have
// (up to 1.0) the late items travel hundreds of slots, one dependent LDS round trip each.  So the items go in ROUNDS by insertion
// time, every round at most about half of the slots still free: what earlier rounds placed is final (a later item never displaces
// an earlier one), a round's items only compete for the free slots, and a next-free table built at the start of the round takes them
// past the final entries in one step -- the same insertion on the table compressed to its free slots, at load <= ~1/2 every round.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "core.hpp"
#include "util.hpp"

namespace bella {

// a "group" is one wavefront (NT == 64: its LDS operations execute in order, no barrier) or a workgroup of NT threads
template <int NT>
__device__ __forceinline__ void group_sync() {
    if (NT == 64) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // keeps the compiler's order; one wavefront's LDS operations are in order
    else __syncthreads();
}
template <int NT>
__device__ __forceinline__ uint32_t group_tid() { return NT == 64 ? lane_id() : threadIdx.x; }

__device__ __forceinline__ uint32_t umin_(uint32_t a, uint32_t b) { return a < b ? a : b; }

// inclusive prefix-min over the 64 lanes (DPP, the pattern of wave_incl_scan with min; lanes without a source keep their value)
__device__ __forceinline__ uint32_t wave_incl_prefix_min(uint32_t v) {
    v = umin_(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x111, 0xF, 0xF, false));
    v = umin_(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x112, 0xF, 0xF, false));
    v = umin_(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x114, 0xF, 0xF, false));
    v = umin_(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x118, 0xF, 0xF, false));
    v = umin_(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x142, 0xA, 0xF, false));
    v = umin_(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x143, 0xC, 0xF, false));
    return v;
}

// min of v over the group's threads with a LARGER index (0xFFFFFFFF if none); *all = min over all threads.
// scr: 2 * (NT / 64) words of LDS (unused for one wavefront).  Workgroups: two barriers inside.
template <int NT>
__device__ __forceinline__ uint32_t group_suffix_min_excl(uint32_t v, uint32_t* scr, uint32_t* all) {
    const uint32_t lane = lane_id();
    const uint32_t rev = (uint32_t)__shfl((int)v, 63 - (int)lane, 64);            // lane l holds thread 63-l: suffixes become prefixes
    const uint32_t inc = wave_incl_prefix_min(rev);
    uint32_t exl = (uint32_t)__shfl_up((int)inc, 1, 64);                          // exclusive (lane 0: nothing to its right)
    if (lane == 0) exl = 0xFFFFFFFFu;
    uint32_t res = (uint32_t)__shfl((int)exl, 63 - (int)lane, 64);
    const uint32_t wtot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    if (NT == 64) { *all = wtot; return res; }
    constexpr int NW = NT / 64;
    const uint32_t w = wave_id();
    if (lane == 0) scr[w] = wtot;
    __syncthreads();
    uint32_t right = 0xFFFFFFFFu, tot = 0xFFFFFFFFu;
#pragma unroll
    for (int x = 0; x < NW; ++x) {
        const uint32_t t = scr[x];
        tot = umin_(tot, t);
        if ((uint32_t)x > w) right = umin_(right, t);
    }
    __syncthreads();
    *all = tot;
    return umin_(res, right);
}

// nf[s] = the first slot at or after s (cyclically) that is empty in T2.  Needs an empty slot (returns false otherwise).
template <int NT, class NF>
__device__ __forceinline__ bool build_next_free(const uint32_t* T2, NF* nf, uint32_t ht, uint32_t* scr) {
    const uint32_t tid = group_tid<NT>();
    const uint32_t c = (ht + NT - 1) / NT;
    const uint32_t lo = tid * c < ht ? tid * c : ht, hi = lo + c < ht ? lo + c : ht;
    uint32_t ff = 0xFFFFFFFFu;
    for (uint32_t s = hi; s-- > lo;) ff = T2[s] == kEmpty ? s : ff;
    uint32_t all;
    uint32_t cur = group_suffix_min_excl<NT>(ff, scr, &all);
    if (all == 0xFFFFFFFFu) return false;
    if (cur == 0xFFFFFFFFu) cur = all;                                           // wraps around to the first free slot of the table
    for (uint32_t s = hi; s-- > lo;) {
        if (T2[s] == kEmpty) cur = s;
        nf[s] = (NF)cur;
    }
    return true;
}

// time bound of round r: the rounds take the items in the order of their insertion times, round r up to the time below which
// about ht * (1 - 2^-(r+2)) items lie if the d items' times are spread evenly over [0, tmax): the first round fills the empty table
// to 3/4 (plain probing is short up to there; most tables need no second round), every later round half of the slots still free
__device__ __forceinline__ uint32_t round_bound(uint32_t r, uint32_t ht, uint32_t d, uint32_t tmax) {
    const uint32_t keep = r + 2 < 32 ? ht >> (r + 2) : 0u;
    if (keep < 32u) return tmax;                                     // the last few dozen items go in one round
    const uint64_t b = (uint64_t)tmax * (ht - keep) / (d ? d : 1u);
    return b >= tmax ? tmax : (uint32_t)b;
}

// one item into the table: plain probing (first round: empty table) or past the final entries through nf
template <bool JUMP, class NF>
__device__ __forceinline__ void slot_insert(uint32_t* T2, const NF* nf, uint32_t mask, uint32_t home, uint32_t item) {
    uint32_t s = JUMP ? (uint32_t)nf[home] : home;
    for (;;) {
        const uint32_t old = atomicMin(&T2[s], item);
        if (old == kEmpty) break;
        if (old > item) item = old;                              // we took the slot; the displaced entry resumes probing
        s = (s + 1) & mask;
        if (JUMP) s = (uint32_t)nf[s];
    }
}

}  // namespace bella
