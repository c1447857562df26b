// logan.hpp -- the exact (growing band) gapped X-drop of the reference's CUDA build, as a second scoring kernel for gfx950.
//
// Reference behaviour reproduced (scores and coordinates; SeqAn's extendSeed(GappedXDrop), which the CUDA kernel ports, is the
// arbiter and the oracle's pin):
//   loganGPU/functions.cuh:223-408   extendSeedLGappedXDropOneDirectionGlobal (three rolling anti-diagonals, cells below
//                                    best - X become UNDEF, the band [minCol, maxCol) follows the defined cells)
//   loganGPU/functions.cuh:505-547   prefixes / suffixes EXCLUDING the seed; :680-682 score = left + right + k
//   include/overlap.hpp:918-944      strand test, reverse complement of the H read, seed remap (as xavierAlign does)
//   include/overlap.hpp:797-871      PostAlignDecisionGPU (threshold test in double)
// Not reproduced: the CUDA kernel returns without writing its score when a prefix / suffix is empty (functions.cuh:270-271); SeqAn
// returns 0 there and so does this kernel.
//
// Mapping: nothing of the CUDA shape (32-thread blocks, anti-diagonals in global memory).  One wavefront per extension, the three
// anti-diagonals as int16 rings in LDS (the band of an X-drop with X = 7 is a few dozen cells wide: one 64-lane sweep per
// anti-diagonal), sequences read from the 2-bit packed reads through per-lane one-word caches (reversal and reverse complement
// are index arithmetic), the band's maximum by a DPP reduction, the band shrink by ballots.  Extensions whose band outgrows the
// LDS rings (kLoganCap cells) are redone by a second launch on rings in HBM.
#pragma once
#include <hip/hip_runtime.h>
#include "xdrop.hpp"

namespace bella {

constexpr int kLgUndef = -32767;       // UNDEF, functions.cuh:18
constexpr int kLgGap = -1;             // GAP_EXT, functions.cuh:16
constexpr int kLoganCap = 2048;        // cells per anti-diagonal ring in LDS

struct LoganArgs {
    const bella_seed* seeds;
    const bella_pair* pairs;
    uint64_t n;                        // pairs; extension e = 2 * pair + (0: left, 1: right)
    const uint32_t* packed;
    const uint64_t* roff;
    uint32_t k;
    int xdrop;
    double ratiophi, delta;
    int4* res;                         // [2n] {score, extCol (query bases), extRow (database bases), anti-diagonals}; score == INT_MIN: redo
    bella_aln* out;
    uint32_t* redo;                    // [2n] extensions whose band outgrew the LDS rings
    uint32_t* nredo;
    int16_t* scratch;                  // rings in HBM for the second launch: 3 * scratch_cap per wavefront
    uint32_t scratch_cap;
    uint64_t ebase;                    // first extension of this launch (k_logan_lds is launched in chunks: grid x block < 2^32)
    uint32_t* status;                  // bit 9: an extension outgrew even the rings in HBM (cannot happen for reads < 65536 bases)
};

// the two sequences of an extension in READING order (element t of the query = column t + 1, of the database = row t + 1)
__device__ __forceinline__ void logan_accessors(const uint32_t* packed, uint64_t goffH, uint64_t goffV, const PairGeom& g, int which, SeqAcc& Q,
                                                SeqAcc& D) {
    Q.packed = packed; D.packed = packed;
    Q.cwi = -1; D.cwi = -1; Q.cw = 0; D.cw = 0;
    Q.comp = 0u; D.comp = g.strand ? 3u : 0u;
    if (which == 0) {                                                 // left: both prefixes backwards from the seed (functions.cuh:539-541, :119-120)
        Q.len = (uint32_t)g.bV; D.len = (uint32_t)g.bH;
        Q.g0 = (int64_t)goffV + g.bV - 1; Q.dir = -1;
        if (!g.strand) { D.g0 = (int64_t)goffH + g.bH - 1; D.dir = -1; }
        else { D.g0 = (int64_t)goffH + (int64_t)g.lenH - g.bH; D.dir = 1; }           // revcomp(row)[bH - 1 - t]
    } else {                                                          // right: both suffixes forwards (:542-543)
        Q.len = g.lenV - (uint32_t)g.eV; D.len = g.lenH - (uint32_t)g.eH;
        Q.g0 = (int64_t)goffV + g.eV; Q.dir = 1;
        if (!g.strand) { D.g0 = (int64_t)goffH + g.eH; D.dir = 1; }
        else { D.g0 = (int64_t)goffH + (int64_t)g.lenH - 1 - g.eH; D.dir = -1; }      // revcomp(row)[eH + t]
    }
}
__device__ __forceinline__ int logan_base(SeqAcc& s, int t) {         // t < len always
    const int64_t g = s.g0 + (int64_t)s.dir * (int64_t)t;
    const int64_t wi = g >> 4;
    if (wi != s.cwi) { s.cw = s.packed[wi]; s.cwi = wi; }
    return (int)(((s.cw >> ((uint32_t)(g & 15) * 2)) & 3u) ^ s.comp);
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { const int o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
    return v;
}

// RING: a pointer-like accessor for the three rings (LDS: plain int16 loads/stores; HBM: loads/stores that bypass the per-CU L1,
// because the lanes of the wavefront read cells their neighbours wrote)
struct RingLds {
    int16_t* p;
    __device__ __forceinline__ int get(int i) const { return (int)p[i]; }
    __device__ __forceinline__ void set(int i, int v) const { p[i] = (int16_t)v; }
};
struct RingHbm {
    int16_t* p;
    __device__ __forceinline__ int get(int i) const { return (int)__hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ void set(int i, int v) const { __hip_atomic_store(p + i, (int16_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
};

// functions.cuh:223-408 on one wavefront.  Returns false if an anti-diagonal needs more than `cap` cells.
template <class Ring>
__device__ __forceinline__ bool logan_one_direction(SeqAcc& Q, SeqAcc& D, const int X, Ring r1, Ring r2, Ring r3, const int cap, int4& out) {
    const int lane = (int)(threadIdx.x & 63u);
    const int cols = (int)Q.len + 1, rows = (int)D.len + 1;            // :260-267
    out = make_int4(0, 0, 0, 0);
    if (rows == 1 || cols == 1) return true;                           // :270 (SeqAn: 0)
    int a1size = 0, a2size = 1, a3size = 2;                            // initAntiDiags :186-216
    int minCol = 1, maxCol = 2, offset1 = 0, offset2 = 0, offset3 = 0;
    if (lane == 0) { r2.set(0, 0); r3.set(0, kLgGap); r3.set(1, kLgGap); }
    int antiDiagNo = 1, best = 0;
    __syncthreads();
    while (minCol < maxCol) {                                           // :290
        ++antiDiagNo;
        const Ring t = r1; r1 = r2; r2 = r3; r3 = t;                    // :299-311
        const int tl = a1size; a1size = a2size; a2size = a3size; a3size = tl;
        offset1 = offset2; offset2 = offset3; offset3 = minCol - 1;
        a3size = maxCol + 1 - offset3;                                  // initAntiDiag3 :160-184
        if (a3size > cap) return false;
        int mx = kLgUndef;
        if (lane == 0) {
            int first = kLgUndef, last = kLgUndef;
            if (antiDiagNo * kLgGap > best - X) {
                if (offset3 == 0) first = antiDiagNo * kLgGap;
                if (antiDiagNo - maxCol == 0) last = antiDiagNo * kLgGap;
            }
            r3.set(0, first);
            r3.set(maxCol - offset3, last);
            mx = first > last ? first : last;
        }
        for (int c0 = minCol; c0 < maxCol; c0 += 64) {                  // computeAntidiag :102-148
            const int col = c0 + lane;
            if (col < maxCol) {
                const int a = r2.get(col - offset2), b = r2.get(col - offset2 - 1);
                int tmp = (a > b ? a : b) + kLgGap;
                const int sc = logan_base(Q, col - 1) == logan_base(D, antiDiagNo - col - 1) ? 1 : -1;   // database in reading order: row - 1
                const int dg = r1.get(col - offset1 - 1) + sc;
                tmp = dg > tmp ? dg : tmp;
                tmp = tmp < best - X ? kLgUndef : tmp;
                r3.set(col - minCol + 1, tmp);
                mx = tmp > mx ? tmp : mx;
            }
        }
        const int adb = wave_max_i32(mx);                               // :318-333
        best = adb > best ? adb : best;
        __syncthreads();
        // :339-343 leading cells that are UNDEF on this and on the previous anti-diagonal leave the band
        for (;;) {
            const int c = minCol + lane;
            const bool gone = c - offset3 < a3size && r3.get(c - offset3) == kLgUndef && c - offset2 - 1 < a2size && r2.get(c - offset2 - 1) == kLgUndef;
            const unsigned long long m = __ballot(gone);
            const int lead = m == ~0ull ? 64 : __builtin_ctzll(~m);
            minCol += lead;
            if (lead < 64) break;
        }
        // :346-350 and trailing ones
        for (;;) {
            const int c = maxCol - lane;                                // candidates maxCol, maxCol - 1, ...
            const bool gone = c - offset3 > 0 && r3.get(c - offset3 - 1) == kLgUndef && r2.get(c - offset2 - 1) == kLgUndef;
            const unsigned long long m = __ballot(gone);
            const int lead = m == ~0ull ? 64 : __builtin_ctzll(~m);
            maxCol -= lead;
            if (lead < 64) break;
        }
        ++maxCol;
        if (minCol < antiDiagNo + 2 - rows) minCol = antiDiagNo + 2 - rows;   // :358
        if (maxCol > cols) maxCol = cols;                                     // :360
        __syncthreads();
    }
    int lcol = a3size + offset3 - 2;                                    // :364-366
    int lrow = antiDiagNo - lcol;
    int lscore = r3.get(lcol - offset3);
    if (lscore == kLgUndef) {
        if (r2.get(a2size - 2) != kLgUndef) {                           // :370-376
            lcol = a2size + offset2 - 2; lrow = antiDiagNo - 1 - lcol; lscore = r2.get(lcol - offset2);
        } else if (a2size > 2 && r2.get(a2size - 3) != kLgUndef) {      // :378-384
            lcol = a2size + offset2 - 3; lrow = antiDiagNo - 1 - lcol; lscore = r2.get(lcol - offset2);
        }
    }
    if (lscore == kLgUndef) {                                           // :389-401: the FIRST maximum of anti-diagonal 1
        int bestv = kLgUndef, besti = 0x7FFFFFFF;
        for (int x0 = 0; x0 < a1size; x0 += 64) {
            const int x = x0 + lane;
            const int v = x < a1size ? r1.get(x) : kLgUndef;
            const int wm = wave_max_i32(v);
            if (wm > bestv) {
                const unsigned long long m = __ballot(v == wm);
                bestv = wm; besti = x0 + __builtin_ctzll(m);
            }
        }
        if (bestv > lscore) { lscore = bestv; lcol = besti + offset1; lrow = antiDiagNo - 2 - lcol; }
    }
    out = lscore != kLgUndef ? make_int4(lscore, lcol, lrow, antiDiagNo - 1) : make_int4(lscore, 0, 0, antiDiagNo - 1);   // :403-404
    return true;
}

__device__ __forceinline__ bool logan_setup(const LoganArgs& a, uint64_t e, PairGeom& g, SeqAcc& Q, SeqAcc& D) {
    const uint64_t p = e >> 1;
    uint32_t rid, cid, seedH, seedV;
    if (a.seeds) { const bella_seed s = a.seeds[p]; rid = s.rid; cid = s.cid; seedH = s.seedH; seedV = s.seedV; }
    else { const bella_pair s = a.pairs[p]; rid = s.rid; cid = s.cid; seedH = s.seedH; seedV = s.seedV; }
    const uint64_t goffH = a.roff[rid], goffV = a.roff[cid];
    make_geom(a.packed, goffH, (uint32_t)(a.roff[rid + 1] - goffH), goffV, (uint32_t)(a.roff[cid + 1] - goffV), seedH, seedV, a.k, g);
    logan_accessors(a.packed, goffH, goffV, g, (int)(e & 1), Q, D);
    return true;
}

// first launch: one extension per 64-thread workgroup (the barriers inside are single-wavefront barriers), rings in LDS
__global__ __launch_bounds__(64) void k_logan_lds(LoganArgs a) {
    __shared__ int16_t rings[3][kLoganCap];
    const uint64_t e = a.ebase + blockIdx.x;
    PairGeom g;
    SeqAcc Q, D;
    logan_setup(a, e, g, Q, D);
    int4 res;
    const bool ok = logan_one_direction(Q, D, a.xdrop, RingLds{rings[0]}, RingLds{rings[1]}, RingLds{rings[2]}, kLoganCap, res);
    if (threadIdx.x == 0) {
        if (ok) a.res[e] = res;
        else { a.res[e] = make_int4((int)0x80000000, 0, 0, 0); a.redo[atomicAdd(a.nredo, 1u)] = (uint32_t)e; }
    }
}

// second launch: the extensions whose band outgrew the LDS rings, on rings in HBM (one 64-thread workgroup each, persistent)
__global__ __launch_bounds__(64) void k_logan_hbm(LoganArgs a) {
    const uint32_t n = *a.nredo;
    int16_t* base = a.scratch + (size_t)blockIdx.x * 3 * a.scratch_cap;
    for (uint32_t x = blockIdx.x; x < n; x += gridDim.x) {
        const uint64_t e = a.redo[x];
        PairGeom g;
        SeqAcc Q, D;
        logan_setup(a, e, g, Q, D);
        int4 res;
        const bool ok = logan_one_direction(Q, D, a.xdrop, RingHbm{base}, RingHbm{base + a.scratch_cap}, RingHbm{base + 2 * a.scratch_cap}, (int)a.scratch_cap, res);
        if (threadIdx.x == 0) { a.res[e] = res; if (!ok) atomicOr(a.status, 512u); }
    }
}

// combine the two extensions of a pair: functions.cuh:680-682, updateExtendedSeedL :67-100, PostAlignDecisionGPU overlap.hpp:797-871
__global__ void k_logan_finish(LoganArgs a) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.n) return;
    PairGeom g;
    SeqAcc Q, D;
    logan_setup(a, 2 * p, g, Q, D);
    const int4 L = a.res[2 * p], R = a.res[2 * p + 1];
    bella_aln out;
    out.score = L.x + R.x + (int)a.k;
    out.begH = g.bH - L.z; out.begV = g.bV - L.y;
    out.endH = g.eH + R.z; out.endV = g.eV + R.y;
    out.strand = (uint8_t)g.strand;
    out.steps = (uint32_t)(L.w + R.w);
    out.flagged = 0;
    const uint16_t read1len = (uint16_t)g.lenH, read2len = (uint16_t)g.lenV;            // :814-823
    const uint16_t olV = (uint16_t)(out.endV - out.begV), olH = (uint16_t)(out.endH - out.begH);
    const uint16_t minLeft = (uint16_t)(out.begV < out.begH ? out.begV : out.begH);
    const int r2 = (int)read2len - out.endV, r1 = (int)read1len - out.endH;
    const uint16_t minRight = (uint16_t)(r2 < r1 ? r2 : r1);
    const uint16_t ov = (uint16_t)(minLeft + minRight + ((int)olV + (int)olH) / 2);
    const double thr = (1 - a.delta) * (a.ratiophi * (double)ov);                        // :830 (double, not float)
    out.ov = ov;
    out.passed = ((double)out.score >= thr) ? 1 : 0;
    a.out[p] = out;
}

}  // namespace bella
