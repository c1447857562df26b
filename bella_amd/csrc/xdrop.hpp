// xdrop.hpp -- Xavier-exact X-drop seed-and-extend on gfx950.
//
// Reference being reproduced bit for bit (scores, coordinates, quirks included):
//   include/align.hpp:152-202   xavierAlign   (strand decision, reverse-complemented copy of `row`)
//   xavier/xavier.h:276-374     XavierXDrop   (EXTEND_BOTH: reversed prefixes incl. seed, suffixes)
//   xavier/xavier.h:20-251      Phase1 (scalar 33x33 triangle), Phase2 (adaptive band), Phase4 (tail)
//   xavier/simdutils.h:22-337   int8 saturating vectors of width 32, shiftLeft/Right, moveRight/Down
//   include/overlap.hpp:413-497 PostAlignDecision (u16 overlap estimate, float threshold)
//
// Mapping (MI355X-first, not the reference's AVX2 shape and not LOGAN's block-per-alignment shape):
// the band is only 32 cells wide and one alignment is a strictly serial chain of ~10^4 dependent
// anti-diagonal steps, so spreading ONE band over 32 lanes would spend most issue slots on cross-lane
// traffic and max/arg-max reductions (about 60 wave instructions per step for 2 extensions).  Instead
// every LANE owns one extension and keeps the whole band in its own VGPRs (32 cells x 5 vectors), so a
// step is ~300-450 VALU instructions for 64 extensions, with no cross-lane traffic and no LDS in the
// loop.  The left and the right extension of a pair sit in adjacent lanes and are summed with one
// shuffle.  Sequences are read from the 2-bit packed reads through a one-word (16 base) lane cache;
// reversal and reverse-complement are index arithmetic, never materialised.
#pragma once
#include <stdint.h>
#include <math.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#endif
#include "../../include/bella_hip.h"
#include "core.hpp"

namespace bella {

constexpr int kXW = 32;        // VECTORWIDTH  (simdutils.h:22)
constexpr int kXLW = 31;       // LOGICALWIDTH (simdutils.h:23)
constexpr int kXNinf = -128;   // NINF         (simdutils.h:48)
constexpr int kXMiddle = 15;   // MIDDLE       (simdutils.h:51)
constexpr int kXCutoff = 102;  // CUTOFF       (simdutils.h:53)
constexpr int kCodeNul = 4;    // the std::string terminator copied by XavierState (simdutils.h:191-192)
constexpr int kCodePad = 5;    // the 32 NINF pad bytes (simdutils.h:194-195); NUL==NUL and pad==pad "match"

// A sequence as the extension sees it: element t = base (g0 + dir*t) of the packed array, complemented
// if comp==3; t == len is the NUL, t > len the padding.
struct SeqAcc {
    const uint32_t* packed;
    int64_t g0;
    int32_t dir;
    uint32_t comp;
    uint32_t len;
    uint32_t cw;
    int64_t cwi;
};
BELLA_HD int seq_code(SeqAcc& s, uint32_t t) {
    if (t >= s.len) return t == s.len ? kCodeNul : kCodePad;
    const int64_t g = s.g0 + (int64_t)s.dir * (int64_t)t;
    const int64_t wi = g >> 4;
    if (wi != s.cwi) { s.cw = s.packed[wi]; s.cwi = wi; }
    return (int)(((s.cw >> ((uint32_t)(g & 15) * 2)) & 3u) ^ s.comp);
}

struct XRes {
    int best;      // state.bestScore
    int endH, endV;  // state.seed end positions
    int flagged;
    int steps;
};

BELLA_HD int sat8(int x) { return x > 127 ? 127 : (x < -128 ? -128 : x); }
BELLA_HD int imax_(int a, int b) { return a > b ? a : b; }

// One anti-diagonal step (xavier.h:111-129 / :193-211) on the lane-private band.
#define BELLA_XSTEP()                                                                  \
    _Pragma("unroll") for (int e = 0; e < kXW; ++e) {                                 \
        const int mm = (vqh[e] == vqv[e]) ? 1 : -1;                                    \
        const int a1f = sat8(a1[e] + mm);                                              \
        const int sh = (e == kXW - 1) ? kXNinf : a2[e + 1 < kXW ? e + 1 : e];          \
        const int a2f = imax_(imax_(sh, a2[e]) - 1, -128);                             \
        a3[e] = imax_(a1f, a2f);                                                       \
    }                                                                                  \
    a3[kXLW] = kXNinf;

// max over the 32 cells and first index of the largest value (> 0 test done by the caller)
#define BELLA_XARGMAX(keyout)                                                          \
    {                                                                                  \
        int kk = a3[0] * 32 + 31;                                                    \
        _Pragma("unroll") for (int e = 1; e < kXW; ++e) kk = imax_(kk, a3[e] * 32 + (31 - e)); \
        keyout = kk;                                                                   \
    }

// xavier.h:257-274 XavierOneDirection on sequences h (len hn) and v (len vn), both >= 32 long.
// dp: lane-private scratch of 132 bytes addressed as dp[idx * dps].
template <class Byte>
BELLA_HD void xavier_one_direction(SeqAcc& H, SeqAcc& V, const int X, Byte* dp, const int dps, XRes& r) {
    int a1[kXW], a2[kXW], a3[kXW], vqh[kXW], vqv[kXW];
    const int hl = (int)H.len + 1, vl = (int)V.len + 1;       // hlength / vlength (simdutils.h:175-176)
    int hoff = kXLW, voff = kXLW;
    // ---- Phase 1 (xavier.h:20-103): scalar DP on the 33x33 upper-left triangle
    uint64_t hp = 0, vp = 0;
    for (int t = 0; t < 32; ++t) {
        hp |= (uint64_t)seq_code(H, (uint32_t)t) << (2 * t);
        vp |= (uint64_t)seq_code(V, (uint32_t)t) << (2 * t);
    }
    Byte* prev = dp;
    Byte* cur = dp + 34 * dps;
    Byte* ex1 = dp + 68 * dps;
    Byte* ex2 = dp + 100 * dps;
    for (int j = 0; j < 34; ++j) prev[j * dps] = (Byte)(-j);
    int DPmax = 0;
    for (int i = 1; i < kXLW + 2; ++i) {
        cur[0] = (Byte)(-i);
        const int hc = (int)((hp >> (2 * (i - 1))) & 3);
        int left = -i;
        for (int j = 1; j <= kXLW + 2 - i; ++j) {
            const int vc = (int)((vp >> (2 * (j - 1))) & 3);
            const int oneF = (int)prev[(j - 1) * dps] + (hc == vc ? 1 : -1);
            const int twoF = imax_((int)prev[j * dps], left) - 1;
            const int val = imax_(oneF, twoF);
            cur[j * dps] = (Byte)val;
            left = val;
            DPmax = imax_(DPmax, val);
        }
        if (i <= kXLW) ex1[(i - 1) * dps] = cur[(kXLW + 1 - i) * dps];     // antiDiag1[i-1] = DP[i][32-i]
        if (i >= 2) ex2[(i - 1) * dps] = cur[(kXLW + 2 - i) * dps];        // antiDiag2[i-1] = DP[i][33-i]
        Byte* tmp = prev; prev = cur; cur = tmp;
    }
    int adm = -128;
#pragma unroll
    for (int e = 0; e < kXW; ++e) {
        a1[e] = e < kXLW ? (int)ex1[e * dps] : kXNinf;
        a2[e] = e >= 1 ? (int)ex2[e * dps] : kXNinf;
        a3[e] = kXNinf;
        if (e < kXLW) adm = imax_(adm, a1[e]);
    }
    // vqueryh[i] = queryh[i+1], vqueryv[i] = queryv[31-i], i < 31 (xavier.h:61-68)
#pragma unroll
    for (int e = 0; e < kXW; ++e) {
        vqh[e] = e < kXLW ? (int)((hp >> (2 * (e + 1 < 32 ? e + 1 : 31))) & 3) : kXNinf;
        vqv[e] = e < kXLW ? (int)((vp >> (2 * (kXLW - e))) & 3) : kXNinf;
    }
    int best = DPmax;
    int off = 0;
    r.flagged = 0;
    if (adm < DPmax - X) { r.best = best; r.endH = hoff; r.endV = voff; r.steps = 0; return; }   // :91-99

    // ---- Phase 2 (xavier.h:105-183)
    int maxpos = 0;        // uninitialised in the reference (:165); 0 + flag when it matters
    int endH = hoff, endV = voff;
    bool first = true;
    while (hoff < hl && voff < vl) {
        BELLA_XSTEP()
        int key;
        BELLA_XARGMAX(key)
        const int adb = key >> 5;
        const int curr = adb + off;                                             // :135
        if (curr < best - X) {                                                  // :137-150
            r.best = best; r.endH = hoff; r.endV = voff; r.steps = (hoff - kXLW) + (voff - kXLW);
            return;
        }
        if (adb > kXCutoff) {                                                   // :152-158
            int mn = a3[0];
#pragma unroll
            for (int e = 1; e < kXLW; ++e) mn = a3[e] < mn ? a3[e] : mn;
#pragma unroll
            for (int e = 0; e < kXW; ++e) { a2[e] = sat8(a2[e] - mn); a3[e] = sat8(a3[e] - mn); }
            off += mn;
            BELLA_XARGMAX(key)
        }
        if (curr > best) best = curr;                                           // :161-162
        if ((key >> 5) > 0) maxpos = 31 - (key & 31);                           // :165-173 first largest value > 0
        else if (first) r.flagged = 1;
        first = false;
        endH = hoff; endV = voff;                                               // :175-176
        if (maxpos > kXMiddle) {                                                // moveRight, simdutils.h:263-274
            const int c = seq_code(H, (uint32_t)hoff);
            hoff++;
#pragma unroll
            for (int e = 0; e < kXW - 1; ++e) { vqh[e] = vqh[e + 1]; a1[e] = a2[e + 1]; }
            vqh[kXW - 1] = kXNinf; a1[kXW - 1] = kXNinf;
            vqh[kXLW - 1] = c;
#pragma unroll
            for (int e = 0; e < kXW; ++e) a2[e] = a3[e];
        } else {                                                                // moveDown, simdutils.h:276-289
            const int c = seq_code(V, (uint32_t)voff);
            voff++;
#pragma unroll
            for (int e = kXW - 1; e > 0; --e) vqv[e] = vqv[e - 1];
            vqv[0] = c;
#pragma unroll
            for (int e = 0; e < kXW; ++e) a1[e] = a2[e];
#pragma unroll
            for (int e = kXW - 1; e > 0; --e) a2[e] = a3[e - 1];
            a2[0] = kXNinf;
        }
    }
    // ---- Phase 4 (xavier.h:185-251): 28 more steps, alternating directions; end position not updated
    int dir = hoff >= hl ? 1 : 0;                                               // goDOWN = 1, goRIGHT = 0
    for (int it = 0; it < kXLW - 3; ++it) {
        BELLA_XSTEP()
        int key;
        BELLA_XARGMAX(key)
        const int adb = key >> 5;
        const int curr = adb + off;
        if (curr < best - X) break;                                             // :219-223
        if (adb > kXCutoff) {
            int mn = a3[0];
#pragma unroll
            for (int e = 1; e < kXLW; ++e) mn = a3[e] < mn ? a3[e] : mn;
#pragma unroll
            for (int e = 0; e < kXW; ++e) { a2[e] = sat8(a2[e] - mn); a3[e] = sat8(a3[e] - mn); }
            off += mn;
        }
        if (curr > best) best = curr;
        const int next = dir ^ 1;
        if (next == 0) {
            const int c = seq_code(H, (uint32_t)hoff);
            hoff++;
#pragma unroll
            for (int e = 0; e < kXW - 1; ++e) { vqh[e] = vqh[e + 1]; a1[e] = a2[e + 1]; }
            vqh[kXW - 1] = kXNinf; a1[kXW - 1] = kXNinf;
            vqh[kXLW - 1] = c;
#pragma unroll
            for (int e = 0; e < kXW; ++e) a2[e] = a3[e];
        } else {
            const int c = seq_code(V, (uint32_t)voff);
            voff++;
#pragma unroll
            for (int e = kXW - 1; e > 0; --e) vqv[e] = vqv[e - 1];
            vqv[0] = c;
#pragma unroll
            for (int e = 0; e < kXW; ++e) a1[e] = a2[e];
#pragma unroll
            for (int e = kXW - 1; e > 0; --e) a2[e] = a3[e - 1];
            a2[0] = kXNinf;
        }
        dir = next;
    }
    r.best = best; r.endH = endH; r.endV = endV; r.steps = (hoff - kXLW) + (voff - kXLW);
}

// Geometry of the two extensions of one seed (align.hpp:152-202 + xavier.h:325-373).
struct PairGeom {
    uint32_t lenH, lenV;
    int bH, eH, bV, eV;     // seed on the oriented H and on V
    uint32_t strand;        // 1 = "c"
};
BELLA_HD void make_geom(const uint32_t* packed, uint64_t goffH, uint32_t lenH, uint64_t goffV, uint32_t lenV, uint32_t seedH,
                        uint32_t seedV, uint32_t k, PairGeom& g) {
    const uint64_t leH = kmer_le(packed, goffH + seedH, k), leV = kmer_le(packed, goffV + seedV, k);
    g.strand = kmer_rc_from_le(leH, k) == kmer_fw_from_le(leV, k) ? 1u : 0u;          // align.hpp:168-176
    g.lenH = lenH; g.lenV = lenV;
    g.bH = g.strand ? (int)lenH - (int)seedH - (int)k : (int)seedH;                    // :181-182
    g.eH = g.bH + (int)k;
    g.bV = (int)seedV; g.eV = (int)seedV + (int)k;
}
// which = 0: left (reversed prefixes INCLUDING the seed, xavier.h:330-334); 1: right (suffixes AFTER it, :351-352)
BELLA_HD void make_accessors(const uint32_t* packed, uint64_t goffH, uint64_t goffV, const PairGeom& g, int which, SeqAcc& H,
                             SeqAcc& V) {
    H.packed = packed; V.packed = packed;
    H.cwi = -1; V.cwi = -1; H.cw = 0; V.cw = 0;
    H.comp = g.strand ? 3u : 0u; V.comp = 0u;
    if (which == 0) {
        H.len = (uint32_t)g.eH; V.len = (uint32_t)g.eV;
        if (!g.strand) { H.g0 = (int64_t)goffH + g.eH - 1; H.dir = -1; }
        else { H.g0 = (int64_t)goffH + (int64_t)g.lenH - g.eH; H.dir = 1; }            // revcomp(row)[eH-1-t]
        V.g0 = (int64_t)goffV + g.eV - 1; V.dir = -1;
    } else {
        H.len = g.lenH - (uint32_t)g.eH; V.len = g.lenV - (uint32_t)g.eV;
        if (!g.strand) { H.g0 = (int64_t)goffH + g.eH; H.dir = 1; }
        else { H.g0 = (int64_t)goffH + (int64_t)g.lenH - 1 - g.eH; H.dir = -1; }       // revcomp(row)[eH+t]
        V.g0 = (int64_t)goffV + g.eV; V.dir = 1;
    }
}

// xavier.h:325-373 bookkeeping + simdutils.h:333-337 + overlap.hpp:413-460
BELLA_HD void finish_pair(const PairGeom& g, bool ranL, const XRes& L, bool ranR, const XRes& R, double ratiophi, double delta,
                          bella_aln& out) {
    int bH, bV, eH = g.eH, eV = g.eV;
    int score = 0;
    if (ranL) { bH = g.eH - L.endH; bV = g.eV - L.endV; score += L.best; }
    else { bH = 0; bV = 0; }                                                           // :338-342
    if (ranR) { eH += R.endH; eV += R.endV; score += R.best; }
    else { bH = g.eH + (int)(g.lenH - (uint32_t)g.eH); bV = g.eV + (int)(g.lenV - (uint32_t)g.eV); }   // :356-360 (sic: begin)
    out.score = score; out.begH = bH; out.endH = eH; out.begV = bV; out.endV = eV;
    out.strand = (uint8_t)g.strand;
    out.steps = (uint32_t)((ranL ? L.steps : 0) + (ranR ? R.steps : 0));
    out.flagged = (uint32_t)((ranL ? L.flagged : 0) | (ranR ? R.flagged : 0));
    // PostAlignDecision, overlap.hpp:441-460 (u16 arithmetic, float compare)
    const uint16_t read1len = (uint16_t)g.lenH, read2len = (uint16_t)g.lenV;
    const uint16_t olV = (uint16_t)(eV - bV), olH = (uint16_t)(eH - bH);
    const uint16_t minLeft = (uint16_t)(bV < bH ? bV : bH);
    const int r2 = (int)read2len - eV, r1 = (int)read1len - eH;
    const uint16_t minRight = (uint16_t)(r2 < r1 ? r2 : r1);
    const uint16_t ov = (uint16_t)(minLeft + minRight + ((int)olV + (int)olH) / 2);
    const float thr = (float)((1 - delta) * (ratiophi * (double)(float)ov));
    out.ov = ov;
    out.passed = ((float)score >= thr) ? 1 : 0;
}

#if defined(__HIPCC__)
struct XdropArgs {
    const bella_seed* seeds;   // either seeds ...
    const bella_pair* pairs;   // ... or pair records
    uint64_t n;
    const uint32_t* packed;
    const uint64_t* roff;
    uint32_t k;
    int xdrop;
    double ratiophi, delta;
    bella_aln* out;
};
constexpr int kXdropBlock = 256;
constexpr int kXdropPairsPerBlock = kXdropBlock / 2;

__global__ __launch_bounds__(kXdropBlock) void k_xdrop(XdropArgs a) {
    __shared__ int8_t dp[132 * kXdropBlock];
    const uint64_t p = (uint64_t)blockIdx.x * kXdropPairsPerBlock + (threadIdx.x >> 1);
    const int which = threadIdx.x & 1;
    const bool valid = p < a.n;
    uint32_t rid = 0, cid = 0, seedH = 0, seedV = 0;
    if (valid) {
        if (a.seeds) { const bella_seed s = a.seeds[p]; rid = s.rid; cid = s.cid; seedH = s.seedH; seedV = s.seedV; }
        else { const bella_pair s = a.pairs[p]; rid = s.rid; cid = s.cid; seedH = s.seedH; seedV = s.seedV; }
    }
    PairGeom g;
    XRes res;
    res.best = 0; res.endH = 0; res.endV = 0; res.flagged = 0; res.steps = 0;
    bool ran = false;
    uint64_t goffH = 0, goffV = 0;
    if (valid) {
        goffH = a.roff[rid]; goffV = a.roff[cid];
        make_geom(a.packed, goffH, (uint32_t)(a.roff[rid + 1] - goffH), goffV, (uint32_t)(a.roff[cid + 1] - goffV), seedH, seedV,
                  a.k, g);
        SeqAcc H, V;
        make_accessors(a.packed, goffH, goffV, g, which, H, V);
        if (H.len >= (uint32_t)kXW && V.len >= (uint32_t)kXW) {
            ran = true;
            xavier_one_direction(H, V, a.xdrop, dp + threadIdx.x, kXdropBlock, res);
        }
    }
    // the right extension (odd lane) hands its result to the even lane
    XRes o;
    o.best = __shfl_down(res.best, 1, 64);
    o.endH = __shfl_down(res.endH, 1, 64);
    o.endV = __shfl_down(res.endV, 1, 64);
    o.flagged = __shfl_down(res.flagged, 1, 64);
    o.steps = __shfl_down(res.steps, 1, 64);
    const int oran = __shfl_down((int)ran, 1, 64);
    if (valid && which == 0) {
        bella_aln out;
        finish_pair(g, ran, res, oran != 0, o, a.ratiophi, a.delta, out);
        a.out[p] = out;
    }
}

__global__ void k_count_passed(const bella_aln* alns, uint64_t n, unsigned long long* cnt) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int pass = (i < n && alns[i].passed) ? 1 : 0;
    const unsigned long long m = __ballot(pass);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(cnt, (unsigned long long)__popcll(m));
}
#endif

}  // namespace bella
