// xdrop_packed.hpp -- the production X-drop lane function (device only): same semantics as xdrop.hpp's
// xavier_one_direction (which stays as the readable, host-testable statement of xavier/xavier.h:20-251), restructured for
// gfx950 issue efficiency:
//   * two band cells per VGPR as packed i16 (v_pk_add/max/min_i16, v_alignbit for the one-cell shifts): the five 32-cell
//     vectors take 80 VGPRs instead of 160, so several wavefronts fit a SIMD;
//   * moveRight / moveDown (simdutils.h:263-289) are per-lane selects, not branches: a wavefront whose lanes disagree does
//     not execute both variants;
//   * sequences come from a 64-base register window per stream, topped up at wave-uniform checkpoints (every 16 steps, all
//     lanes at once, the next block prefetched one period ahead): no lane ever makes the wavefront wait on its own reload.
// int8 saturation (simdutils.h:26-27, AVX2 adds/subs) is emulated exactly with one instruction per add: a cell value v lives
// in its 16-bit half as v*256, so v_pk_add_i16 with the clamp bit saturates at -128*256 exactly; the upper bound
// (127*256 + 255 after a clamp) is restored by one v_pk_min_i16 against 127*256 where a positive term can be added.
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#include "xdrop.hpp"

namespace bella {

typedef short s2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ s2 mk2(int a, int b) { s2 r; r.x = (short)a; r.y = (short)b; return r; }
__device__ __forceinline__ s2 splat2(int a) { return mk2(a, a); }
__device__ __forceinline__ uint32_t u32_of(s2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ s2 s2_of(uint32_t v) { return __builtin_bit_cast(s2, v); }
__device__ __forceinline__ s2 pmax(s2 a, s2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ s2 pmin(s2 a, s2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ s2 adds2(s2 a, s2 b) { return __builtin_elementwise_add_sat(a, b); }   // v_pk_add_i16 clamp
__device__ __forceinline__ s2 subs2(s2 a, s2 b) { return __builtin_elementwise_sub_sat(a, b); }   // v_pk_sub_i16 clamp
constexpr int kXScale = 256;            // cell value v is held as v * 256
constexpr int kXTop = 127 * kXScale;
constexpr int kQScale = 512;            // base codes are held as code * 512: the xor of two codes is 0 or >= 512
// cells (e+1, e+2) from words holding (e, e+1) and (e+2, e+3)
__device__ __forceinline__ s2 shl_cell(s2 cur, s2 next) { return s2_of(__builtin_amdgcn_alignbit(u32_of(next), u32_of(cur), 16)); }
// cells (e-1, e) from words holding (e-2, e-1) and (e, e+1)
__device__ __forceinline__ s2 shr_cell(s2 prev, s2 cur) { return s2_of(__builtin_amdgcn_alignbit(u32_of(cur), u32_of(prev), 16)); }

__device__ __forceinline__ uint32_t rev2_32(uint32_t x) {      // reverse the order of the 16 two-bit groups
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    return __builtin_bswap32(x);
}

// One sequence as the extension reads it (element t = base g0 + dir*t, complemented if comp), through a 64-bit SHIFT REGISTER: bits 1:0
// of `hs` are the next base the lane will read; consuming it is a funnel shift by two bits (by zero for the stream that does not move:
// no select).  Up to 16 bases are consumed between two checkpoints (every 16 steps, all lanes at once); a checkpoint tops the register
// up to 32 bases again from the block fetched one period ahead: no lane ever makes the wavefront wait on its own reload.
struct SeqWin {
    const uint32_t* packed;
    int64_t g0;
    int32_t dir;
    uint32_t comp;       // 0 or 0xFFFFFFFF (complement = bitwise not of the 2-bit codes)
    uint32_t len;
    uint32_t tp;         // stream index the register held at bits 1:0 at the last checkpoint
    uint32_t lo, hi;     // bases [t, t + 32): base t + i at bits 2i of hi:lo
    uint32_t nw;         // prefetched block [tp + 32, tp + 48)

    // 16 consecutive stream elements starting at t0, element j at bits 2j
    __device__ __forceinline__ uint32_t block16(uint32_t t0) const {
        const int64_t g = dir > 0 ? g0 + (int64_t)t0 : g0 - (int64_t)t0 - 15;
        uint32_t v;
        if (g >= 0) {
            const uint64_t w = (uint64_t)g >> 4;
            const uint32_t sh = (uint32_t)(g & 15) * 2;
            const uint32_t a = packed[w], b = packed[w + 1];
            v = sh ? ((a >> sh) | (b << (32 - sh))) : a;
        } else {                                                   // a reversed stream running off the front of the array:
            v = g > -16 ? packed[0] << (uint32_t)(-2 * g) : 0u;    // the missing positions are past the stream's end (masked)
        }
        if (dir < 0) v = rev2_32(v);
        return v ^ comp;
    }
    __device__ __forceinline__ void init(uint32_t t0) {
        tp = t0;
        lo = block16(t0);
        hi = block16(t0 + 16);
        nw = block16(t0 + 32);
    }
    // wave-uniform checkpoint (every 16 steps): t = next element this lane will read (t - tp <= 16 were consumed since the last one)
    __device__ __forceinline__ void checkpoint(uint32_t t) {
        const uint32_t c = t - tp;                                 // 0 .. 16 bases to top up: they enter at bit 2 * (32 - c)
        if (c) {
            // the register holds 32 - c valid bases; the c new ones are the first c elements of nw
            const uint64_t add = (uint64_t)nw << (2u * (32u - c));  // (c == 16: bits 32..63; smaller c: the top bits fall off, they are re-fetched)
            lo |= (uint32_t)add;
            hi |= (uint32_t)(add >> 32);
            tp = t;
            nw = block16(t + 32);
        }
    }
    // the base at stream index t = the register's bits 1:0 (t must be the lane's current read position)
    __device__ __forceinline__ int code(uint32_t t) const {
        const int c = (int)(lo & 3u);
        return t >= len ? (t == len ? kCodeNul : kCodePad) : c;
    }
    // consume one base if `take` = ~0, none if 0 (funnel shift by 2 or 0 bits)
    __device__ __forceinline__ void advance(uint32_t take) {
        const uint32_t sh = take & 2u;
        lo = __builtin_amdgcn_alignbit(hi, lo, sh);
        hi >>= sh;
    }
};

// ---- the two sequence windows of the band as BIT PLANES --------------------------------------------------------------------
// vqueryh / vqueryv (xavier.h:60-75, simdutils.h:263-289) hold one base per band cell.  A base code has three bits (0..3, the
// std::string terminator 4, the pad 5: xdrop.hpp), and all the step needs from the two windows is one flag per cell: do the bases
// differ.  So each window is three 32-bit words, plane p = bit p of the 32 cells' codes, and the 32 flags are
// (h0 ^ v0) | (h1 ^ v1) | (h2 ^ v2): four instructions per anti-diagonal instead of forty-eight on per-cell packed values, and a
// window shift is a handful of instructions instead of fifteen v_perm_b32.  Cell e sits at bit (e >> 1) + 16 * (e & 1) -- the even
// cells in the low half-word, the odd ones in the high half-word -- so that bits i and 16 + i are the two cells of band word i
// and `M & (0x00010001 << i)` is that word's pair of flags, one per 16-bit half: one v_and_b32 and one v_pk_mad_i16 make the
// packed +1 / -1 term from it.  Codes above 3 only enter a window in the step that takes the band onto a sequence end or after it
// (Phase 4), so plane 2 is all zero -- and is neither shifted nor written -- before that.
struct BandSeq { uint32_t h0, h1, h2, v0, v1, v2; };
__device__ __forceinline__ uint32_t bfi32(uint32_t m, uint32_t a, uint32_t b) { return (m & a) | (~m & b); }   // v_bfi_b32
// a * b + c per 16-bit half, b wave-uniform (the compiler would rather shift and subtract: two instructions)
__device__ __forceinline__ s2 pk_mad_s(s2 a, uint32_t b, s2 c) {
    s2 r;
    asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}
// vqueryh[e] = queryh[e + 1], vqueryv[e] = queryv[31 - e] for e < 31 (xavier.h:60-75); cell 31 never reaches a score (antiDiag3[31] = NINF)
__device__ __forceinline__ void bandseq_init(BandSeq& q, uint64_t hp, uint64_t vp) {
    q.h0 = q.h1 = q.h2 = q.v0 = q.v1 = q.v2 = 0u;
#pragma unroll
    for (int e = 0; e < kXLW; ++e) {
        const uint32_t hc = (uint32_t)(hp >> (2 * (e + 1))) & 3u, vc = (uint32_t)(vp >> (2 * (kXLW - e))) & 3u;
        const int b = (e >> 1) + 16 * (e & 1);
        q.h0 |= (hc & 1u) << b; q.h1 |= (hc >> 1) << b;
        q.v0 |= (vc & 1u) << b; q.v1 |= (vc >> 1) << b;
    }
}
// moveRight (rm = ~0: cell e <- cell e + 1 of the h window, the stream's next base into cell 30) or moveDown (rm = 0: cell e <- cell e - 1
// of the v window, the next base into cell 0), simdutils.h:263-289, per lane and without a branch: the byte selector of the one
// v_perm_b32 per word is the identity for the window that stays.  hlo / vlo: the streams' shift registers (bits 1:0 = next base).
__device__ __forceinline__ void bandseq_move(BandSeq& q, uint32_t rm, uint32_t hlo, uint32_t vlo) {
    const uint32_t selH = 0x03020100u ^ (rm & (0x03020100u ^ 0x05040302u));   // perm(w >> 1, w, .): {w.hi, (w >> 1).lo} / w
    const uint32_t selV = 0x05040100u ^ (rm & (0x05040100u ^ 0x07060504u));   // perm(w, w >> 15, .): {(w >> 15).lo, w.lo} / w
    const uint32_t mH = rm & 0x8000u, mV = ~rm & 1u;
    q.h0 = bfi32(mH, hlo << 15, __builtin_amdgcn_perm(q.h0 >> 1, q.h0, selH));
    q.h1 = bfi32(mH, hlo << 14, __builtin_amdgcn_perm(q.h1 >> 1, q.h1, selH));
    q.v0 = bfi32(mV, vlo, __builtin_amdgcn_perm(q.v0, q.v0 >> 15, selV));
    q.v1 = bfi32(mV, vlo >> 1, __builtin_amdgcn_perm(q.v1, q.v1 >> 15, selV));
}
// The same move for plane 2, and the code of the base that entered if it lies at or past its sequence's end (t = its stream
// index): the terminator (4) at t == len, the pad (5) beyond.  Only called once the band has stepped onto a sequence end.
__device__ __forceinline__ void bandseq_move_end(BandSeq& q, uint32_t rm, uint32_t t, uint32_t len) {
    const uint32_t selH = 0x03020100u ^ (rm & (0x03020100u ^ 0x05040302u));
    const uint32_t selV = 0x05040100u ^ (rm & (0x05040100u ^ 0x07060504u));
    const uint32_t mH = rm & 0x8000u, mV = ~rm & 1u;
    q.h2 = bfi32(mH, 0u, __builtin_amdgcn_perm(q.h2 >> 1, q.h2, selH));
    q.v2 = bfi32(mV, 0u, __builtin_amdgcn_perm(q.v2, q.v2 >> 15, selV));
    const uint32_t spm = (uint32_t)((int)(len - 1u - t) >> 31);   // ~0: t >= len (stream indices are far below 2^31)
    const uint32_t pad = (uint32_t)((int)(len - t) >> 31);        // ~0: t > len
    const uint32_t sH = mH & spm, sV = mV & spm;
    q.h0 = bfi32(sH, pad, q.h0); q.h1 &= ~sH; q.h2 |= sH;
    q.v0 = bfi32(sV, pad, q.v0); q.v1 &= ~sV; q.v2 |= sV;
}

// ---- one anti-diagonal on the packed band (shared by the three kernels below; they name the state a1 a2 a3 q off) ---------------------
// The upper clamp of adds_epi8 (127) is DEFERRED: the 16-bit add saturates at 32767 = 127 * 256 + 255, so a cell that would have been
// clamped is exactly 0x7FFF -- the largest value there is, and the only one with a non-zero low byte -- and the band maximum, which is
// computed anyway, shows it: a maximum above 0x7F00 means some cell overflowed, and only then are the sixteen v_pk_min_i16 of the clamp
// executed and the maximum taken again (BELLA_PFIX).  Cells above CUTOFF = 102 are rebased away, so in practice never; exact on any input.
#define BELLA_PSTEP()                                                                                               \
    {                                                                                                                 \
        const uint32_t M_ = (q.h0 ^ q.v0) | (q.h1 ^ q.v1) | (q.h2 ^ q.v2);        /* bit per cell: the bases differ */  \
        const uint32_t M8_ = M_ >> 8;                                                                                 \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                                             \
            const s2 fl = s2_of((i < 8 ? M_ : M8_) & (0x00010001u << (i & 7)));   /* 0 or 1 << (i & 7) per half */       \
            const s2 mt = pk_mad_s(fl, 0x00010001u * (uint32_t)(65536 - (512 >> (i & 7))), one); /* +1 match, -1 mismatch (x 256) */ \
            const s2 a1s = adds2(a1[i], mt);                                      /* adds_epi8, upper clamp deferred */  \
            const s2 shv = shl_cell(a2[i], i < 15 ? a2[i < 15 ? i + 1 : 15] : ninf);  /* shiftLeft(antiDiag2) */        \
            const s2 a2f = adds2(pmax(shv, a2[i]), mone);                         /* lower clamp = NINF exactly */       \
            a3[i] = pmax(a1s, a2f);                                                                                   \
        }                                                                                                             \
        a3[15].y = (short)(kXNinf * kXScale);                                                                         \
    }
// The band maximum (x 256) and the maximum of cells 0..15: the FIRST maximum lies above MIDDLE = 15 -- all xavier.h:160-175 asks of the
// arg-max -- iff every cell of the lower half is below the band maximum.
#define BELLA_PKEY(mxout, loout)                                                                                     \
    {                                                                                                                 \
        s2 lo_ = a3[0], up_ = a3[8];                                                                                  \
        _Pragma("unroll") for (int i = 1; i < 8; ++i) { lo_ = pmax(lo_, a3[i]); up_ = pmax(up_, a3[8 + i]); }          \
        up_ = pmax(up_, lo_);                                                                                         \
        loout = imax_((int)lo_.x, (int)lo_.y);                                                                        \
        mxout = imax_((int)up_.x, (int)up_.y);                                                                        \
    }
#define BELLA_PFIX(mxio, loio)                                                                                       \
    if ((mxio) > 0x7F00) {                                                                                            \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) a3[i] = pmin(a3[i], top);                                      \
        BELLA_PKEY(mxio, loio)                                                                                        \
    }
// rebase (xavier.h:152-158): antiDiag2 / antiDiag3 -= min of cells 0..30, saturating.  With a minimum >= 0 (the usual case: the band
// is full of real scores) nothing can saturate upwards, so the upper clamp is left out, and every cell that is not at the lower bound
// moves by the same amount: the first maximum stays where it is and the maxima follow by subtraction -- no second pass over the band.
// A negative minimum (cells near NINF inside the logical width) takes the clamped form and the maxima are taken again.
#define BELLA_PREBASE(mxio, loio)                                                                                    \
    {                                                                                                                 \
        s2 mn2 = a3[0];                                                                                               \
        _Pragma("unroll") for (int i = 1; i < 15; ++i) mn2 = pmin(mn2, a3[i]);                                        \
        const int mn = imin_(imin_((int)mn2.x, (int)mn2.y), (int)a3[15].x) >> 8; /* cells 0..30 (LOGICALWIDTH) */      \
        const s2 mnv = splat2(mn * kXScale);                                                                          \
        if (mn >= 0) {                                                                                                \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) { a2[i] = subs2(a2[i], mnv); a3[i] = subs2(a3[i], mnv); }  \
            mxio -= mn * kXScale;                                                                                     \
            loio = imax_(loio - mn * kXScale, kXNinf * kXScale);                                                      \
        } else {                                                                                                      \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) { a2[i] = pmin(subs2(a2[i], mnv), top); a3[i] = pmin(subs2(a3[i], mnv), top); } \
            BELLA_PKEY(mxio, loio)                                                                                    \
        }                                                                                                             \
        off += mn;                                                                                                    \
    }
// moveRight (simdutils.h:263-274) if rm = ~0, else (rm = 0) moveDown (:276-289), on the score vectors.  Each band word is produced by
// ONE v_perm_b32 whose byte selector is a per-lane register: selL = "cells (e+1,e+2)" for moveRight / "cells (e,e+1)" otherwise,
// selD = "cells (e,e+1)" for moveRight / "cells (e-1,e)" otherwise -- shift and select in a single instruction, no copies.
#define BELLA_PMOVE(rm)                                                                                              \
    {                                                                                                                 \
        const uint32_t selL = 0x03020100u ^ ((rm) & (0x03020100u ^ 0x05040302u));  /* perm(next, cur, .): shifted left / plain cur */   \
        const uint32_t selD = 0x05040302u ^ ((rm) & (0x05040302u ^ 0x07060504u));  /* perm(cur, prev, .): plain cur / shifted right */  \
        const uint32_t nf = u32_of(ninf);                                                                             \
        _Pragma("unroll") for (int i = 0; i < 16; ++i)                                                                \
            a1[i] = s2_of(__builtin_amdgcn_perm(i < 15 ? u32_of(a2[i < 15 ? i + 1 : 15]) : nf, u32_of(a2[i]), selL));      \
        _Pragma("unroll") for (int i = 0; i < 16; ++i)                                                                \
            a2[i] = s2_of(__builtin_amdgcn_perm(u32_of(a3[i]), i > 0 ? u32_of(a3[i > 0 ? i - 1 : 0]) : nf, selD));        \
    }

// xavier.h:257-274 on packed state.  h (len >= 32) and v (len >= 32) as SeqAcc geometry; dp = lane-strided LDS scratch.
__device__ __forceinline__ void xavier_one_direction_packed(const SeqAcc& Hacc, const SeqAcc& Vacc, const int X, int8_t* dp,
                                                            const int dps, XRes& r) {
    SeqWin H, V;
    H.packed = Hacc.packed; H.g0 = Hacc.g0; H.dir = Hacc.dir; H.comp = Hacc.comp ? 0xFFFFFFFFu : 0u; H.len = Hacc.len;
    V.packed = Vacc.packed; V.g0 = Vacc.g0; V.dir = Vacc.dir; V.comp = Vacc.comp ? 0xFFFFFFFFu : 0u; V.len = Vacc.len;
    const int hl = (int)H.len + 1, vl = (int)V.len + 1;
    // ---- Phase 1 (xavier.h:20-103): scalar DP on the 33x33 upper-left triangle (once per extension)
    const uint64_t hp = (uint64_t)H.block16(0) | ((uint64_t)H.block16(16) << 32);
    const uint64_t vp = (uint64_t)V.block16(0) | ((uint64_t)V.block16(16) << 32);
    int8_t* prev = dp;
    int8_t* cur = dp + 34 * dps;
    int8_t* ex1 = dp + 68 * dps;
    int8_t* ex2 = dp + 100 * dps;
    for (int j = 0; j < 34; ++j) prev[j * dps] = (int8_t)(-j);
    int DPmax = 0;
    for (int i = 1; i < kXLW + 2; ++i) {
        cur[0] = (int8_t)(-i);
        const int hc = (int)((hp >> (2 * (i - 1))) & 3);
        int left = -i;
        for (int j = 1; j <= kXLW + 2 - i; ++j) {
            const int vc = (int)((vp >> (2 * (j - 1))) & 3);
            const int oneF = (int)prev[(j - 1) * dps] + (hc == vc ? 1 : -1);
            const int twoF = imax_((int)prev[j * dps], left) - 1;
            const int val = imax_(oneF, twoF);
            cur[j * dps] = (int8_t)val;
            left = val;
            DPmax = imax_(DPmax, val);
        }
        if (i <= kXLW) ex1[(i - 1) * dps] = cur[(kXLW + 1 - i) * dps];
        if (i >= 2) ex2[(i - 1) * dps] = cur[(kXLW + 2 - i) * dps];
        int8_t* tmp = prev; prev = cur; cur = tmp;
    }
    s2 a1[16], a2[16], a3[16];
    BandSeq q;
    int adm = -128;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e0 = 2 * i, e1 = 2 * i + 1;
        const int a1x = (int)ex1[e0 * dps], a1y = e1 < kXLW ? (int)ex1[e1 * dps] : kXNinf;
        const int a2x = e0 >= 1 ? (int)ex2[e0 * dps] : kXNinf, a2y = (int)ex2[e1 * dps];
        a1[i] = mk2(a1x * kXScale, a1y * kXScale);
        a2[i] = mk2(a2x * kXScale, a2y * kXScale);
        a3[i] = splat2(kXNinf * kXScale);
        adm = imax_(adm, a1x);
        if (e1 < kXLW) adm = imax_(adm, a1y);
    }
    bandseq_init(q, hp, vp);
    int best = DPmax, off = 0;
    int hoff = kXLW, voff = kXLW;
    r.flagged = 0;
    if (adm < DPmax - X) { r.best = best; r.endH = hoff; r.endV = voff; r.steps = 0; return; }

    H.init((uint32_t)kXLW);
    V.init((uint32_t)kXLW);
    const s2 one = splat2(kXScale), mone = splat2(-kXScale), ninf = splat2(kXNinf * kXScale), top = splat2(kXTop);

    // ---- Phase 2 (xavier.h:105-183)
    uint32_t upm = 0u;                                             // ~0: the last positive band maximum lay above MIDDLE (maxpos > MIDDLE)
    int endH = hoff, endV = voff;
    bool first = true;
    uint32_t tick = 0;
    bool dropped = false;
    uint32_t rm = 0u;
    while (hoff < hl && voff < vl) {
        if ((tick & 15u) == 0u) { H.checkpoint((uint32_t)hoff); V.checkpoint((uint32_t)voff); }
        ++tick;
        BELLA_PSTEP()
        int mx, lo;
        BELLA_PKEY(mx, lo)
        BELLA_PFIX(mx, lo)
        const int curr = (mx >> 8) + off;
        if (curr < best - X) { dropped = true; break; }        // xavier.h:128-135: X-drop termination
        if ((mx >> 8) > kXCutoff) {
            BELLA_PREBASE(mx, lo)
        }
        if (curr > best) best = curr;
        if ((mx >> 8) > 0) upm = lo < mx ? ~0u : 0u;
        else if (first) r.flagged = 1;
        first = false;
        endH = hoff; endV = voff;
        rm = upm;
        const uint32_t hlo = H.lo, vlo = V.lo;
        H.advance(rm); V.advance(~rm);
        hoff += rm ? 1 : 0;
        voff += rm ? 0 : 1;
        BELLA_PMOVE(rm)
        bandseq_move(q, rm, hlo, vlo);
    }
    if (dropped) { r.best = best; r.endH = hoff; r.endV = voff; r.steps = (hoff - kXLW) + (voff - kXLW); return; }
    // the move that ended Phase 2 took the band onto a sequence end: the base that entered is the terminator
    bandseq_move_end(q, rm, rm ? (uint32_t)(hoff - 1) : (uint32_t)(voff - 1), rm ? H.len : V.len);
    // ---- Phase 4 (xavier.h:185-251)
    int dir = hoff >= hl ? 1 : 0;
    H.checkpoint((uint32_t)hoff); V.checkpoint((uint32_t)voff);      // 28 more steps: at most 14 per stream, inside the window
    for (int it = 0; it < kXLW - 3; ++it) {
        BELLA_PSTEP()
        int mx, lo;
        BELLA_PKEY(mx, lo)
        BELLA_PFIX(mx, lo)
        const int curr = (mx >> 8) + off;
        if (curr < best - X) break;
        if ((mx >> 8) > kXCutoff) {
            BELLA_PREBASE(mx, lo)
        }
        if (curr > best) best = curr;
        const int next = dir ^ 1;
        rm = next == 0 ? ~0u : 0u;
        const uint32_t hlo = H.lo, vlo = V.lo;
        H.advance(rm); V.advance(~rm);
        hoff += rm ? 1 : 0;
        voff += rm ? 0 : 1;
        BELLA_PMOVE(rm)
        bandseq_move(q, rm, hlo, vlo);
        bandseq_move_end(q, rm, rm ? (uint32_t)(hoff - 1) : (uint32_t)(voff - 1), rm ? H.len : V.len);
        dir = next;
    }
    r.best = best; r.endH = endH; r.endV = endV; r.steps = (hoff - kXLW) + (voff - kXLW);
}

__global__ __launch_bounds__(kXdropBlock) void k_xdrop_packed(XdropArgs a) {
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
    if (valid) {
        const uint64_t goffH = a.roff[rid], goffV = a.roff[cid];
        make_geom(a.packed, goffH, (uint32_t)(a.roff[rid + 1] - goffH), goffV, (uint32_t)(a.roff[cid + 1] - goffV), seedH, seedV,
                  a.k, g);
        SeqAcc H, V;
        make_accessors(a.packed, goffH, goffV, g, which, H, V);
        if (H.len >= (uint32_t)kXW && V.len >= (uint32_t)kXW) {
            ran = true;
            xavier_one_direction_packed(H, V, a.xdrop, dp + threadIdx.x, kXdropBlock, res);
        }
    }
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

// ---- length-sorted scheduling ---------------------------------------------------------------------------------------
// The extensions of a batch differ a lot in length (seed position, read lengths, junk pairs): with pair order a wavefront
// runs until its longest lane ends.  Plan -> sort by an upper bound of the steps (descending: longest first) -> run in
// that order -> combine left/right per pair.
struct XdropSortedArgs {
    XdropArgs a;
    uint32_t* est;          // [2n] upper bound on anti-diagonal steps of extension e = 2*pair + which
    uint32_t* ids;          // [2n] extension ids (iota before the sort, order after)
    const uint32_t* order;  // [2n] sorted extension ids
    int4* res;              // [2n] {best, endH, endV, flagged | ran << 1 | steps << 2}
};

__device__ __forceinline__ void xdrop_load_pair(const XdropArgs& a, uint64_t p, uint32_t& rid, uint32_t& cid, uint32_t& seedH,
                                                uint32_t& seedV) {
    if (a.seeds) { const bella_seed s = a.seeds[p]; rid = s.rid; cid = s.cid; seedH = s.seedH; seedV = s.seedV; }
    else { const bella_pair s = a.pairs[p]; rid = s.rid; cid = s.cid; seedH = s.seedH; seedV = s.seedV; }
}

__global__ void k_xdrop_plan(XdropSortedArgs sa) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 2 * sa.a.n) return;
    uint32_t rid, cid, seedH, seedV;
    xdrop_load_pair(sa.a, e >> 1, rid, cid, seedH, seedV);
    const uint64_t goffH = sa.a.roff[rid], goffV = sa.a.roff[cid];
    PairGeom g;
    make_geom(sa.a.packed, goffH, (uint32_t)(sa.a.roff[rid + 1] - goffH), goffV, (uint32_t)(sa.a.roff[cid + 1] - goffV), seedH, seedV,
              sa.a.k, g);
    SeqAcc H, V;
    make_accessors(sa.a.packed, goffH, goffV, g, (int)(e & 1), H, V);
    const bool run = H.len >= (uint32_t)kXW && V.len >= (uint32_t)kXW;
    // Phase 2 ends when EITHER sequence is exhausted and the band follows the diagonal: about 2 * min(lenH, lenV) moves (the hard
    // bound lenH + lenV orders unequal pairs badly; the key only schedules, any value is correct)
    const uint32_t mn = H.len < V.len ? H.len : V.len;
    sa.est[e] = run ? 2u * mn : 0u;
    sa.ids[e] = (uint32_t)e;
}

__global__ __launch_bounds__(kXdropBlock) void k_xdrop_sorted(XdropSortedArgs sa) {
    __shared__ int8_t dp[132 * kXdropBlock];
    const XdropArgs& a = sa.a;
    const uint64_t t = (uint64_t)blockIdx.x * kXdropBlock + threadIdx.x;
    if (t >= 2 * a.n) return;
    const uint32_t e = sa.order[t];
    uint32_t rid, cid, seedH, seedV;
    xdrop_load_pair(a, e >> 1, rid, cid, seedH, seedV);
    const uint64_t goffH = a.roff[rid], goffV = a.roff[cid];
    PairGeom g;
    make_geom(a.packed, goffH, (uint32_t)(a.roff[rid + 1] - goffH), goffV, (uint32_t)(a.roff[cid + 1] - goffV), seedH, seedV, a.k, g);
    SeqAcc H, V;
    make_accessors(a.packed, goffH, goffV, g, (int)(e & 1), H, V);
    XRes res;
    res.best = 0; res.endH = 0; res.endV = 0; res.flagged = 0; res.steps = 0;
    bool ran = false;
    if (H.len >= (uint32_t)kXW && V.len >= (uint32_t)kXW) {
        ran = true;
        xavier_one_direction_packed(H, V, a.xdrop, dp + threadIdx.x, kXdropBlock, res);
    }
    sa.res[e] = make_int4(res.best, res.endH, res.endV, (res.flagged & 1) | (ran ? 2 : 0) | (res.steps << 2));
}

// ---- slices: no lane waits for the longest extension of its wavefront ------------------------------------------------------
// k_xdrop_sorted runs a wavefront until its LONGEST lane ends: even in length-sorted order about a fifth of the lane-steps are
// idle tails.  Here an extension's state lives in HBM between launches (46 words: the two band vectors, the six bit planes of the
// band's two sequence windows, the scalars) and the work goes in SLICES of at most kXdropSlice anti-diagonal steps: k_xdrop_begin runs
// Phase 1 (xavier.h:20-103) for every extension and stores its state; every launch of k_xdrop_slice loads the state of the
// extensions still alive -- compacted: every wavefront is full --, steps them, and stores the survivors for the next launch.
// A lane idles at most the rest of ONE slice (half a slice on average against ~4,300 steps per extension).  Phase 2 and
// Phase 4 (xavier.h:105-183 / :185-251) are one loop body with a per-lane mode.  Same arithmetic as
// xavier_one_direction_packed, same results.
#ifndef BELLA_XDROP_SLICE
#define BELLA_XDROP_SLICE 512
#endif
constexpr int kXdropSlice = BELLA_XDROP_SLICE;
constexpr uint32_t kXStateWords = 46;       // a1[16] a2[16] h0 h1 h2 v0 v1 v2 best off hoff voff endH endV flags e
constexpr uint32_t kXsQ = 32, kXsBest = 38, kXsOff = 39, kXsHoff = 40, kXsVoff = 41, kXsEndH = 42, kXsEndV = 43, kXsFlags = 44, kXsExt = 45;
struct XdropSliceArgs {
    XdropSortedArgs s;
    uint32_t* state;             // [kXStateWords][cap]: word w of slot t at state[w * cap + t]
    uint64_t cap;                // slots of this batch
    uint64_t first;              // position of the batch's first extension in the sorted order
    uint64_t count;              // extensions of the batch
    const uint32_t* live_in;     // slots to step (nullptr: all slots of the batch, first launch)
    const uint32_t* nlive_in;    // their number (device)
    uint32_t* live_out;          // survivors
    uint32_t* nlive_out;
    int steps;                   // anti-diagonal steps of this launch (a multiple of 16, at most kXdropSlice)
    int prio;                    // wave priority of this launch (0 .. 3): the classes of the longest extensions run ahead (bella_hip.hip: run_xdrop)
};

// flags word: maxpos > MIDDLE (bit 0) | first << 5 | flagged << 6 | dir << 8 | it4 << 9 (5 bits) | dead << 15
__global__ __launch_bounds__(kXdropBlock) void k_xdrop_begin(XdropSliceArgs xa) {
    __shared__ int8_t dpm[132 * kXdropBlock];
    const XdropSortedArgs& sa = xa.s;
    const XdropArgs& a = sa.a;
    const uint64_t t = (uint64_t)blockIdx.x * kXdropBlock + threadIdx.x;
    if (t >= xa.count) return;
    const uint32_t e = sa.order[xa.first + t];
    uint32_t* const st = xa.state + t;
    const uint64_t cap = xa.cap;
    uint32_t rid, cid, seedH, seedV;
    xdrop_load_pair(a, e >> 1, rid, cid, seedH, seedV);
    const uint64_t goffH = a.roff[rid], goffV = a.roff[cid];
    PairGeom g;
    make_geom(a.packed, goffH, (uint32_t)(a.roff[rid + 1] - goffH), goffV, (uint32_t)(a.roff[cid + 1] - goffV), seedH, seedV, a.k, g);
    SeqAcc Hacc, Vacc;
    make_accessors(a.packed, goffH, goffV, g, (int)(e & 1), Hacc, Vacc);
    st[kXsExt * cap] = e;
    if (!(Hacc.len >= (uint32_t)kXW && Vacc.len >= (uint32_t)kXW)) {
        sa.res[e] = make_int4(0, 0, 0, 0);                         // did not run (xavier.h:338-342 / :356-360)
        st[kXsFlags * cap] = 1u << 15;
        return;
    }
    SeqWin H, V;
    H.packed = Hacc.packed; H.g0 = Hacc.g0; H.dir = Hacc.dir; H.comp = Hacc.comp ? 0xFFFFFFFFu : 0u; H.len = Hacc.len;
    V.packed = Vacc.packed; V.g0 = Vacc.g0; V.dir = Vacc.dir; V.comp = Vacc.comp ? 0xFFFFFFFFu : 0u; V.len = Vacc.len;
    // ---- Phase 1 (xavier.h:20-103): scalar DP on the 33x33 upper-left triangle (once per extension)
    int8_t* const dp = dpm + threadIdx.x;
    constexpr int dps = kXdropBlock;
    const uint64_t hp = (uint64_t)H.block16(0) | ((uint64_t)H.block16(16) << 32);
    const uint64_t vp = (uint64_t)V.block16(0) | ((uint64_t)V.block16(16) << 32);
    int8_t* prev = dp;
    int8_t* cur = dp + 34 * dps;
    int8_t* ex1 = dp + 68 * dps;
    int8_t* ex2 = dp + 100 * dps;
    for (int j = 0; j < 34; ++j) prev[j * dps] = (int8_t)(-j);
    int DPmax = 0;
    for (int i = 1; i < kXLW + 2; ++i) {
        cur[0] = (int8_t)(-i);
        const int hc = (int)((hp >> (2 * (i - 1))) & 3);
        int left = -i;
        for (int j = 1; j <= kXLW + 2 - i; ++j) {
            const int vc = (int)((vp >> (2 * (j - 1))) & 3);
            const int oneF = (int)prev[(j - 1) * dps] + (hc == vc ? 1 : -1);
            const int twoF = imax_((int)prev[j * dps], left) - 1;
            const int val = imax_(oneF, twoF);
            cur[j * dps] = (int8_t)val;
            left = val;
            DPmax = imax_(DPmax, val);
        }
        if (i <= kXLW) ex1[(i - 1) * dps] = cur[(kXLW + 1 - i) * dps];
        if (i >= 2) ex2[(i - 1) * dps] = cur[(kXLW + 2 - i) * dps];
        int8_t* tmp = prev; prev = cur; cur = tmp;
    }
    int adm = -128;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e0 = 2 * i, e1 = 2 * i + 1;
        const int a1x = (int)ex1[e0 * dps], a1y = e1 < kXLW ? (int)ex1[e1 * dps] : kXNinf;
        const int a2x = e0 >= 1 ? (int)ex2[e0 * dps] : kXNinf, a2y = (int)ex2[e1 * dps];
        adm = imax_(adm, a1x);
        if (e1 < kXLW) adm = imax_(adm, a1y);
        st[(uint64_t)i * cap] = u32_of(mk2(a1x * kXScale, a1y * kXScale));
        st[(uint64_t)(16 + i) * cap] = u32_of(mk2(a2x * kXScale, a2y * kXScale));
    }
    BandSeq q;
    bandseq_init(q, hp, vp);
    st[(kXsQ + 0) * cap] = q.h0; st[(kXsQ + 1) * cap] = q.h1; st[(kXsQ + 2) * cap] = q.h2;
    st[(kXsQ + 3) * cap] = q.v0; st[(kXsQ + 4) * cap] = q.v1; st[(kXsQ + 5) * cap] = q.v2;
    if (adm < DPmax - a.xdrop) {                                   // xavier.h:96-101: X-drop inside Phase 1
        sa.res[e] = make_int4(DPmax, kXLW, kXLW, 2);
        st[kXsFlags * cap] = 1u << 15;
        return;
    }
    st[kXsBest * cap] = (uint32_t)DPmax; st[kXsOff * cap] = 0u; st[kXsHoff * cap] = (uint32_t)kXLW; st[kXsVoff * cap] = (uint32_t)kXLW;
    st[kXsEndH * cap] = (uint32_t)kXLW; st[kXsEndV * cap] = (uint32_t)kXLW;
    st[kXsFlags * cap] = 1u << 5;                                  // maxpos 0, first, not flagged
}

#ifndef BELLA_XSLICE_WAVES
#define BELLA_XSLICE_WAVES 4                 // 128 VGPRs: this kernel lives on big batches, where the state loads want occupancy (with the
                                             // clamp-free variant of the step next to the clamped one it needs 161: 3.35 s against 3.26 s on the 100k set)
#endif
__global__ __launch_bounds__(kXdropBlock, BELLA_XSLICE_WAVES) void k_xdrop_slice(XdropSliceArgs xa) {
    const XdropSortedArgs& sa = xa.s;
    const XdropArgs& a = sa.a;
    const uint64_t x = (uint64_t)blockIdx.x * kXdropBlock + threadIdx.x;
    const uint64_t nlive = xa.live_in ? (uint64_t)*xa.nlive_in : xa.count;
    const uint64_t cap = xa.cap;
    if (xa.prio >= 3) __builtin_amdgcn_s_setprio(3);
    else if (xa.prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (xa.prio == 1) __builtin_amdgcn_s_setprio(1);
    bool active = x < nlive;
    uint64_t t = 0;
    uint32_t flags = 1u << 15;
    if (active) {
        t = xa.live_in ? (uint64_t)xa.live_in[x] : x;
        flags = xa.state[kXsFlags * cap + t];
        active = !((flags >> 15) & 1u);
    }
    const unsigned long long any = __ballot(active);
    if (!any) return;
    uint32_t* const st = xa.state + t;
    s2 a1[16], a2[16], a3[16];
    BandSeq q;
    int best = 0, off = 0, hoff = kXLW, voff = kXLW, endH = 0, endV = 0;
    uint32_t e = 0;
    SeqWin H, V;
    H.packed = a.packed; V.packed = a.packed; H.g0 = 0; V.g0 = 0; H.dir = 1; V.dir = 1; H.comp = 0; V.comp = 0; H.len = 64; V.len = 64;
    if (active) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { a1[i] = s2_of(st[(uint64_t)i * cap]); a2[i] = s2_of(st[(uint64_t)(16 + i) * cap]); }
        q.h0 = st[(kXsQ + 0) * cap]; q.h1 = st[(kXsQ + 1) * cap]; q.h2 = st[(kXsQ + 2) * cap];
        q.v0 = st[(kXsQ + 3) * cap]; q.v1 = st[(kXsQ + 4) * cap]; q.v2 = st[(kXsQ + 5) * cap];
        best = (int)st[kXsBest * cap]; off = (int)st[kXsOff * cap]; hoff = (int)st[kXsHoff * cap]; voff = (int)st[kXsVoff * cap];
        endH = (int)st[kXsEndH * cap]; endV = (int)st[kXsEndV * cap]; e = st[kXsExt * cap];
        uint32_t rid, cid, seedH, seedV;
        xdrop_load_pair(a, e >> 1, rid, cid, seedH, seedV);
        const uint64_t goffH = a.roff[rid], goffV = a.roff[cid];
        PairGeom g;
        make_geom(a.packed, goffH, (uint32_t)(a.roff[rid + 1] - goffH), goffV, (uint32_t)(a.roff[cid + 1] - goffV), seedH, seedV, a.k, g);
        SeqAcc Hacc, Vacc;
        make_accessors(a.packed, goffH, goffV, g, (int)(e & 1), Hacc, Vacc);
        H.g0 = Hacc.g0; H.dir = Hacc.dir; H.comp = Hacc.comp ? 0xFFFFFFFFu : 0u; H.len = Hacc.len;
        V.g0 = Vacc.g0; V.dir = Vacc.dir; V.comp = Vacc.comp ? 0xFFFFFFFFu : 0u; V.len = Vacc.len;
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) { a1[i] = splat2(0); a2[i] = splat2(0); }
        q.h0 = q.h1 = q.h2 = q.v0 = q.v1 = q.v2 = 0u;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) a3[i] = splat2(0);
    H.init((uint32_t)hoff);
    V.init((uint32_t)voff);
    const int hl = (int)H.len + 1, vl = (int)V.len + 1;
    // per-lane switches as masks (0 / ~0): the step below selects with them (v_bfi_b32, and / or / xor) instead of branching
    uint32_t upm = (flags & 1u) ? ~0u : 0u;                       // the last positive band maximum lay above MIDDLE
    uint32_t dirm = ((flags >> 8) & 1u) ? ~0u : 0u;               // Phase 4: the last move was down
    int flagged = (int)((flags >> 6) & 1u), it4 = (int)((flags >> 9) & 31u);
    uint32_t firstm = ((flags >> 5) & 1u) ? ~0u : 0u;            // the first step of Phase 2 is still to come
    const int X = a.xdrop;
    const s2 one = splat2(kXScale), mone = splat2(-kXScale), ninf = splat2(kXNinf * kXScale), top = splat2(kXTop);

    // one loop, single exit (an in-loop return makes the compiler keep two copies of the band state).  Phase 2 (xavier.h:105-183) and
    // Phase 4 (:185-251) share the body: the per-lane choices are masks, not branches.  A finished lane has stored its result and idles
    // for the rest of the slice.
    bool done = false;
#ifndef BELLA_X_FLAT_LOOP
    // sixteen steps between two checkpoints are one inner loop: the wave-uniform exit test and the checkpoint test are made once per
    // sixteen steps instead of every step (a wavefront whose last lane ends idles at most fifteen steps more)
    for (int step16 = 0; step16 < xa.steps; step16 += 16) {
        if (!__ballot(active && !done)) break;                    // (wave-uniform)
        H.checkpoint((uint32_t)hoff); V.checkpoint((uint32_t)voff);
#pragma unroll 1
      for (int step = 0; step < 16; ++step) {
#else
    for (int step = 0; step < xa.steps; ++step) {
        if (!__ballot(active && !done)) break;                    // (wave-uniform)
        if ((step & 15) == 0) { H.checkpoint((uint32_t)hoff); V.checkpoint((uint32_t)voff); }
      {
#endif
        BELLA_PSTEP()
        int mx, lo;
        BELLA_PKEY(mx, lo)
        BELLA_PFIX(mx, lo)                                       // (the deferred upper clamp: sixteen more instructions only when a cell overflowed)
        const int curr = (mx >> 8) + off;
        // Phase 4 = the band stands on or past a sequence end (xavier.h:185-190) -- a function of the offsets, not a state of its own
        const uint32_t m4 = (uint32_t)(((hl - 1 - hoff) | (vl - 1 - voff)) >> 31);
        const bool live = active && !done;
        const bool drop = curr < best - X;
        if (live && drop) {
            // xavier.h:128-135 (Phase 2: the seed ends at the current offsets) / :212-219 (Phase 4: at the last Phase-2 offsets)
            sa.res[e] = make_int4(best, (int)bfi32(m4, (uint32_t)endH, (uint32_t)hoff), (int)bfi32(m4, (uint32_t)endV, (uint32_t)voff),
                                  (flagged & 1) | 2 | (((hoff - kXLW) + (voff - kXLW)) << 2));
            done = true;
        }
        if (live && !drop) {
            if ((mx >> 8) > kXCutoff) {
                BELLA_PREBASE(mx, lo)
            }
            best = imax_(best, curr);
            // xavier.h:160-175: Phase 2 follows the first band maximum if it is positive (else the previous choice stands; on the very
            // first step that is a flag of the result); Phase 4 alternates
            const uint32_t pos = (uint32_t)((kXScale - 1 - mx) >> 31);     // ~0: band maximum > 0
            const uint32_t upn = (uint32_t)((lo - mx) >> 31);              // ~0: every cell of the lower half is below it
            flagged |= (int)(firstm & ~pos & 1u);
            firstm = 0u;
            const uint32_t sel = pos & ~m4;
            upm = bfi32(sel, upn, upm);
            dirm ^= m4;
            const uint32_t rm = bfi32(m4, ~dirm, upm);            // ~0: moveRight, 0: moveDown
            const uint32_t hlo = H.lo, vlo = V.lo;
            H.advance(rm); V.advance(~rm);
            hoff -= (int)rm;
            voff += 1 + (int)rm;
            BELLA_PMOVE(rm)
            bandseq_move(q, rm, hlo, vlo);
            if (!(hoff < hl && voff < vl)) {                      // on or past a sequence end: the step that ends Phase 2, and all of Phase 4
                bandseq_move_end(q, rm, rm ? (uint32_t)(hoff - 1) : (uint32_t)(voff - 1), rm ? H.len : V.len);
                // xavier.h:185-190: the step that ends Phase 2 notes the offsets before its move and which sequence ended; Phase 4
                // counts its 28 steps (:192-251).  Straight-line on masks: another region of per-lane branches here costs the band's
                // registers a copy per step.
                const uint32_t hend = (uint32_t)((hl - 1 - hoff) >> 31);   // ~0: hoff >= hl
                dirm = bfi32(m4, dirm, hend);
                endH = (int)bfi32(m4, (uint32_t)endH, (uint32_t)(hoff + (int)rm));
                endV = (int)bfi32(m4, (uint32_t)endV, (uint32_t)(voff - 1 - (int)rm));
                it4 = (it4 + 1) & (int)m4;
                if (it4 == kXLW - 3) {
                    sa.res[e] = make_int4(best, endH, endV, (flagged & 1) | 2 | (((hoff - kXLW) + (voff - kXLW)) << 2));
                    done = true;
                }
            }
        }
      }
    }
    if (active && done) st[kXsFlags * cap] = 1u << 15;
    const bool keep = active && !done;
    if (keep) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { st[(uint64_t)i * cap] = u32_of(a1[i]); st[(uint64_t)(16 + i) * cap] = u32_of(a2[i]); }
        st[(kXsQ + 0) * cap] = q.h0; st[(kXsQ + 1) * cap] = q.h1; st[(kXsQ + 2) * cap] = q.h2;
        st[(kXsQ + 3) * cap] = q.v0; st[(kXsQ + 4) * cap] = q.v1; st[(kXsQ + 5) * cap] = q.v2;
        st[kXsBest * cap] = (uint32_t)best; st[kXsOff * cap] = (uint32_t)off; st[kXsHoff * cap] = (uint32_t)hoff; st[kXsVoff * cap] = (uint32_t)voff;
        st[kXsEndH * cap] = (uint32_t)endH; st[kXsEndV * cap] = (uint32_t)endV;
        st[kXsFlags * cap] = (upm & 1u) | ((firstm & 1u) << 5) | ((uint32_t)flagged << 6) | ((dirm & 1u) << 8) | ((uint32_t)it4 << 9);
    }
    // survivors, compacted: one atomic per wavefront, the wavefront's survivors stay together (similar remaining lengths)
    const unsigned long long km = __ballot(keep);
    if (km) {
        uint32_t base = 0;
        if (lane_id() == (uint32_t)(__ffsll((long long)km) - 1)) base = atomicAdd(xa.nlive_out, (uint32_t)__popcll(km));
        base = (uint32_t)__shfl((int)base, __ffsll((long long)km) - 1, 64);
        if (keep) xa.live_out[base + (uint32_t)__popcll(km & ((1ull << lane_id()) - 1ull))] = (uint32_t)t;
    }
}

__global__ void k_xdrop_finish(XdropSortedArgs sa) {
    const XdropArgs& a = sa.a;
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.n) return;
    uint32_t rid, cid, seedH, seedV;
    xdrop_load_pair(a, p, rid, cid, seedH, seedV);
    const uint64_t goffH = a.roff[rid], goffV = a.roff[cid];
    PairGeom g;
    make_geom(a.packed, goffH, (uint32_t)(a.roff[rid + 1] - goffH), goffV, (uint32_t)(a.roff[cid + 1] - goffV), seedH, seedV, a.k, g);
    const int4 l = sa.res[2 * p], r = sa.res[2 * p + 1];
    XRes L, R;
    L.best = l.x; L.endH = l.y; L.endV = l.z; L.flagged = l.w & 1; L.steps = l.w >> 2;
    R.best = r.x; R.endH = r.y; R.endV = r.z; R.flagged = r.w & 1; R.steps = r.w >> 2;
    bella_aln out;
    finish_pair(g, (l.w & 2) != 0, L, (r.w & 2) != 0, R, a.ratiophi, a.delta, out);
    a.out[p] = out;
}

#undef BELLA_PSTEP
#undef BELLA_PFIX
#undef BELLA_PKEY
#undef BELLA_PREBASE
#undef BELLA_PMOVE

}  // namespace bella
#endif
