// core.hpp -- per-lane building blocks of the HIP kernels (semiring fold, seed choice, k-mer words).
//
// Everything here is a pure function of its arguments and is inlined into the gfx950 kernels of
// spgemm.hpp / assemble.hpp.  The same functions compile as plain C++ so that tests/harness/ can
// unit-test the exact code the kernels run against the oracle on the CPU (that harness is test
// infrastructure; libbella_hip.so exports none of this and has no CPU execution path).
//
// Reference semantics restated here (paths relative to the reference tree):
//   include/chain.hpp:35-71   checkstrand / overlapop
//   include/chain.hpp:74-150  multiop / chainop  (the order-dependent position-binning semiring)
//   include/common/common.h:142-170  chain() / choose()  (std::sort by support, keep ids[0])
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define BELLA_HD __host__ __device__ __forceinline__
#define BELLA_HD_NOINLINE __host__ __device__
#else
#define BELLA_HD inline
#define BELLA_HD_NOINLINE inline
#endif

namespace bella {

constexpr uint32_t kEmpty = 0xFFFFFFFFu;

BELLA_HD int iabs_(int x) { return x < 0 ? -x : x; }
BELLA_HD int imin_(int a, int b) { return a < b ? a : b; }

// chain.hpp:47-71 overlapop with the strand test already decided (`oriented` == checkstrand()).
// Returns the int the reference returns; callers truncate to u16 where the reference stores it in a
// vector<unsigned short> (chain.hpp:85).
BELLA_HD int overlap_estimate(uint32_t posH, uint32_t posV, uint32_t lenH, uint32_t lenV, bool oriented, uint32_t k) {
    uint16_t begpH = (uint16_t)posH, begpV = (uint16_t)posV;
    if (!oriented) begpH = (uint16_t)(lenH - begpH - k);                 // chain.hpp:57-60 (u16 wrap)
    uint16_t endpH = (uint16_t)(begpH + k), endpV = (uint16_t)(begpV + k);  // :63-64
    int margin1 = imin_(begpH, begpV);                                    // :66
    int margin2 = imin_((int)lenH - (int)endpH, (int)lenV - (int)endpV);  // :67
    return margin1 + margin2 + (int)k;                                    // :68
}

// ---- 2-bit packed reads: base t of the concatenated read array sits at bits 2(t%16).. of word t/16 ----
// A=0 C=1 G=2 T=3 (kmercode/Kmer.cpp:205-228).  The array is padded with 4 zero words.
BELLA_HD uint64_t kmer_le(const uint32_t* packed, uint64_t g, uint32_t k) {  // base t at bits 2t (little-endian)
    uint64_t w = g >> 4;
    uint32_t sh = (uint32_t)(g & 15) * 2;
    uint64_t lo = (uint64_t)packed[w] | ((uint64_t)packed[w + 1] << 32);
    uint64_t hi = packed[w + 2];
    uint64_t x = sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
    return k >= 32 ? x : (x & ((1ull << (2 * k)) - 1));
}
BELLA_HD uint64_t rev2_(uint64_t x) {  // reverse the order of the 32 two-bit groups
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    x = ((x >> 8) & 0x00FF00FF00FF00FFull) | ((x & 0x00FF00FF00FF00FFull) << 8);
    x = ((x >> 16) & 0x0000FFFF0000FFFFull) | ((x & 0x0000FFFF0000FFFFull) << 16);
    return (x >> 32) | (x << 32);
}
// forward word (base 0 most significant, the order of Kmer::operator<, Kmer.cpp:160-169) and the word of
// the reverse complement (Kmer::twin, Kmer.cpp:324-355), both right-aligned in 2k bits
BELLA_HD uint64_t kmer_fw_from_le(uint64_t le, uint32_t k) { return rev2_(le) >> (64 - 2 * k); }
BELLA_HD uint64_t kmer_rc_from_le(uint64_t le, uint32_t k) { return k >= 32 ? ~le : ((~le) & ((1ull << (2 * k)) - 1)); }

// ---- libstdc++ std::sort restated for u16 ids with comp(a,b) = sup(a) > sup(b)  (common.h:112-117,145) ----
// Needed only when a pair ends with more than 16 bins (_S_threshold): then std::sort is an unstable
// introsort and ids[0] is not simply "lowest index among the maxima".
template <class SupFn>
struct IdSorter {
    SupFn sup;
    uint16_t* a;
    BELLA_HD bool gt(uint16_t x, uint16_t y) const { return sup(x) > sup(y); }
    BELLA_HD void swp(long i, long j) { uint16_t t = a[i]; a[i] = a[j]; a[j] = t; }
    BELLA_HD void adjust_heap(long first, long hole, long len, uint16_t value) {
        const long top = hole;
        long child = hole;
        while (child < (len - 1) / 2) {
            child = 2 * (child + 1);
            if (gt(a[first + child], a[first + child - 1])) child--;
            a[first + hole] = a[first + child];
            hole = child;
        }
        if ((len & 1) == 0 && child == (len - 2) / 2) {
            child = 2 * (child + 1);
            a[first + hole] = a[first + child - 1];
            hole = child - 1;
        }
        long parent = (hole - 1) / 2;
        while (hole > top && gt(a[first + parent], value)) {
            a[first + hole] = a[first + parent];
            hole = parent;
            parent = (hole - 1) / 2;
        }
        a[first + hole] = value;
    }
    BELLA_HD void heap_sort(long first, long last) {
        long len = last - first;
        if (len >= 2) {
            long parent = (len - 2) / 2;
            for (;;) {
                uint16_t v = a[first + parent];
                adjust_heap(first, parent, len, v);
                if (parent == 0) break;
                parent--;
            }
        }
        while (last - first > 1) {
            --last;
            uint16_t v = a[last];
            a[last] = a[first];
            adjust_heap(first, 0, last - first, v);
        }
    }
    BELLA_HD void median_to_first(long r, long x, long y, long z) {
        if (gt(a[x], a[y])) {
            if (gt(a[y], a[z])) swp(r, y);
            else if (gt(a[x], a[z])) swp(r, z);
            else swp(r, x);
        } else if (gt(a[x], a[z])) swp(r, x);
        else if (gt(a[y], a[z])) swp(r, z);
        else swp(r, y);
    }
    BELLA_HD long partition(long first, long last, long pivot) {
        for (;;) {
            while (gt(a[first], a[pivot])) ++first;
            --last;
            while (gt(a[pivot], a[last])) --last;
            if (!(first < last)) return first;
            swp(first, last);
            ++first;
        }
    }
    BELLA_HD void linear_insert(long last) {
        uint16_t v = a[last];
        long next = last - 1;
        while (gt(v, a[next])) { a[last] = a[next]; last = next; --next; }
        a[last] = v;
    }
    BELLA_HD void insertion_sort(long first, long last) {
        if (first == last) return;
        for (long i = first + 1; i != last; ++i) {
            if (gt(a[i], a[first])) {
                uint16_t v = a[i];
                for (long j = i; j > first; --j) a[j] = a[j - 1];
                a[first] = v;
            } else linear_insert(i);
        }
    }
    BELLA_HD_NOINLINE void sort(long n) {
        if (n <= 0) return;
        long lg = 0;
        while ((1L << (lg + 1)) <= n) lg++;
        // __introsort_loop with an explicit stack (the recursion is on the right part, bits/stl_algo.h)
        int stf[40], stl[40], std_[40];      // depth <= 2*lg(65536) = 32 pending right parts
        int sp = 0;
        long first = 0, last = n, depth = 2 * lg;
        for (;;) {
            while (last - first > 16) {
                if (depth == 0) { heap_sort(first, last); break; }
                --depth;
                long mid = first + (last - first) / 2;
                median_to_first(first, first + 1, mid, last - 1);
                long cut = partition(first + 1, last, first);
                stf[sp] = (int)cut; stl[sp] = (int)last; std_[sp] = (int)depth; sp++;   // recurse on [cut,last) FIRST
                last = cut;
            }
            // the reference recursion order is right part first, then the loop continues on the left; the
            // final array does not depend on that order (the parts are disjoint), so process the stack now
            if (sp == 0) break;
            --sp;
            first = stf[sp]; last = stl[sp]; depth = std_[sp];
        }
        if (n > 16) {
            insertion_sort(0, 16);
            for (long i = 16; i != n; ++i) linear_insert(i);
        } else insertion_sort(0, n);
    }
};

// ---- the semiring value of ONE pair, folded in place over the pair's product list -------------------
// P[0..m)  : products in the reference's order, P[t] = posH | posV << 16
// Bm[0..m) : Bm[t] & 0xFFFF = u16 overlap estimate of product t (multiop, chain.hpp:74-86); the upper
//            half is scratch on entry.
// While folding, the state lives in the same two arrays: positions P[0..npos) and bins Bm[0..nb)
// (Bm[b] = overlap | support << 16), both in REVERSED logical order (the logical head bin, chain.hpp's
// m1->pos[0], is the LAST bin in memory; its first position -- choose()'s seed -- is its last word).
// Product t is read before anything at index >= t is written, and the state never holds more than t
// positions / bins before product t, so no extra arena is needed.
struct FoldResult {
    uint16_t count;    // spmatType_::count
    uint16_t nbins;
    uint16_t support;  // of the winning bin
    uint16_t binov;    // overlap[] of the winning bin
    uint32_t seed;     // posH | posV << 16 of the winning bin's pos[0]
    uint32_t many_bins;  // nb > 16 (std::sort tie path used)
};

// 1 iff |posH - qH| > k and |posV - qV| > k for two packed (posH | posV << 16) positions (chain.hpp:88-97)
BELLA_HD uint32_t far_apart(uint32_t pp, uint32_t q, int k) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const us2 a = __builtin_bit_cast(us2, pp), b = __builtin_bit_cast(us2, q);
    const us2 d = __builtin_elementwise_max(a, b) - __builtin_elementwise_min(a, b);       // v_pk_max/min/sub_u16
    const us2 k1 = {(unsigned short)(k + 1), (unsigned short)(k + 1)};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(d, k1)) == __builtin_bit_cast(uint32_t, k1) ? 1u : 0u;
#else
    const int dh = iabs_((int)(pp & 0xFFFFu) - (int)(q & 0xFFFFu)), dv = iabs_((int)(pp >> 16) - (int)(q >> 16));
    return (dh > k && dv > k) ? 1u : 0u;
#endif
}

// fold_core: state arrays (P, Bm) + a product source.  prod(t, q, ovq) delivers product t; it is called once per t, in
// order, BEFORE anything at index >= t of the state arrays is written, so the in-place use (state and products in the
// same arrays) stays valid.  capP / capB bound the state (lane-private LDS in k_fold): returns false on overflow
// (nothing useful in `out`), true otherwise.
template <class PtrP, class PtrB, class ProdFn>
BELLA_HD bool fold_core(PtrP P, PtrB Bm, uint32_t m, ProdFn prod, uint32_t capP, uint32_t capB, int k, int binSize,
                        uint16_t* sort_scratch, FoldResult& out) {
    uint32_t nb = 1, np = 1;
    uint32_t count = 1;
    {
        uint32_t q0, ov0;
        prod(0u, q0, ov0);
        P[0] = q0;
        Bm[0] = (ov0 & 0xFFFFu) | (1u << 16);                                  // multiop: one bin, support 1
    }
    for (uint32_t t = 1; t < m; ++t) {
        uint32_t q, ovq;
        prod(t, q, ovq);
        ovq &= 0xFFFFu;
        uint32_t r = 0, w = 0, bw = 0, ins = 0;
        for (uint32_t b = 0; b < nb; ++b) {                                   // chainop, chain.hpp:109-135
            const uint32_t meta = Bm[b];
            const uint32_t bn = meta >> 16;
            const bool close = iabs_((int)(meta & 0xFFFFu) - (int)ovq) < binSize;   // :114
            if (!close) {                                                     // orphan: survives as is (:131-149)
                if (ins > 0) {
                    // an orphan bin AFTER kept positions in memory order: slide it in front of them
                    for (uint32_t x = 0; x < bn; ++x) {
                        const uint32_t tmp = P[r + x];
                        for (uint32_t y = w + x; y > w - ins + x; --y) P[y] = P[y - 1];
                        P[w - ins + x] = tmp;
                    }
                } else if (w != r) {
                    for (uint32_t x = 0; x < bn; ++x) P[w + x] = P[r + x];
                }
                w += bn;
                Bm[bw++] = meta;
            } else {                                                          // dissolved into the new head bin
                // :88-97,121: a position survives iff BOTH coordinates are strictly more than k away from the new one.
                // Branch-free compaction: every position is written at the cursor, the cursor advances only when it is
                // kept (w <= r+x, so nothing unread is clobbered).  Four independent reads per round.
                uint32_t x = 0;
                for (; x + 4 <= bn; x += 4) {
                    const uint32_t p0 = P[r + x], p1 = P[r + x + 1], p2 = P[r + x + 2], p3 = P[r + x + 3];
                    const uint32_t k0 = far_apart(p0, q, k), k1 = far_apart(p1, q, k), k2 = far_apart(p2, q, k), k3 = far_apart(p3, q, k);
                    P[w] = p0; w += k0;
                    P[w] = p1; w += k1;
                    P[w] = p2; w += k2;
                    P[w] = p3; w += k3;
                    ins += k0 + k1 + k2 + k3;
                }
                for (; x < bn; ++x) {
                    const uint32_t pp = P[r + x];
                    const uint32_t kk = far_apart(pp, q, k);
                    P[w] = pp; w += kk; ins += kk;
                }
            }
            r += bn;
        }
        if (w >= capP || bw >= capB) return false;
        P[w++] = q;                                                            // the new k-mer heads the bin
        Bm[bw++] = ovq | ((ins + 1) << 16);
        nb = bw; np = w;
        count = (count + 1 + ins) & 0xFFFFu;                                   // :104,:140 (u16)
    }
    // choose() / chain(): std::sort ids by support desc, take ids[0]  (common.h:142-170)
    uint32_t win = 0;                                                          // LOGICAL index
    out.many_bins = 0;
    if (nb <= 16 || sort_scratch == nullptr) {
        uint32_t best = 0;
        for (uint32_t l = 0; l < nb; ++l) {                                    // insertion sort => first maximum
            const uint32_t s = Bm[nb - 1 - l] >> 16;
            if (l == 0 || s > best) { best = s; win = l; }
        }
        if (nb > 16) out.many_bins = 1;
    } else {
        for (uint32_t l = 0; l < nb; ++l) sort_scratch[l] = (uint16_t)l;
        auto supf = [&](uint16_t id) -> uint32_t { return Bm[nb - 1 - id] >> 16; };
        IdSorter<decltype(supf)> srt{supf, sort_scratch};
        srt.sort((long)nb);
        win = sort_scratch[0];
    }
    const uint32_t wm = nb - 1 - win;                                          // memory index of the winner
    uint32_t end = 0;
    for (uint32_t b = 0; b <= wm; ++b) end += Bm[b] >> 16;
    out.count = (uint16_t)count;
    out.nbins = nb > 65535u ? 65535u : (uint16_t)nb;
    out.support = (uint16_t)(Bm[wm] >> 16);
    out.binov = (uint16_t)(Bm[wm] & 0xFFFFu);
    out.seed = P[end - 1];
    (void)np;
    return true;
}

// in-place form: products and state share (P, Bm)
template <class PtrP, class PtrB>
BELLA_HD void fold_pair(PtrP P, PtrB Bm, uint32_t m, int k, int binSize, uint16_t* sort_scratch, FoldResult& out) {
    auto prod = [&](uint32_t t, uint32_t& q, uint32_t& ovq) { q = P[t]; ovq = Bm[t] & 0xFFFFu; };
    (void)fold_core(P, Bm, m, prod, 0xFFFFFFFFu, 0xFFFFFFFFu, k, binSize, sort_scratch, out);
}

// order a pair's (P, Bm) entries by the product index stored in Bm's upper half (scatter order fix-up)
template <class PtrP, class PtrB>
BELLA_HD void sort_products_by_index(PtrP P, PtrB Bm, uint32_t m) {
    for (uint32_t i = 1; i < m; ++i) {
        const uint32_t kb = Bm[i], kp = P[i];
        uint32_t j = i;
        while (j > 0 && (Bm[j - 1] >> 16) > (kb >> 16)) { Bm[j] = Bm[j - 1]; P[j] = P[j - 1]; --j; }
        if (j != i) { Bm[j] = kb; P[j] = kp; }
    }
}

}  // namespace bella
