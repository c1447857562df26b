// kcount.hpp -- k-mer counting, the reliable dictionary and tuple generation on the device.
//
// Reference: SplitCount (include/kmercount.hpp:467-677) + the tuple loop of src/main.cpp:393-416.  There: every position of
// every read -> Kmer::rep(); HyperLogLog + Bloom filter + a libcuckoo table count the k-mers seen at least twice (u16,
// no saturation); those with lower <= count <= upper get an id; a second parse of the reads emits (id, read, position).
// Here, for a machine with 288 GB of HBM: all canonical k-mers of the (2-bit, device-resident) reads are written out and
// radix-sorted on their 2k significant bits (rocPRIM), run lengths give the exact multiplicities, the reliable runs form the
// dictionary (id = rank in ascending canonical order -- the reference's ids are libcuckoo's iteration order, a label), an
// open-addressing table over the dictionary serves the lookups of the tuple pass, and the tuples land in the buffers the
// assembly kernels read: nothing returns to the host.  Inputs larger than the per-pass budget are split by the top bits of the
// canonical word (an exact histogram first), so the concatenated dictionary stays sorted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "core.hpp"
#include "util.hpp"

namespace bella {

constexpr uint32_t kCountBins = 256;           // top 8 bits (or all 2k bits when k < 4) of the canonical word
constexpr uint64_t kHashEmpty = ~0ull;         // TT..T is never canonical (its twin AA..A = 0 is smaller)

__device__ __forceinline__ uint64_t canonical_word(const uint32_t* packed, uint64_t g, uint32_t k) {
    const uint64_t le = kmer_le(packed, g, k);
    const uint64_t fw = kmer_fw_from_le(le, k), rc = kmer_rc_from_le(le, k);   // Kmer::rep, Kmer.cpp:314-317
    return rc < fw ? rc : fw;
}
__device__ __forceinline__ uint32_t code_bin(uint64_t code, uint32_t k) {
    return k >= 4 ? (uint32_t)(code >> (2 * k - 8)) : (uint32_t)code;
}
__device__ __forceinline__ uint64_t mix64(uint64_t x) {                        // splitmix64 finaliser
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}

// ---- syncmer selection (-s; include/syncmer.hpp:47-79, smerlen = 5) -----------------------------------------------------------
// Kmer::hash (Kmer.cpp:304-307) = first word of MurmurHash3_x64_128 (published algorithm, kmercode/hash_funcs.c:40-140), seed 313,
// over the 8 bytes of the left-aligned 2-bit word.  Only s-mers are hashed: all 4^5 of them fit a 8 KB LDS table.
constexpr uint32_t kSmerLen = 5;
constexpr uint32_t kSmerCount = 1u << (2 * kSmerLen);
__device__ __forceinline__ uint64_t kmer_hash_left(uint64_t left_aligned) {
    auto rotl = [](uint64_t x, int r) { return (x << r) | (x >> (64 - r)); };
    auto fmix = [](uint64_t k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33; return k; };
    uint64_t h1 = 313, h2 = 313, k1 = left_aligned;
    k1 *= 0x87c37b91114253d5ull; k1 = rotl(k1, 31); k1 *= 0x4cf5ad432745937full; h1 ^= k1;
    h1 ^= 8; h2 ^= 8;
    h1 += h2; h2 += h1;
    h1 = fmix(h1); h2 = fmix(h2);
    return h1 + h2;
}
__device__ __forceinline__ void smer_table_fill(uint64_t* tab) {              // call by the whole workgroup, then __syncthreads()
    for (uint32_t v = threadIdx.x; v < kSmerCount; v += blockDim.x) tab[v] = kmer_hash_left((uint64_t)v << (64 - 2 * kSmerLen));
}
// isSyncmer on the right-aligned forward word: kept unless an interior s-mer hashes below both end s-mers
__device__ __forceinline__ bool is_syncmer(const uint64_t* tab, uint64_t fw, uint32_t k) {
    const uint32_t last = k - kSmerLen;
    const uint64_t st = tab[(fw >> (2 * last)) & (kSmerCount - 1)], en = tab[fw & (kSmerCount - 1)];
    const uint64_t lim = st < en ? st : en;
    bool ok = true;
    for (uint32_t i = 1; i < last; ++i) ok = ok && !(tab[(fw >> (2 * (last - i))) & (kSmerCount - 1)] < lim);
    return ok;
}
// the word a position contributes to the COUNT and whether it is counted at all: every position's Kmer::rep() (SplitCount), or
// the strand-specific word of the syncmer positions (SyncmerCount, kmercount.hpp:904-911)
//   mode 0 every position's rep();  mode 1 syncmers, strand-specific;  mode 2 rep() of the minimizer positions (sel != 0)
__device__ __forceinline__ bool counted_word(const uint32_t* packed, uint64_t g, uint32_t k, uint32_t mode, const uint64_t* tab, uint32_t selected,
                                             uint64_t& w) {
    if (mode != 1) { w = canonical_word(packed, g, k); return mode == 0 || selected != 0; }
    w = kmer_fw_from_le(kmer_le(packed, g, k), k);
    return is_syncmer(tab, w, k);
}

// ---- minimizer selection (-w; include/minimizer.hpp:49-79, robustwinnow = 1) -------------------------------------------------
// One lane per read runs the reference's monotone deque over the read's k-mers (order = rep().hash()) in a ring of window + 2
// entries (slot s of read r at s * nreads + r: lanes of a wavefront touch neighbouring words).  The reference's range test
// `front.first <= static_cast<int>(i) - window` is size_t arithmetic: for i < window it is always true and the deque is
// emptied, so the first `window` k-mers are never sampled; furtherPop skips a run of equal orders and one more entry is
// popped.  sel[koff[r] + j] = 1 for the sampled positions (sample() de-duplicates: the front's position never decreases).
__global__ void k_minimizer_select(const uint32_t* packed, const uint64_t* roff, const uint32_t* nk, const uint64_t* koff, uint32_t nreads,
                                   uint32_t k, uint32_t window, uint32_t cap, uint64_t* ring_ord, uint32_t* ring_pos, uint8_t* sel) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nreads) return;
    const uint64_t g0 = roff[r], o = koff[r];
    const uint32_t n = nk[r];
    uint32_t head = 0, len = 0;                                 // deque = ring slots head .. head + len - 1 (mod cap)
    uint32_t last = 0xFFFFFFFFu;
    auto slot = [&](uint32_t x) { return (uint64_t)(x >= cap ? x - cap : x) * nreads + r; };
    for (uint32_t i = 0; i < n; ++i) {
        const uint64_t ord = kmer_hash_left(canonical_word(packed, g0 + i, k) << (64 - 2 * k));
        while (len && ring_ord[slot(head + len - 1)] > ord) --len;
        { const uint64_t sl = slot(head + len); ring_ord[sl] = ord; ring_pos[sl] = i; ++len; }
        const uint64_t bound = (uint64_t)(int64_t)(int)i - (uint64_t)window;
        while (len && (uint64_t)ring_pos[slot(head)] <= bound) {
            while (len > 1 && ring_ord[slot(head)] == ring_ord[slot(head + 1)]) { head = head + 1 == cap ? 0 : head + 1; --len; }
            head = head + 1 == cap ? 0 : head + 1; --len;
        }
        if (len) {
            const uint32_t p = ring_pos[slot(head)];
            if (p != last) { last = p; sel[o + p] = 1; }
        }
    }
}

// nk[r] = k-mers of read r (kmercount.hpp:525: j = 0 .. len-k)
__global__ void k_kmers_per_read(const uint64_t* roff, uint32_t nreads, uint32_t k, uint32_t* nk) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > nreads) return;
    uint32_t v = 0;
    if (r < nreads) { const uint64_t len = roff[r + 1] - roff[r]; v = len >= k ? (uint32_t)(len - k + 1) : 0u; }
    nk[r] = v;
}

// exact histogram of the canonical words' bins: sizes the passes.  One workgroup per read.
__global__ __launch_bounds__(kBlock) void k_code_hist(const uint32_t* packed, const uint64_t* roff, const uint32_t* nk, uint32_t nreads,
                                                      uint32_t k, uint32_t mode, const uint64_t* koff, const uint8_t* sel,
                                                      unsigned long long* hist) {
    __shared__ uint32_t h[kCountBins];
    __shared__ uint64_t tab[kSmerCount];
    if (mode) { smer_table_fill(tab); __syncthreads(); }
    for (uint32_t r = blockIdx.x; r < nreads; r += gridDim.x) {
        h[threadIdx.x] = 0;
        __syncthreads();
        const uint64_t g0 = roff[r], o = koff[r];
        const uint32_t n = nk[r];
        for (uint32_t j = threadIdx.x; j < n; j += kBlock) {
            uint64_t w;
            if (counted_word(packed, g0 + j, k, mode, tab, mode == 2 ? sel[o + j] : 1u, w)) atomicAdd(&h[code_bin(w, k)], 1u);
        }
        __syncthreads();
        if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
        __syncthreads();
    }
}

// the canonical words of the bins [b0, b1) of one pass.  Single pass (cursor == nullptr): word j of read r goes to koff[r] + j.
// Otherwise the order is irrelevant (the words are sorted next): a workgroup reserves room for 1024 positions at a time.
// pb > 0: the word leaves as (word << pb) | global position index (koff[r] + j): the sort then runs on the word's bits only and every
// sorted word still knows where it came from -- the reliable ones hand their id straight to their position, no dictionary look-up.
__global__ __launch_bounds__(kBlock) void k_emit_codes(const uint32_t* packed, const uint64_t* roff, const uint32_t* nk,
                                                       const uint64_t* koff, uint32_t nreads, uint32_t k, uint32_t mode, uint32_t b0,
                                                       uint32_t b1, const uint8_t* sel, uint64_t* out, unsigned long long* cursor, uint32_t pb) {
    __shared__ uint32_t scr[kWaves];
    __shared__ unsigned long long s_base;
    __shared__ uint64_t tab[kSmerCount];
    if (mode) { smer_table_fill(tab); __syncthreads(); }
    for (uint32_t r = blockIdx.x; r < nreads; r += gridDim.x) {
        const uint64_t g0 = roff[r], o = koff[r];
        const uint32_t n = nk[r];
        if (!cursor) {
            for (uint32_t j = threadIdx.x; j < n; j += kBlock) out[o + j] = (canonical_word(packed, g0 + j, k) << pb) | (pb ? o + j : 0ull);
            continue;
        }
        for (uint32_t base = 0; base < n; base += 4 * kBlock) {
            uint64_t w[4];
            uint32_t take = 0;
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
                const uint32_t j = base + u * kBlock + threadIdx.x;
                w[u] = 0;
                if (j < n && counted_word(packed, g0 + j, k, mode, tab, mode == 2 ? sel[o + j] : 1u, w[u])) {
                    const uint32_t b = code_bin(w[u], k);
                    if (b >= b0 && b < b1) take |= 1u << u;
                }
            }
            uint32_t tot;
            const uint32_t ex = block_excl_scan<kWaves>((uint32_t)__popc(take), scr, &tot);
            if (threadIdx.x == 0) s_base = tot ? atomicAdd(cursor, (unsigned long long)tot) : 0ull;
            __syncthreads();
            uint64_t wo = s_base + ex;
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) if (take & (1u << u)) out[wo++] = (w[u] << pb) | (pb ? o + base + u * kBlock + threadIdx.x : 0ull);
            __syncthreads();
        }
    }
}

// run i of the sorted words is reliable if lower <= (length mod 65536) <= upper (kmercount.hpp:632-655: `++num` on an
// unsigned short, then the range test)
// (the syncmer counter saturates at 65535 instead: kmercount.hpp:853)
__device__ __forceinline__ uint32_t count16(uint32_t run, uint32_t saturate) { return saturate ? (run > 65535u ? 65535u : run) : (run & 0xFFFFu); }
__global__ void k_flag_reliable(const uint32_t* run_len, uint32_t nruns, uint32_t lower, uint32_t upper, uint32_t saturate, uint32_t* flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > nruns) return;
    uint32_t f = 0;
    if (i < nruns) { const uint32_t c = count16(run_len[i], saturate); f = (c >= lower && c <= upper) ? 1u : 0u; }
    flag[i] = f;
}

__global__ void k_write_dict(const uint64_t* run_code, const uint32_t* run_len, const uint32_t* flag, const uint32_t* slot, uint32_t nruns,
                             uint32_t saturate, uint64_t* dict_code, uint16_t* dict_count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nruns || !flag[i]) return;
    dict_code[slot[i]] = run_code[i];
    dict_count[slot[i]] = (uint16_t)count16(run_len[i], saturate);
}

// ---- the sorted words with their positions in the low pb bits (k_emit_codes, pb > 0) ------------------------------------------------
// The runs of equal words, tile by tile (2,048 sorted words of a workgroup in LDS + 32 words of look-ahead).  The radix sort leaves
// out the word's lowest `gb` bits when that saves it a pass (gb = 2: 32 instead of 34 bits for k = 17 -- four passes instead of five):
// the sorted array is then ordered by GROUP = word >> gb, and the up to 2^gb words of a group stand interleaved, each in position
// order.  A group is handled by the thread of its first word: it counts the group's words per value of the low bits, which gives the
// group's runs in ascending word order without moving anything.  A group belongs to the tile that holds its first word.  Two passes
// over the sorted words with a scan over the TILES in between -- no per-word flag or slot arrays:
//   k_run_count   per tile: runs that start in it (-> distinct words), reliable runs that start in it (-> ids of the tile's runs) and
//                 their words (-> the tile's place in the list of occurrences)
//   k_run_assign  per tile: the same groups again; every reliable run gets its id (tile's first id + reliable runs before it), writes
//                 its dictionary entry and appends (position, id) of every word of the run to the list of occurrences -- compact,
//                 sequential writes; one radix sort of that list by position (a fifth of the words) then IS the tuple list in the
//                 reference's order.  (Handing the id to `ids[position]` directly was 205 M random 4-byte writes into a 4 GB
//                 array at 100k reads: 12.7 ms, and three more passes over that array.)
constexpr uint32_t kRunTile = 2048, kRunHalo = 32, kRunBlock = 256, kRunGroupMax = 4;
struct RunTile {
    uint64_t w[kRunTile + kRunHalo];          // the tile's words and the look-ahead
    unsigned long long bm[3][kRunTile / 64];  // reliable runs of the group that starts at a word (0 .. 4), as three bit planes
    uint32_t pre[kRunTile / 64 + 1];          // reliable runs before each group of 64 words
    uint64_t prev;                            // the GROUP of the word before the tile (~0 for the first tile)
};
struct RunGroup {                             // what the first word of a group learns about it
    uint64_t len;                             // words of the group
    uint32_t cnt[kRunGroupMax];               // ... per value of the low bits
};
// loads the tile and finds the groups that start in it; fn(u, e, group) for every group head (e = u * 256 + thread: place in the tile)
template <class Fn>
__device__ __forceinline__ void run_tile_groups(RunTile& T, const uint64_t* s, uint64_t n, uint32_t pb, uint32_t gb, Fn&& fn) {
    const uint64_t x0 = (uint64_t)blockIdx.x * kRunTile;
    const uint32_t have = (uint32_t)(n - x0 < kRunTile + kRunHalo ? n - x0 : kRunTile + kRunHalo);
    for (uint32_t e = threadIdx.x; e < kRunTile + kRunHalo; e += kRunBlock) T.w[e] = e < have ? s[x0 + e] : ~0ull;
    if (threadIdx.x == 0) T.prev = x0 ? s[x0 - 1] >> (pb + gb) : ~0ull;
    __syncthreads();
    const uint32_t nt = have < kRunTile ? have : kRunTile;      // words of the tile itself
    const uint32_t lowmask = (1u << gb) - 1u;
#pragma unroll
    for (uint32_t u = 0; u < kRunTile / kRunBlock; ++u) {
        const uint32_t e = u * kRunBlock + threadIdx.x;
        if (e >= nt) continue;
        const uint64_t g = T.w[e] >> (pb + gb);
        const uint64_t before = e ? T.w[e - 1] >> (pb + gb) : T.prev;
        if (before == g && (e || x0)) continue;                   // not the first word of its group
        RunGroup G;
#pragma unroll
        for (uint32_t v = 0; v < kRunGroupMax; ++v) G.cnt[v] = 0;
        uint32_t len = 0;
        while (e + len < have && (T.w[e + len] >> (pb + gb)) == g) {
            const uint32_t v = (uint32_t)(T.w[e + len] >> pb) & lowmask;
            G.cnt[0] += v == 0; G.cnt[1] += v == 1; G.cnt[2] += v == 2; G.cnt[3] += v == 3;
            ++len;
        }
        G.len = len;
        if (e + len == have && x0 + have < n) {                   // the group leaves the look-ahead: its end by bisection, its words one by one
            uint64_t l = x0 + have, h = n;
            while (l < h) { const uint64_t m = (l + h) >> 1; if ((s[m] >> (pb + gb)) <= g) l = m + 1; else h = m; }
            for (uint64_t y = x0 + have; y < l; ++y) {
                const uint32_t v = (uint32_t)(s[y] >> pb) & lowmask;
                G.cnt[0] += v == 0; G.cnt[1] += v == 1; G.cnt[2] += v == 2; G.cnt[3] += v == 3;   // (u32: a word with >= 2^32 copies is beyond every limit here)
            }
            G.len = l - (x0 + e);
        }
        fn(u, e, G);
    }
}
__global__ __launch_bounds__(kRunBlock) void k_run_count(const uint64_t* s, uint64_t n, uint32_t pb, uint32_t gb, uint32_t lower, uint32_t upper,
                                                         uint32_t saturate, uint32_t* tile_rel, uint32_t* tile_heads, uint32_t* tile_words) {
    __shared__ RunTile T;
    __shared__ uint32_t s_cnt[3];
    if (threadIdx.x < 3) s_cnt[threadIdx.x] = 0;
    uint32_t rel = 0, heads = 0, words = 0;
    run_tile_groups(T, s, n, pb, gb, [&](uint32_t, uint32_t, const RunGroup& G) {
#pragma unroll
        for (uint32_t v = 0; v < kRunGroupMax; ++v) {
            const uint32_t c = count16(G.cnt[v], saturate);
            const bool r = G.cnt[v] && c >= lower && c <= upper;
            heads += G.cnt[v] ? 1u : 0u;
            rel += r ? 1u : 0u;
            words += r ? G.cnt[v] : 0u;                           // (at most the words of the pass: < 2^31)
        }
    });
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { rel += __shfl_xor(rel, d, 64); heads += __shfl_xor(heads, d, 64); words += __shfl_xor(words, d, 64); }
    if (lane_id() == 0) { if (rel) atomicAdd(&s_cnt[0], rel); if (heads) atomicAdd(&s_cnt[1], heads); if (words) atomicAdd(&s_cnt[2], words); }
    __syncthreads();
    if (threadIdx.x == 0) { tile_rel[blockIdx.x] = s_cnt[0]; tile_heads[blockIdx.x] = s_cnt[1]; tile_words[blockIdx.x] = s_cnt[2]; }
}
// tile_base[t] = reliable runs that start before tile t, tile_wbase[t] = their words (exclusive scans of k_run_count's tile_rel /
// tile_words); occ_pos / occ_id: this pass's part of the list of occurrences (the order inside a tile's part is arbitrary: the list is
// sorted by position afterwards, and positions are unique)
__global__ __launch_bounds__(kRunBlock) void k_run_assign(const uint64_t* s, uint64_t n, uint32_t pb, uint32_t gb, uint32_t lower, uint32_t upper,
                                                          uint32_t saturate, const uint32_t* tile_base, const uint32_t* tile_wbase, uint32_t id_base,
                                                          uint32_t* occ_pos, uint32_t* occ_id, uint64_t* dict_code, uint16_t* dict_count) {
    __shared__ RunTile T;
    __shared__ uint32_t s_words, s_nrec;
    // the groups with a reliable run, collected: {place in the tile | reliable runs << 16, words of the group, words of its reliable runs}.
    // One in twelve group heads has one; handled where they stand, eight rounds of 256 threads would each run as long as their
    // longest group with a handful of lanes busy (10.6 ms at 100k reads); collected, they fill the lanes of one round.
    __shared__ uint32_t s_rec[kRunTile / 2][3];                   // (a reliable group has at least two words: at most 1,024 start in a tile)
    if (threadIdx.x == 0) { s_words = 0; s_nrec = 0; }
    const uint64_t x0 = (uint64_t)blockIdx.x * kRunTile;
    uint32_t relcnt[kRunTile / kRunBlock];                        // reliable runs of my group of round u (0 .. 4)
#pragma unroll
    for (uint32_t u = 0; u < kRunTile / kRunBlock; ++u) relcnt[u] = 0;
    run_tile_groups(T, s, n, pb, gb, [&](uint32_t u, uint32_t e, const RunGroup& G) {
        uint32_t mk = 0, words = 0;
#pragma unroll
        for (uint32_t v = 0; v < kRunGroupMax; ++v) {
            const uint32_t c = count16(G.cnt[v], saturate);
            const bool r = G.cnt[v] && c >= lower && c <= upper;
            mk |= r ? 1u << v : 0u;
            words += r ? G.cnt[v] : 0u;
        }
        relcnt[u] = (uint32_t)__popc(mk);
        // (all group heads of a round arrive here together: one LDS atomic for the lanes that have something to append)
        const unsigned long long bal = __ballot(mk != 0u);
        if (bal) {
            const int leader = __ffsll((long long)bal) - 1;
            uint32_t at = 0;
            if ((int)lane_id() == leader) at = atomicAdd(&s_nrec, (uint32_t)__popcll(bal));
            at = (uint32_t)__shfl((int)at, leader, 64) + (uint32_t)__popcll(bal & ((1ull << lane_id()) - 1ull));
            if (mk) { s_rec[at][0] = e | (mk << 16); s_rec[at][1] = (uint32_t)(G.len > 0xFFFFFFFFull ? 0xFFFFFFFFull : G.len); s_rec[at][2] = words; }
        }
    });
#pragma unroll
    for (uint32_t u = 0; u < kRunTile / kRunBlock; ++u) {          // words u * 256 + w * 64 + lane: three ballots per group of 64 words
        const uint32_t nr = relcnt[u];
#pragma unroll
        for (uint32_t b = 0; b < 3; ++b) {
            const unsigned long long m = __ballot((nr >> b) & 1u);
            if (lane_id() == 0) T.bm[b][u * (kRunBlock / 64) + wave_id()] = m;
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {                                        // reliable runs before each group of 64 words (32 groups: one wavefront)
        uint32_t c = 0;
        if (threadIdx.x < kRunTile / 64)
            c = (uint32_t)__popcll(T.bm[0][threadIdx.x]) + 2u * (uint32_t)__popcll(T.bm[1][threadIdx.x]) + 4u * (uint32_t)__popcll(T.bm[2][threadIdx.x]);
        const uint32_t inc = wave_incl_scan(c);
        if (threadIdx.x < kRunTile / 64) T.pre[threadIdx.x] = inc - c;
    }
    __syncthreads();
    const uint32_t base = tile_base[blockIdx.x];
    const uint32_t wbase = tile_wbase[blockIdx.x];
    const uint64_t pmask = (1ull << pb) - 1ull;
    const uint32_t lowmask = (1u << gb) - 1u;
    const uint32_t have = (uint32_t)(n - x0 < kRunTile + kRunHalo ? n - x0 : kRunTile + kRunHalo);
    const uint32_t nrec = s_nrec;
    for (uint32_t r0 = 0; r0 < nrec; r0 += kRunBlock) {
        const uint32_t r = r0 + threadIdx.x;
        const bool valid = r < nrec;
        const uint32_t r0w = valid ? s_rec[r][0] : 0u, glen = valid ? s_rec[r][1] : 0u, mywords = valid ? s_rec[r][2] : 0u;
        // this group's place in the tile's part of the list: one LDS atomic per wavefront (a same-address atomic per lane serialises)
        const uint32_t inc = wave_incl_scan(mywords);
        uint32_t w0 = 0;
        if (lane_id() == 63 && inc) w0 = atomicAdd(&s_words, inc);
        w0 = (uint32_t)__shfl((int)w0, 63, 64);
        if (!valid) continue;
        const uint32_t e = r0w & 0xFFFFu, mk = r0w >> 16;
        const uint32_t q = e / 64;
        const unsigned long long below = (1ull << (e & 63u)) - 1ull;
        const uint32_t first = base + T.pre[q] + (uint32_t)__popcll(T.bm[0][q] & below) + 2u * (uint32_t)__popcll(T.bm[1][q] & below) +
                               4u * (uint32_t)__popcll(T.bm[2][q] & below);
        const uint64_t g = T.w[e] >> (pb + gb);
        uint32_t idv[kRunGroupMax], cnt[kRunGroupMax];            // local id of the group's run with low bits v (if reliable), its words
        uint32_t nxt = first;
#pragma unroll
        for (uint32_t v = 0; v < kRunGroupMax; ++v) { idv[v] = nxt; cnt[v] = 0; nxt += (mk >> v) & 1u; }
        uint32_t w = wbase + w0 + inc - mywords;
        for (uint32_t m2 = 0; m2 < glen; ++m2) {                   // (a reliable run is at most `upper` words unless its 16-bit count wrapped)
            const uint64_t wv = e + m2 < have ? T.w[e + m2] : s[x0 + e + m2];
            const uint32_t v = (uint32_t)(wv >> pb) & lowmask;
            if (!((mk >> v) & 1u)) continue;
            const uint32_t id = v == 0 ? idv[0] : v == 1 ? idv[1] : v == 2 ? idv[2] : idv[3];
            cnt[0] += v == 0; cnt[1] += v == 1; cnt[2] += v == 2; cnt[3] += v == 3;
            occ_pos[w] = (uint32_t)(wv & pmask);
            occ_id[w] = id_base + id;
            ++w;
        }
#pragma unroll
        for (uint32_t v = 0; v < kRunGroupMax; ++v)
            if ((mk >> v) & 1u) { dict_code[idv[v]] = (g << gb) | v; dict_count[idv[v]] = (uint16_t)count16(cnt[v], saturate); }
    }
}

// the list of occurrences sorted by position: where the tuples of each read of the block begin (first[r] = the first occurrence at or
// after the read's first position, r = 0 .. nreads) and, relative to the block's first tuple, tstart
__global__ void k_occ_bounds(const uint32_t* spos, uint64_t nocc, const uint64_t* koff, uint32_t nreads, uint32_t* first, uint64_t* tstart) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > nreads) return;
    auto lower = [&](uint64_t key) -> uint64_t { uint64_t l = 0, h = nocc; while (l < h) { const uint64_t m = (l + h) >> 1; if ((uint64_t)spos[m] < key) l = m + 1; else h = m; } return l; };
    const uint64_t me = lower(koff[r]);
    first[r] = (uint32_t)me;
    tstart[r] = me - lower(koff[0]);
}
// tuples (id, read, position) from the sorted list: the reference's generation order -- read by read, positions ascending (main.cpp:393-416)
__global__ __launch_bounds__(kBlock) void k_write_tuples_sorted(const uint32_t* spos, const uint32_t* sid, const uint64_t* koff, const uint32_t* first,
                                                                uint32_t nreads, uint32_t read_base, uint32_t* t_kmer, uint32_t* t_read, uint16_t* t_pos) {
    for (uint32_t r = blockIdx.x; r < nreads; r += gridDim.x) {
        const uint32_t f0 = first[0], lo = first[r], hi = first[r + 1];
        const uint64_t o = koff[r];
        for (uint32_t i = lo + threadIdx.x; i < hi; i += kBlock) {
            t_kmer[i - f0] = sid[i]; t_read[i - f0] = read_base + r; t_pos[i - f0] = (uint16_t)((uint64_t)spos[i] - o);
        }
    }
}

// countsreliable (a CuckooDict in the reference, main.cpp:410 `find`): open addressing over the dictionary, value = id
__global__ void k_hash_fill(uint64_t* hkey, uint64_t slots) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < slots) hkey[i] = kHashEmpty;
}
__global__ void k_hash_build(const uint64_t* dict_code, uint32_t nk, uint64_t* hkey, uint32_t* hval, uint64_t mask) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nk) return;
    const uint64_t code = dict_code[i];
    if (code == kHashEmpty) return;             // TT..T (k = 32, strand-specific dictionary): never the canonical word of a position
    uint64_t h = mix64(code) & mask;
    for (;;) {
        const unsigned long long old = atomicCAS((unsigned long long*)&hkey[h], (unsigned long long)kHashEmpty, (unsigned long long)code);
        if (old == kHashEmpty) { hval[h] = i; break; }
        h = (h + 1) & mask;
    }
}

// tuple pass 1 (main.cpp:393-416): the id of every position's k-mer (0xFFFFFFFF: not reliable) and the tuples per read
__global__ __launch_bounds__(kBlock) void k_lookup_ids(const uint32_t* packed, const uint64_t* roff, const uint32_t* nk, const uint64_t* koff,
                                                       uint32_t nreads, uint32_t k, const uint64_t* hkey, const uint32_t* hval, uint64_t mask,
                                                       const uint8_t* sel, uint32_t* ids, uint32_t* found_per_read) {
    __shared__ uint32_t s_cnt;
    for (uint32_t r = blockIdx.x; r < nreads; r += gridDim.x) {
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        const uint64_t g0 = roff[r], o = koff[r];
        const uint32_t n = nk[r];
        uint32_t mine = 0;
        for (uint32_t j = threadIdx.x; j < n; j += kBlock) {
            const uint64_t code = canonical_word(packed, g0 + j, k);
            uint64_t h = mix64(code) & mask;
            uint32_t id = 0xFFFFFFFFu;
            if (sel && !sel[o + j]) { ids[o + j] = id; continue; }     // -w: only the minimizer positions make tuples (main.cpp:363-388)
            for (;;) {
                const uint64_t kk = hkey[h];
                if (kk == code) { id = hval[h]; break; }
                if (kk == kHashEmpty) break;
                h = (h + 1) & mask;
            }
            ids[o + j] = id;
            mine += (id != 0xFFFFFFFFu);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) mine += __shfl_xor(mine, d, 64);
        if (lane_id() == 0 && mine) atomicAdd(&s_cnt, mine);
        __syncthreads();
        if (threadIdx.x == 0) found_per_read[r] = s_cnt;
        __syncthreads();
    }
}

// tuple pass 2: (id, read, position) in the reference's generation order -- read by read, positions ascending
// (read_base: the reads handed in are the block read_base .. read_base + nreads - 1 of the read set; t_read holds absolute ids)
__global__ __launch_bounds__(kBlock) void k_write_tuples(const uint32_t* ids, const uint32_t* nk, const uint64_t* koff, const uint64_t* tstart,
                                                         uint32_t nreads, uint32_t read_base, uint32_t* t_kmer, uint32_t* t_read, uint16_t* t_pos) {
    __shared__ uint32_t scr[kWaves];
    for (uint32_t r = blockIdx.x; r < nreads; r += gridDim.x) {
        const uint64_t o = koff[r];
        const uint32_t n = nk[r];
        uint64_t w = tstart[r];
        for (uint32_t base = 0; base < n; base += kBlock) {
            const uint32_t j = base + threadIdx.x;
            const uint32_t id = j < n ? ids[o + j] : 0xFFFFFFFFu;
            const uint32_t f = id != 0xFFFFFFFFu ? 1u : 0u;
            uint32_t tot;
            const uint32_t ex = block_excl_scan<kWaves>(f, scr, &tot);
            if (f) { t_kmer[w + ex] = id; t_read[w + ex] = read_base + r; t_pos[w + ex] = (uint16_t)j; }
            w += tot;
        }
    }
}

}  // namespace bella
