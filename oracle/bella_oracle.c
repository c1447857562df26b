/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the BELLA overlap hot path.
 *
 * A plain-C restatement of the reference's algorithm (PASSIONLab/BELLA, paths relative to
 * /root/reference).  It is the checker for the HIP path; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The product (bella_amd/) never links or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks this file against (a) golden .out files produced
 * by the reference binary at OMP_NUM_THREADS=1 (tests/golden/, made by oracle/make_golden.py), (b) the
 * reference's own code called in-process through oracle/_ref/libbella_ref.so (HashSpGEMM, xavierAlign)
 * when that library is present, and (c) the xavier/demo.cpp known answer.
 *
 * Every function cites the reference file:line it restates.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define EMPTY32 0xFFFFFFFFu

/* ---------------------------------------------------------------------------------------------------
 * K-mer counting, reliable dictionary, tuple generation
 * ------------------------------------------------------------------------------------------------- */

/* Kmer::set_kmer (kmercode/Kmer.cpp:205-228): A,C,G,T = 0,1,2,3, first base in the most significant bits, so
 * Kmer::operator< (Kmer.cpp:160-169) is the lexicographic order of the strings; Kmer::rep (Kmer.cpp:314-317) = the smaller
 * of the k-mer and its reverse complement (Kmer::twin, Kmer.cpp:324-355).  Returns the canonical word, right-aligned. */
static uint64_t oracle_canonical(const char* s, uint32_t k) {
    uint64_t fw = 0, rc = 0;
    for (uint32_t i = 0; i < k; ++i) {
        const uint64_t x = ((uint64_t)(s[i] & 4)) >> 1;
        const uint64_t code = x + ((x ^ (uint64_t)(s[i] & 2)) >> 1);            /* Kmer.cpp:215-216 */
        fw = (fw << 2) | code;
        rc |= (3 - code) << (2 * i);
    }
    return rc < fw ? rc : fw;
}

static int cmp_u64(const void* a, const void* b) {
    const uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* SplitCount (include/kmercount.hpp:467-677): every position j <= len-k of every read contributes rep(kmer) (:525-537); a
 * k-mer enters the table at its second sighting (Bloom filter, :606-613) and the second pass counts EVERY occurrence in an
 * unsigned short without saturation (:632-641, `++num`): count = occurrences mod 65536 (a one-off k-mer that slips through
 * the Bloom filter ends with count 1 -- never reliable for lower >= 2; for lower <= 1 the reference's set would depend on
 * Bloom false positives, callers use lower >= 2).  Reliable: lower <= count <= upper (:648-655).
 * The reference numbers the reliable k-mers in libcuckoo's bucket order (:645-656), which depends on the table's insert
 * history and thread count; the numbering is a LABEL: here id = rank in ascending canonical order.
 * Tuple generation (src/main.cpp:393-416): read by read, positions ascending, one tuple (id, read, j) per occurrence of a
 * reliable k-mer.
 * seqs[r] has lens[r] bases.  dict_codes/dict_counts need room for the distinct k-mers, tuples for the positions; pass NULL
 * to only count.  Returns the number of reliable k-mers; *ntuples and *ndistinct are set. */
int64_t oracle_count_kmers(uint32_t nreads, const char* const* seqs, const uint32_t* lens, uint32_t k, uint32_t lower,
                           uint32_t upper, uint64_t* dict_codes, uint16_t* dict_counts, uint32_t* t_kmer, uint32_t* t_read,
                           uint16_t* t_pos, uint64_t* ntuples, uint64_t* ndistinct) {
    uint64_t total = 0;
    for (uint32_t r = 0; r < nreads; ++r) if (lens[r] >= k) total += lens[r] - k + 1;
    uint64_t* all = (uint64_t*)malloc(sizeof(uint64_t) * (total ? total : 1));
    uint64_t n = 0;
    for (uint32_t r = 0; r < nreads; ++r)
        for (uint32_t j = 0; j + k <= lens[r]; ++j) all[n++] = oracle_canonical(seqs[r] + j, k);
    qsort(all, n, sizeof(uint64_t), cmp_u64);
    uint64_t* rel = (uint64_t*)malloc(sizeof(uint64_t) * (n ? n : 1));
    uint64_t nrel = 0, ndist = 0;
    for (uint64_t i = 0; i < n;) {
        uint64_t j = i;
        while (j < n && all[j] == all[i]) ++j;
        const uint32_t cnt = (uint32_t)((j - i) & 0xFFFFu);                     /* unsigned short, no saturation */
        ndist++;
        if (cnt >= lower && cnt <= upper) {
            if (dict_codes) dict_codes[nrel] = all[i];
            if (dict_counts) dict_counts[nrel] = (uint16_t)cnt;
            rel[nrel++] = all[i];
        }
        i = j;
    }
    uint64_t nt = 0;
    for (uint32_t r = 0; r < nreads; ++r)
        for (uint32_t j = 0; j + k <= lens[r]; ++j) {
            const uint64_t c = oracle_canonical(seqs[r] + j, k);
            uint64_t lo = 0, hi = nrel;                                        /* countsreliable.find (main.cpp:410) */
            while (lo < hi) { const uint64_t mid = (lo + hi) / 2; if (rel[mid] < c) lo = mid + 1; else hi = mid; }
            if (lo < nrel && rel[lo] == c) {
                if (t_kmer) { t_kmer[nt] = (uint32_t)lo; t_read[nt] = r; t_pos[nt] = (uint16_t)j; }
                nt++;
            }
        }
    free(all); free(rel);
    if (ntuples) *ntuples = nt;
    if (ndistinct) *ndistinct = ndist;
    return (int64_t)nrel;
}

/* Kmer::hash (kmercode/Kmer.cpp:304-307) = MurmurHash3_x64_64 (kmercode/hash_funcs.c:135-140): the first word of
 * MurmurHash3_x64_128 (the published algorithm, hash_funcs.c:40-128) with seed 313 over the N_BYTES = 8 bytes of the k-mer,
 * i.e. over ONE little-endian u64 = the left-aligned 2-bit word (Kmer.hpp:27-28, MAX_KMER_SIZEK 32). */
static uint64_t rotl64_(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static uint64_t fmix64_(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return k;
}
uint64_t oracle_kmer_hash(uint64_t left_aligned_word) {
    uint64_t h1 = 313, h2 = 313, k1 = left_aligned_word;
    k1 *= 0x87c37b91114253d5ULL; k1 = rotl64_(k1, 31); k1 *= 0x4cf5ad432745937fULL; h1 ^= k1;     /* tail, len & 15 == 8 */
    h1 ^= 8; h2 ^= 8;
    h1 += h2; h2 += h1;
    h1 = fmix64_(h1); h2 = fmix64_(h2);
    h1 += h2;
    return h1;
}

static uint64_t oracle_forward(const char* s, uint32_t k) {                      /* right-aligned forward word */
    uint64_t fw = 0;
    for (uint32_t i = 0; i < k; ++i) {
        const uint64_t x = ((uint64_t)(s[i] & 4)) >> 1;
        fw = (fw << 2) | (x + ((x ^ (uint64_t)(s[i] & 2)) >> 1));
    }
    return fw;
}

/* isSyncmer (include/syncmer.hpp:47-68), smerlen = 5: the k-mer is kept unless an interior s-mer (offsets 1 .. k-s-1) hashes
 * below BOTH the first and the last s-mer. */
int oracle_is_syncmer(const char* s, uint32_t k) {
    const uint32_t sl = 5;
    const uint64_t st = oracle_kmer_hash(oracle_forward(s, sl) << (64 - 2 * sl));
    const uint64_t en = oracle_kmer_hash(oracle_forward(s + k - sl, sl) << (64 - 2 * sl));
    for (uint32_t i = 1; i + sl < k; ++i) {
        const uint64_t h = oracle_kmer_hash(oracle_forward(s + i, sl) << (64 - 2 * sl));
        if (h < st && h < en) return 0;
    }
    return 1;
}

/* -s mode: SyncmerCount (include/kmercount.hpp:845-985) + the tuple loop of src/main.cpp:393-416 (which has no syncmer branch).
 * Counted: the STRAND-SPECIFIC string of every position that is a syncmer (kmercount.hpp:904-911, no rep()), in an unsigned
 * short that saturates at 65535 (:853).  Reliable: lower <= count <= upper (:958-966).  Tuples: EVERY position whose
 * canonical k-mer (rep) is a key of that dictionary (main.cpp:399-415) -- so a dictionary entry that is not its own
 * canonical form is never hit, and positions that are not syncmers themselves still match.  ids: ascending order of the
 * (forward) dictionary words.  Same outputs as oracle_count_kmers. */
int64_t oracle_count_syncmers(uint32_t nreads, const char* const* seqs, const uint32_t* lens, uint32_t k, uint32_t lower,
                              uint32_t upper, uint64_t* dict_codes, uint16_t* dict_counts, uint32_t* t_kmer, uint32_t* t_read,
                              uint16_t* t_pos, uint64_t* ntuples, uint64_t* ndistinct) {
    uint64_t total = 0;
    for (uint32_t r = 0; r < nreads; ++r) if (lens[r] >= k) total += lens[r] - k + 1;
    uint64_t* all = (uint64_t*)malloc(sizeof(uint64_t) * (total ? total : 1));
    uint64_t n = 0;
    for (uint32_t r = 0; r < nreads; ++r)
        for (uint32_t j = 0; j + k <= lens[r]; ++j)
            if (k > 5 && oracle_is_syncmer(seqs[r] + j, k)) all[n++] = oracle_forward(seqs[r] + j, k);
    qsort(all, n, sizeof(uint64_t), cmp_u64);
    uint64_t* rel = (uint64_t*)malloc(sizeof(uint64_t) * (n ? n : 1));
    uint64_t nrel = 0, ndist = 0;
    for (uint64_t i = 0; i < n;) {
        uint64_t j = i;
        while (j < n && all[j] == all[i]) ++j;
        const uint64_t run = j - i;
        const uint32_t cnt = run > 65535 ? 65535u : (uint32_t)run;              /* saturating updatefn */
        ndist++;
        if (cnt >= lower && cnt <= upper) {
            if (dict_codes) dict_codes[nrel] = all[i];
            if (dict_counts) dict_counts[nrel] = (uint16_t)cnt;
            rel[nrel++] = all[i];
        }
        i = j;
    }
    uint64_t nt = 0;
    for (uint32_t r = 0; r < nreads; ++r)
        for (uint32_t j = 0; j + k <= lens[r]; ++j) {
            const uint64_t c = oracle_canonical(seqs[r] + j, k);
            uint64_t lo = 0, hi = nrel;
            while (lo < hi) { const uint64_t mid = (lo + hi) / 2; if (rel[mid] < c) lo = mid + 1; else hi = mid; }
            if (lo < nrel && rel[lo] == c) {
                if (t_kmer) { t_kmer[nt] = (uint32_t)lo; t_read[nt] = r; t_pos[nt] = (uint16_t)j; }
                nt++;
            }
        }
    free(all); free(rel);
    if (ntuples) *ntuples = nt;
    if (ndistinct) *ndistinct = ndist;
    return (int64_t)nrel;
}

/* getMinimizers (include/minimizer.hpp:49-79) with robustwinnow = 1 (:10): a monotone deque of (position, order), order =
 * rep().hash() (:23-26).  Restated with the reference's integer conversions: the range test `front.first <=
 * static_cast<int>(i) - window` (:64) is evaluated in size_t, so for i < window the right side wraps to a huge value and the
 * deque is emptied at every step -- the first `window` k-mers never become minimizers.  furtherPop (:12-21) skips a run of
 * equal orders at the front, then one more entry is popped (:67) even if it is still in range.  sample (:28-32) appends the
 * front's position unless it is the last one appended.  sel[j] = 1 for the sampled positions; returns their number. */
static uint64_t oracle_minimizers(const char* seq, uint32_t len, uint32_t k, uint32_t window, uint8_t* sel) {
    if (len < k) return 0;
    const uint32_t n = len - k + 1;
    int64_t* dq_pos = (int64_t*)malloc(sizeof(int64_t) * (n + 1));
    uint64_t* dq_ord = (uint64_t*)malloc(sizeof(uint64_t) * (n + 1));
    uint64_t head = 0, tail = 0, count = 0;                                    /* deque = [head, tail) */
    int64_t last = -1;
    memset(sel, 0, n);
    for (uint32_t i = 0; i < n; ++i) {
        const uint64_t ord = oracle_kmer_hash(oracle_canonical(seq + i, k) << (64 - 2 * k));
        while (tail > head && dq_ord[tail - 1] > ord) --tail;                   /* :57-60 */
        dq_pos[tail] = i; dq_ord[tail] = ord; ++tail;                           /* :61 */
        while (tail > head && (uint64_t)dq_pos[head] <= (uint64_t)((int64_t)(int)i) - (uint64_t)window) {   /* :63 */
            while (tail - head > 1 && dq_ord[head] == dq_ord[head + 1]) ++head; /* furtherPop */
            ++head;                                                             /* :66 */
        }
        if (tail > head && dq_pos[head] != last) {                             /* :69-72 sample */
            last = dq_pos[head];
            sel[last] = 1;
            ++count;
        }
    }
    free(dq_pos); free(dq_ord);
    return count;
}

/* -w mode: MinimizerCount (include/kmercount.hpp:691-835) + the minimizer branch of the tuple loop (src/main.cpp:363-388).
 * Counted: rep() of every minimizer position (:744-748), unsigned short saturating at 65535 (:699); reliable: lower <= count
 * <= upper; tuples: the minimizer positions whose rep() is reliable, read by read, positions ascending.  ids: ascending
 * canonical order.  Same outputs as oracle_count_kmers. */
int64_t oracle_count_minimizers(uint32_t nreads, const char* const* seqs, const uint32_t* lens, uint32_t k, uint32_t window,
                                uint32_t lower, uint32_t upper, uint64_t* dict_codes, uint16_t* dict_counts, uint32_t* t_kmer,
                                uint32_t* t_read, uint16_t* t_pos, uint64_t* ntuples, uint64_t* ndistinct) {
    uint64_t total = 0, maxlen = 1;
    for (uint32_t r = 0; r < nreads; ++r) { if (lens[r] >= k) total += lens[r] - k + 1; if (lens[r] > maxlen) maxlen = lens[r]; }
    uint64_t* all = (uint64_t*)malloc(sizeof(uint64_t) * (total ? total : 1));
    uint8_t* sel = (uint8_t*)malloc(maxlen + 1);
    uint64_t n = 0;
    for (uint32_t r = 0; r < nreads; ++r) {
        oracle_minimizers(seqs[r], lens[r], k, window, sel);
        for (uint32_t j = 0; j + k <= lens[r]; ++j) if (sel[j]) all[n++] = oracle_canonical(seqs[r] + j, k);
    }
    qsort(all, n, sizeof(uint64_t), cmp_u64);
    uint64_t* rel = (uint64_t*)malloc(sizeof(uint64_t) * (n ? n : 1));
    uint64_t nrel = 0, ndist = 0;
    for (uint64_t i = 0; i < n;) {
        uint64_t j = i;
        while (j < n && all[j] == all[i]) ++j;
        const uint64_t run = j - i;
        const uint32_t cnt = run > 65535 ? 65535u : (uint32_t)run;
        ndist++;
        if (cnt >= lower && cnt <= upper) {
            if (dict_codes) dict_codes[nrel] = all[i];
            if (dict_counts) dict_counts[nrel] = (uint16_t)cnt;
            rel[nrel++] = all[i];
        }
        i = j;
    }
    uint64_t nt = 0;
    for (uint32_t r = 0; r < nreads; ++r) {
        oracle_minimizers(seqs[r], lens[r], k, window, sel);
        for (uint32_t j = 0; j + k <= lens[r]; ++j) {
            if (!sel[j]) continue;
            const uint64_t c = oracle_canonical(seqs[r] + j, k);
            uint64_t lo = 0, hi = nrel;
            while (lo < hi) { const uint64_t mid = (lo + hi) / 2; if (rel[mid] < c) lo = mid + 1; else hi = mid; }
            if (lo < nrel && rel[lo] == c) {
                if (t_kmer) { t_kmer[nt] = (uint32_t)lo; t_read[nt] = r; t_pos[nt] = (uint16_t)j; }
                nt++;
            }
        }
    }
    free(all); free(rel); free(sel);
    if (ntuples) *ntuples = nt;
    if (ndistinct) *ndistinct = ndist;
    return (int64_t)nrel;
}

/* ---------------------------------------------------------------------------------------------------
 * Operand assembly
 * ------------------------------------------------------------------------------------------------- */

static uint32_t pow2_at_least(uint32_t minsz, uint64_t n) { /* CSC.cpp:322-326, overlap.hpp:230-234,291-295 */
    uint64_t s = minsz;
    while (s < n) s <<= 1;
    return (uint32_t)s;
}

/* B = transpmat: CSC tuple constructor (src/CSC.cpp:422-479, needsort=false) followed by
 * MergeDuplicates (src/CSC.cpp:301-420, issorted=false) with the "keep p1" lambda of
 * src/main.cpp:477-480 (the LATER position of a duplicated k-mer wins, CSC.cpp:344).
 * Tuples are (kmer, read, pos) in generation order.  colptr[nreads+1]; rowids/values sized ntuples.
 * Returns nnz after the merge. */
int64_t oracle_build_B(uint32_t nreads, uint64_t ntuples, const uint32_t* t_kmer, const uint32_t* t_read,
                       const uint16_t* t_pos, uint32_t* colptr, uint32_t* rowids, uint16_t* values) {
    uint32_t* cnt = (uint32_t*)calloc((size_t)nreads + 1, sizeof(uint32_t));
    uint32_t* start = (uint32_t*)calloc((size_t)nreads + 1, sizeof(uint32_t));
    uint32_t* r0 = (uint32_t*)malloc(sizeof(uint32_t) * (ntuples ? ntuples : 1));
    uint16_t* v0 = (uint16_t*)malloc(sizeof(uint16_t) * (ntuples ? ntuples : 1));
    for (uint64_t t = 0; t < ntuples; ++t) cnt[t_read[t]]++;                 /* CSC.cpp:433-437 */
    for (uint32_t r = 0; r < nreads; ++r) start[r + 1] = start[r] + cnt[r];  /* CumulativeSum :441 */
    memset(cnt, 0, sizeof(uint32_t) * nreads);
    for (uint64_t t = 0; t < ntuples; ++t) {                                 /* CSC.cpp:466-472 (stable) */
        uint32_t c = t_read[t];
        r0[start[c] + cnt[c]] = t_kmer[t];
        v0[start[c] + cnt[c]] = t_pos[t];
        cnt[c]++;
    }
    uint64_t nnz = 0;
    colptr[0] = 0;
    uint32_t cap = 16;
    uint32_t* hk = (uint32_t*)malloc(sizeof(uint32_t) * cap);
    uint16_t* hv = (uint16_t*)malloc(sizeof(uint16_t) * cap);
    for (uint32_t i = 0; i < nreads; ++i) {                                  /* CSC.cpp:316-375 */
        uint32_t n = start[i + 1] - start[i];
        uint32_t ht = pow2_at_least(16, n);
        if (ht > cap) {
            cap = ht;
            hk = (uint32_t*)realloc(hk, sizeof(uint32_t) * cap);
            hv = (uint16_t*)realloc(hv, sizeof(uint16_t) * cap);
        }
        for (uint32_t j = 0; j < ht; ++j) hk[j] = EMPTY32;
        for (uint32_t j = start[i]; j < start[i + 1]; ++j) {
            uint32_t key = r0[j];
            uint32_t h = (key * 107u) & (ht - 1);
            for (;;) {
                if (hk[h] == key) { hv[h] = v0[j]; break; }       /* addop(values[j], old) -> values[j] */
                if (hk[h] == EMPTY32) { hk[h] = key; hv[h] = v0[j]; break; }
                h = (h + 1) & (ht - 1);
            }
        }
        for (uint32_t j = 0; j < ht; ++j)                                    /* slot order, CSC.cpp:358-373 */
            if (hk[j] != EMPTY32) { rowids[nnz] = hk[j]; values[nnz] = hv[j]; nnz++; }
        colptr[i + 1] = (uint32_t)nnz;
    }
    free(hk); free(hv); free(cnt); free(start); free(r0); free(v0);
    return (int64_t)nnz;
}

/* A = B.Transpose() (src/CSC.cpp:289-299 -> include/common/transpose.h:13-52).  With one thread the
 * atomic counter hands out slots in row order, i.e. every k-mer column lists its reads ASCENDING. */
void oracle_transpose(uint32_t nreads, uint32_t nkmers, const uint32_t* Bcolptr, const uint32_t* Browids,
                      const uint16_t* Bvalues, uint32_t* Acolptr, uint32_t* Arowids, uint16_t* Avalues) {
    uint64_t nnz = Bcolptr[nreads];
    memset(Acolptr, 0, sizeof(uint32_t) * ((size_t)nkmers + 1));
    for (uint64_t e = 0; e < nnz; ++e) Acolptr[Browids[e] + 1]++;
    for (uint32_t c = 0; c < nkmers; ++c) Acolptr[c + 1] += Acolptr[c];
    uint32_t* fill = (uint32_t*)calloc((size_t)nkmers + 1, sizeof(uint32_t));
    for (uint32_t r = 0; r < nreads; ++r)
        for (uint32_t e = Bcolptr[r]; e < Bcolptr[r + 1]; ++e) {
            uint32_t c = Browids[e];
            uint32_t loc = Acolptr[c] + fill[c]++;
            Arowids[loc] = r;
            Avalues[loc] = Bvalues[e];
        }
    free(fill);
}

/* ---------------------------------------------------------------------------------------------------
 * The semiring value (include/common/common.h:119-183) and its operations (include/chain.hpp)
 * ------------------------------------------------------------------------------------------------- */
typedef struct { uint16_t h, v; } opos_t;
typedef struct { uint16_t overlap; uint32_t npos, cap; opos_t* pos; } obin_t;  /* support == npos always */
typedef struct { uint16_t count; uint32_t nbins, cap; obin_t* bins; } oval_t;

static void bin_push(obin_t* b, opos_t p) {
    if (b->npos == b->cap) { b->cap = b->cap ? b->cap * 2 : 4; b->pos = (opos_t*)realloc(b->pos, sizeof(opos_t) * b->cap); }
    b->pos[b->npos++] = p;
}
static void val_free(oval_t* v) {
    for (uint32_t b = 0; b < v->nbins; ++b) free(v->bins[b].pos);
    free(v->bins);
    v->bins = NULL; v->nbins = v->cap = 0;
}

/* chain.hpp:35-44 checkstrand + :47-71 overlapop.  read1 = H (row key), read2 = V (column i).
 * All the u16 truncations of the reference are kept. */
static int o_overlapop(const char* read1, uint32_t len1, const char* read2, uint32_t len2, uint16_t begpH,
                       uint16_t begpV, uint16_t k) {
    int read1len = (int)len1, read2len = (int)len2;
    int oriented = memcmp(read1 + begpH, read2 + begpV, k) == 0;              /* :35-44 */
    if (!oriented) begpH = (uint16_t)(len1 - begpH - k);                      /* :57-60 */
    uint16_t endpH = (uint16_t)(begpH + k), endpV = (uint16_t)(begpV + k);    /* :63-64 */
    int margin1 = begpH < begpV ? begpH : begpV;                              /* :66 */
    int m2a = read1len - endpH, m2b = read2len - endpV;
    int margin2 = m2a < m2b ? m2a : m2b;                                      /* :67 */
    return margin1 + margin2 + k;                                             /* :68 */
}
int oracle_overlapop(const char* read1, uint32_t len1, const char* read2, uint32_t len2, uint16_t begpH,
                     uint16_t begpV, uint16_t k) {
    return o_overlapop(read1, len1, read2, len2, begpH, begpV, k);
}

static int iabs(int x) { return x < 0 ? -x : x; }

/* chain.hpp:74-86 multiop builds the singleton; chain.hpp:100-150 chainop is called as
 * addop(result_new, slot_old) (overlap.hpp:326, main.cpp:514-524) and the NEW object survives.
 * `S` is the accumulated value and is updated in place to the new value. */
static void o_fold(oval_t* S, int first, opos_t q, uint16_t ovq, int binSize, int k) {
    if (first) {                                                               /* multiop */
        S->count = 1; S->nbins = 1; S->cap = 2;
        S->bins = (obin_t*)calloc(S->cap, sizeof(obin_t));
        S->bins[0].overlap = ovq;
        bin_push(&S->bins[0], q);
        return;
    }
    obin_t head; memset(&head, 0, sizeof(head));
    head.overlap = ovq;
    bin_push(&head, q);
    uint16_t count = (uint16_t)(1 + S->count);                                /* :104 */
    obin_t* nb = (obin_t*)calloc((size_t)S->nbins + 1, sizeof(obin_t));
    uint32_t nn = 1;
    uint32_t inserted = 0;
    for (uint32_t i = 0; i < S->nbins; ++i) {                                 /* :109-135 (m1 has ONE bin) */
        obin_t* b = &S->bins[i];
        if (iabs((int)b->overlap - (int)ovq) < binSize) {                     /* :114 */
            for (uint32_t x = 0; x < b->npos; ++x) {                          /* :116-126 */
                int dh = iabs((int)q.h - (int)b->pos[x].h), dv = iabs((int)q.v - (int)b->pos[x].v);
                if (dh > k && dv > k) { bin_push(&head, b->pos[x]); inserted++; }   /* :88-97,121 */
            }
            free(b->pos);
        } else {
            nb[nn++] = *b;                                                    /* orphan :131-134,144-149 */
        }
    }
    count = (uint16_t)(count + inserted);                                     /* :140 */
    nb[0] = head;
    free(S->bins);
    S->bins = nb; S->nbins = nn; S->cap = S->nbins; S->count = count;
}

/* ---- libstdc++ std::sort (bits/stl_algo.h: __sort -> __introsort_loop + __final_insertion_sort),
 * restated for an array of u16 ids with comparator comp(a,b) = support[a] > support[b]
 * (common.h:112-117 SortBy).  Needed because choose() (common.h:162-170) keeps only ids[0] and the
 * sort is not stable once there are more than 16 bins (_S_threshold).                              */
typedef struct { const uint32_t* sup; } cmp_t;
static int cmp_gt(const cmp_t* c, uint16_t a, uint16_t b) { return c->sup[a] > c->sup[b]; }
static void ss_swap(uint16_t* a, uint16_t* b) { uint16_t t = *a; *a = *b; *b = t; }
static void ss_adjust_heap(uint16_t* first, long hole, long len, uint16_t value, const cmp_t* c) {
    const long top = hole; long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (cmp_gt(c, first[child], first[child - 1])) child--;
        first[hole] = first[child]; hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1]; hole = child - 1;
    }
    long parent = (hole - 1) / 2;                                              /* __push_heap */
    while (hole > top && cmp_gt(c, first[parent], value)) {
        first[hole] = first[parent]; hole = parent; parent = (hole - 1) / 2;
    }
    first[hole] = value;
}
static void ss_heap_sort(uint16_t* first, uint16_t* last, const cmp_t* c) {   /* __partial_sort(first,last,last) */
    long len = last - first;
    if (len >= 2) {                                                            /* __make_heap */
        long parent = (len - 2) / 2;
        for (;;) { uint16_t v = first[parent]; ss_adjust_heap(first, parent, len, v, c); if (parent == 0) break; parent--; }
    }
    /* __heap_select's loop over [middle,last) is empty; __sort_heap: */
    while (last - first > 1) {
        --last;
        uint16_t v = *last; *last = *first;
        ss_adjust_heap(first, 0, last - first, v, c);
    }
}
static void ss_move_median_to_first(uint16_t* result, uint16_t* a, uint16_t* b, uint16_t* cc, const cmp_t* c) {
    if (cmp_gt(c, *a, *b)) {
        if (cmp_gt(c, *b, *cc)) ss_swap(result, b);
        else if (cmp_gt(c, *a, *cc)) ss_swap(result, cc);
        else ss_swap(result, a);
    } else if (cmp_gt(c, *a, *cc)) ss_swap(result, a);
    else if (cmp_gt(c, *b, *cc)) ss_swap(result, cc);
    else ss_swap(result, b);
}
static uint16_t* ss_unguarded_partition(uint16_t* first, uint16_t* last, uint16_t* pivot, const cmp_t* c) {
    for (;;) {
        while (cmp_gt(c, *first, *pivot)) ++first;
        --last;
        while (cmp_gt(c, *pivot, *last)) --last;
        if (!(first < last)) return first;
        ss_swap(first, last);
        ++first;
    }
}
static void ss_introsort_loop(uint16_t* first, uint16_t* last, long depth, const cmp_t* c) {
    while (last - first > 16) {
        if (depth == 0) { ss_heap_sort(first, last, c); return; }
        --depth;
        uint16_t* mid = first + (last - first) / 2;
        ss_move_median_to_first(first, first + 1, mid, last - 1, c);
        uint16_t* cut = ss_unguarded_partition(first + 1, last, first, c);
        ss_introsort_loop(cut, last, depth, c);
        last = cut;
    }
}
static void ss_unguarded_linear_insert(uint16_t* last, const cmp_t* c) {
    uint16_t val = *last; uint16_t* next = last - 1;
    while (cmp_gt(c, val, *next)) { *last = *next; last = next; --next; }
    *last = val;
}
static void ss_insertion_sort(uint16_t* first, uint16_t* last, const cmp_t* c) {
    if (first == last) return;
    for (uint16_t* i = first + 1; i != last; ++i) {
        if (cmp_gt(c, *i, *first)) { uint16_t v = *i; memmove(first + 1, first, (size_t)(i - first) * sizeof(uint16_t)); *first = v; }
        else ss_unguarded_linear_insert(i, c);
    }
}
static void std_sort_ids(uint16_t* first, uint16_t* last, const cmp_t* c) {
    if (first == last) return;
    long n = last - first, lg = 0;
    while ((1L << (lg + 1)) <= n) lg++;                                        /* std::__lg */
    ss_introsort_loop(first, last, lg * 2, c);
    if (last - first > 16) {
        ss_insertion_sort(first, first + 16, c);
        for (uint16_t* i = first + 16; i != last; ++i) ss_unguarded_linear_insert(i, c);
    } else ss_insertion_sort(first, last, c);
}
/* common.h:142-150 chain() / :162-170 choose(): index of the winning bin.  ids are u16 (iota wraps for
 * more than 65536 bins, which cannot happen: bins <= products of a pair <= 65535 positions). */
uint32_t oracle_choose_bin(const uint32_t* support, uint32_t nbins) {
    if (nbins <= 1) return 0;
    uint16_t* ids = (uint16_t*)malloc(sizeof(uint16_t) * nbins);
    for (uint32_t i = 0; i < nbins; ++i) ids[i] = (uint16_t)i;
    cmp_t c; c.sup = support;
    std_sort_ids(ids, ids + nbins, &c);
    uint32_t w = ids[0];
    free(ids);
    return w;
}

/* ---------------------------------------------------------------------------------------------------
 * HashSpGEMM (include/overlap.hpp:650-789): estimateFLOP :157, estimateNNZ_Hash :205, LocalSpGEMM :281
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t rid;       /* row of C = the larger read id (H, "read1")           */
    uint32_t cid;       /* column of C = read i (V, "read2")                     */
    uint16_t count;     /* spmatType_::count                                     */
    uint16_t seedH;     /* choose().first                                        */
    uint16_t seedV;     /* choose().second                                       */
    uint16_t nbins;     /* number of bins at the end (saturated at 65535)        */
    uint16_t support;   /* chain(): support of the winning bin                   */
    uint16_t binov;     /* overlap estimate stored in the winning bin            */
    int32_t overlap;    /* overlapop(read[rid], read[cid], seedH, seedV) (:583)  */
} oracle_pair;

/* Phase 1: flops per column (estimateFLOP, lowtri) and nnz per column (estimateNNZ_Hash).
 * colflop/colnnz have nreads entries. */
void oracle_symbolic_range(uint32_t lo, uint32_t hi, const uint32_t* Bcolptr, const uint32_t* Browids, const uint32_t* Acolptr,
                           const uint32_t* Arowids, uint32_t* colflop, uint32_t* colnnz) {
    /* columns lo .. hi-1 (they are independent; full-size tests spread ranges over the host cores) */
    uint32_t cap = 16;
    uint32_t* tab = (uint32_t*)malloc(sizeof(uint32_t) * cap);
    for (uint32_t i = lo; i < hi; ++i) {
        uint32_t f = 0;
        for (uint32_t j = Bcolptr[i]; j < Bcolptr[i + 1]; ++j) {              /* overlap.hpp:177-198 */
            uint32_t c = Browids[j];
            for (uint32_t k = Acolptr[c]; k < Acolptr[c + 1]; ++k) if (i < Arowids[k]) ++f;
        }
        colflop[i] = f;
        uint32_t ht = pow2_at_least(16, f);                                   /* :230-234 */
        if (ht > cap) { cap = ht; tab = (uint32_t*)realloc(tab, sizeof(uint32_t) * cap); }
        for (uint32_t j = 0; j < ht; ++j) tab[j] = EMPTY32;
        uint32_t nz = 0;
        for (uint32_t j = Bcolptr[i]; j < Bcolptr[i + 1]; ++j) {              /* :242-272 */
            uint32_t c = Browids[j];
            for (uint32_t k = Acolptr[c]; k < Acolptr[c + 1]; ++k) {
                uint32_t key = Arowids[k];
                if (i >= key) continue;
                uint32_t h = (key * 107u) & (ht - 1);
                for (;;) {
                    if (tab[h] == key) break;
                    if (tab[h] == EMPTY32) { tab[h] = key; nz++; break; }
                    h = (h + 1) & (ht - 1);
                }
            }
        }
        colnnz[i] = nz;
    }
    free(tab);
}

void oracle_symbolic(uint32_t nreads, const uint32_t* Bcolptr, const uint32_t* Browids, const uint32_t* Acolptr,
                     const uint32_t* Arowids, uint32_t* colflop, uint32_t* colnnz) {
    oracle_symbolic_range(0, nreads, Bcolptr, Browids, Acolptr, Arowids, colflop, colnnz);
}

/* Phase 2: LocalSpGEMM (overlap.hpp:281-363) + what RunPairWiseAlignments reads from each value
 * (overlap.hpp:531-585: chain(), choose(), overlapop on the chosen seed).
 * colptrC = exclusive scan of colnnz (nreads+1).  out has colptrC[nreads] records, column-major, slot
 * order within a column -- exactly the order the reference writes lines in at one thread. */
/* the columns cols[0..ncols) only (cols == NULL: the columns 0..ncols-1); column cols[x] has colnnz[x] pairs and its records go
 * to out + outoff[x] */
void oracle_numeric_cols(const uint32_t* cols, uint32_t ncols, const uint32_t* colnnz, const uint64_t* outoff,
                         const uint32_t* Bcolptr, const uint32_t* Browids, const uint16_t* Bvalues,
                         const uint32_t* Acolptr, const uint32_t* Arowids, const uint16_t* Avalues,
                         const char* const* seqs, const uint32_t* lens, int k, int binSize, oracle_pair* out) {
    uint32_t cap = 16;
    uint32_t* hk = (uint32_t*)malloc(sizeof(uint32_t) * cap);
    oval_t* hv = (oval_t*)calloc(cap, sizeof(oval_t));
    for (uint32_t x = 0; x < ncols; ++x) {
        const uint32_t i = cols ? cols[x] : x;
        uint32_t nnzc = colnnz[x];
        uint32_t ht = pow2_at_least(16, nnzc);                                /* :291-295 */
        if (ht > cap) {
            hk = (uint32_t*)realloc(hk, sizeof(uint32_t) * ht);
            hv = (oval_t*)realloc(hv, sizeof(oval_t) * ht);
            memset(hv + cap, 0, sizeof(oval_t) * (ht - cap));
            cap = ht;
        }
        for (uint32_t j = 0; j < ht; ++j) hk[j] = EMPTY32;
        for (uint32_t j = Bcolptr[i]; j < Bcolptr[i + 1]; ++j) {              /* :306 */
            uint32_t c = Browids[j];
            uint16_t posV = Bvalues[j];
            for (uint32_t kk = Acolptr[c]; kk < Acolptr[c + 1]; ++kk) {       /* :310 */
                uint32_t key = Arowids[kk];
                if (i >= key) continue;                                       /* :315 */
                uint16_t posH = Avalues[kk];
                opos_t q; q.h = posH; q.v = posV;
                uint16_t ovq = (uint16_t)o_overlapop(seqs[key], lens[key], seqs[i], lens[i], posH, posV, (uint16_t)k);
                uint32_t h = (key * 107u) & (ht - 1);
                for (;;) {
                    if (hk[h] == key) { o_fold(&hv[h], 0, q, ovq, binSize, k); break; }       /* :324-328 */
                    if (hk[h] == EMPTY32) { hk[h] = key; o_fold(&hv[h], 1, q, ovq, binSize, k); break; } /* :329-334 */
                    h = (h + 1) & (ht - 1);
                }
            }
        }
        oracle_pair* o = out + outoff[x];
        uint32_t idx = 0;
        for (uint32_t j = 0; j < ht; ++j) {                                   /* :343-361 slot order */
            if (hk[j] == EMPTY32) continue;
            oval_t* v = &hv[j];
            uint32_t* sup = (uint32_t*)malloc(sizeof(uint32_t) * v->nbins);
            for (uint32_t b = 0; b < v->nbins; ++b) sup[b] = (uint16_t)v->bins[b].npos;   /* support is u16 */
            uint32_t w = oracle_choose_bin(sup, v->nbins);
            oracle_pair p;
            p.rid = hk[j]; p.cid = i; p.count = v->count;
            p.seedH = v->bins[w].pos[0].h; p.seedV = v->bins[w].pos[0].v;
            p.nbins = v->nbins > 65535 ? 65535 : (uint16_t)v->nbins;
            p.support = (uint16_t)sup[w];
            p.binov = v->bins[w].overlap;
            p.overlap = o_overlapop(seqs[p.rid], lens[p.rid], seqs[i], lens[i], p.seedH, p.seedV, (uint16_t)k);
            o[idx++] = p;
            free(sup);
            val_free(v);
        }
    }
    free(hk); free(hv);
}

void oracle_numeric(uint32_t nreads, const uint32_t* Bcolptr, const uint32_t* Browids, const uint16_t* Bvalues,
                    const uint32_t* Acolptr, const uint32_t* Arowids, const uint16_t* Avalues,
                    const char* const* seqs, const uint32_t* lens, int k, int binSize, const uint32_t* colptrC,
                    oracle_pair* out) {
    uint32_t* nz = (uint32_t*)malloc(sizeof(uint32_t) * (nreads ? nreads : 1));
    uint64_t* off = (uint64_t*)malloc(sizeof(uint64_t) * (nreads ? nreads : 1));
    for (uint32_t i = 0; i < nreads; ++i) { nz[i] = colptrC[i + 1] - colptrC[i]; off[i] = colptrC[i]; }
    oracle_numeric_cols(NULL, nreads, nz, off, Bcolptr, Browids, Bvalues, Acolptr, Arowids, Avalues, seqs, lens, k, binSize, out);
    free(nz); free(off);
}

/* ---------------------------------------------------------------------------------------------------
 * Xavier X-drop (xavier/xavier.h, xavier/simdutils.h) -- scalar restatement of the AVX2 int8 code
 * ------------------------------------------------------------------------------------------------- */
#define XW 32          /* VECTORWIDTH  simdutils.h:22 */
#define XLW 31         /* LOGICALWIDTH simdutils.h:23 */
#define XNINF (-128)   /* NINF         simdutils.h:48 */
#define XMIDDLE 15     /* MIDDLE       simdutils.h:51 */
#define XCUTOFF 102    /* CUTOFF       simdutils.h:53 */

static int8_t adds8(int a, int b) { int s = a + b; return (int8_t)(s > 127 ? 127 : (s < -128 ? -128 : s)); }
static int8_t subs8(int a, int b) { int s = a - b; return (int8_t)(s > 127 ? 127 : (s < -128 ? -128 : s)); }

typedef struct {
    const int8_t *qh, *qv;    /* padded: len+1 chars (incl. NUL) then 32 x NINF, simdutils.h:187-196 */
    uint32_t hl, vl;          /* hlength = len+1, simdutils.h:175-176 */
    int hoff, voff;
    int8_t vqh[XW], vqv[XW], a1[XW], a2[XW], a3[XW];
    int64_t best, curr, off, X;
    int endH, endV;           /* state.seed end positions */
    int xdrop;
    int flagged;              /* the uninitialised-maxpos first iteration was hit */
} xstate;

static void x_step(xstate* s) {                                              /* xavier.h:111-129 / :193-211 */
    int8_t a1f[XW], a2f[XW];
    for (int e = 0; e < XW; ++e) {
        int m = (s->vqh[e] == s->vqv[e]) ? 1 : -1;                           /* cmpeq + blendv */
        a1f[e] = adds8(m, s->a1[e]);
        int sh = (e == XW - 1) ? XNINF : s->a2[e + 1];                        /* shiftLeft simdutils.h:106-117 */
        int mx = sh > s->a2[e] ? sh : s->a2[e];
        a2f[e] = adds8(mx, -1);
    }
    for (int e = 0; e < XW; ++e) s->a3[e] = a1f[e] > a2f[e] ? a1f[e] : a2f[e];
    s->a3[XLW] = XNINF;                                                      /* :128 */
}
static void x_move_right(xstate* s) {                                        /* simdutils.h:263-274 */
    for (int e = 0; e < XW - 1; ++e) s->vqh[e] = s->vqh[e + 1];
    s->vqh[XW - 1] = XNINF;
    s->vqh[XLW - 1] = s->qh[s->hoff++];
    for (int e = 0; e < XW - 1; ++e) s->a1[e] = s->a2[e + 1];
    s->a1[XW - 1] = XNINF;
    memcpy(s->a2, s->a3, XW);
}
static void x_move_down(xstate* s) {                                         /* simdutils.h:276-289 */
    for (int e = XW - 1; e > 0; --e) s->vqv[e] = s->vqv[e - 1];
    s->vqv[0] = s->qv[s->voff++];
    memcpy(s->a1, s->a2, XW);
    for (int e = XW - 1; e > 0; --e) s->a2[e] = s->a3[e - 1];
    s->a2[0] = XNINF;
}
/* returns 1 if the X-drop fired (caller returns) */
static int x_after_step(xstate* s, int8_t* adb_out) {
    int8_t adb = s->a3[0];
    for (int e = 1; e < XW; ++e) if (s->a3[e] > adb) adb = s->a3[e];          /* :134 max over VECTORWIDTH */
    s->curr = adb + s->off;                                                  /* :135 */
    if (s->curr < s->best - s->X) { s->xdrop = 1; return 1; }                 /* :137-150 */
    if (adb > XCUTOFF) {                                                     /* :152-158 */
        int8_t mn = s->a3[0];
        for (int e = 1; e < XLW; ++e) if (s->a3[e] < mn) mn = s->a3[e];       /* min over LOGICALWIDTH */
        for (int e = 0; e < XW; ++e) { s->a2[e] = subs8(s->a2[e], mn); s->a3[e] = subs8(s->a3[e], mn); }
        s->off += mn;
    }
    if (s->curr > s->best) s->best = s->curr;                                 /* :161-162 */
    *adb_out = adb;
    return 0;
}

static void x_one_direction(xstate* s) {                                     /* xavier.h:257-274 */
    /* ---- Phase 1, xavier.h:20-103 */
    int DP[XLW + 2][XLW + 2];
    DP[0][0] = 0;
    for (int i = 1; i < XLW + 2; ++i) { DP[0][i] = -i; DP[i][0] = -i; }
    int DPmax = 0;
    for (int i = 1; i < XLW + 2; ++i)
        for (int j = 1; j <= XLW + 2 - i; ++j) {
            int oneF = DP[i - 1][j - 1] + ((s->qh[i - 1] == s->qv[j - 1]) ? 1 : -1);
            int twoF = (DP[i - 1][j] > DP[i][j - 1] ? DP[i - 1][j] : DP[i][j - 1]) - 1;
            DP[i][j] = oneF > twoF ? oneF : twoF;
            if (DP[i][j] > DPmax) DPmax = DP[i][j];
        }
    for (int i = 0; i < XLW; ++i) { s->vqh[i] = s->qh[i + 1]; s->vqv[i] = s->qv[XLW - i]; }
    s->vqh[XLW] = XNINF; s->vqv[XLW] = XNINF;
    int adm = -128;
    for (int i = 1; i < XLW + 1; ++i) {
        int v1 = DP[i][XLW - i + 1], v2 = DP[i + 1][XLW - i + 1];
        s->a1[i - 1] = (int8_t)v1; s->a2[i] = (int8_t)v2;
        if (v1 > adm) adm = v1;
    }
    s->a1[XLW] = XNINF; s->a2[0] = XNINF;
    for (int e = 0; e < XW; ++e) s->a3[e] = XNINF;
    s->best = DPmax; s->curr = adm;
    if (adm < DPmax - s->X) { s->xdrop = 1; s->endH = s->hoff; s->endV = s->voff; return; }   /* :91-99 */

    /* ---- Phase 2, xavier.h:105-183 */
    int maxpos = 0;          /* uninitialised in the reference (:165); policy: 0, and flag first-iteration use */
    int first = 1;
    while (s->hoff < (int)s->hl && s->voff < (int)s->vl) {
        int8_t adb;
        x_step(s);
        if (x_after_step(s, &adb)) { s->endH = s->hoff; s->endV = s->voff; return; }          /* :141-149 */
        int mx = 0, found = 0;
        for (int e = 0; e < XW; ++e) if (s->a3[e] > mx) { maxpos = e; mx = s->a3[e]; found = 1; }  /* :165-173 */
        if (first && !found) s->flagged = 1;
        first = 0;
        s->endH = s->hoff; s->endV = s->voff;                                 /* :175-176 */
        if (maxpos > XMIDDLE) x_move_right(s); else x_move_down(s);           /* :178-181 */
    }
    /* ---- Phase 4, xavier.h:185-251 */
    int dir = s->hoff >= (int)s->hl ? 1 : 0;                                  /* goDOWN=1, goRIGHT=0 */
    for (int i = 0; i < XLW - 3; ++i) {
        int8_t adb;
        x_step(s);
        if (x_after_step(s, &adb)) return;                                    /* :219-223 (end not updated) */
        int next = dir ^ 1;
        if (next == 0) x_move_right(s); else x_move_down(s);
        dir = next;
    }
}

static void x_run(const char* h, uint32_t hn, const char* v, uint32_t vn, int X, xstate* s, int8_t** bufs) {
    memset(s, 0, sizeof(*s));
    int8_t* qh = (int8_t*)malloc((size_t)hn + 1 + XW);
    int8_t* qv = (int8_t*)malloc((size_t)vn + 1 + XW);
    memcpy(qh, h, hn); qh[hn] = 0; memset(qh + hn + 1, XNINF, XW);            /* simdutils.h:191-196 */
    memcpy(qv, v, vn); qv[vn] = 0; memset(qv + vn + 1, XNINF, XW);
    s->qh = qh; s->qv = qv; s->hl = hn + 1; s->vl = vn + 1;
    s->hoff = XLW; s->voff = XLW; s->X = X;
    bufs[0] = qh; bufs[1] = qv;
}

typedef struct {
    int32_t score;
    int32_t begH, endH, begV, endV;
    int32_t strand;      /* 0 = "n", 1 = "c" */
    int32_t flagged;     /* an extension used the uninitialised maxpos (reference is ASLR-dependent there) */
    int32_t steps;       /* anti-diagonal steps taken, both directions (work unit for GCUPS) */
} oracle_aln;

static char comp(char c) { switch (c) { case 'A': return 'T'; case 'T': return 'A'; case 'G': return 'C'; case 'C': return 'G'; } return 'N'; }

/* xavier.h:276-374 XavierXDrop, XAVIER_EXTEND_BOTH branch (:325-373), on already-oriented strings. */
void oracle_xavier_xdrop(const char* target, uint32_t tlen, const char* query, uint32_t qlen, int begH, int begV,
                         int k, int X, oracle_aln* out) {
    int bH = begH, bV = begV, eH = begH + k, eV = begV + k;
    int64_t best1 = 0, best2 = 0;
    int flagged = 0, steps = 0;
    /* left: reversed prefixes INCLUDING the seed (:330-334) */
    uint32_t tpn = (uint32_t)eH, qpn = (uint32_t)eV;
    if (tpn < XW || qpn < XW) { bH = eH - (int)tpn; bV = eV - (int)qpn; }      /* :338-342 */
    else {
        char* tp = (char*)malloc(tpn); char* qp = (char*)malloc(qpn);
        for (uint32_t i = 0; i < tpn; ++i) tp[i] = target[tpn - 1 - i];
        for (uint32_t i = 0; i < qpn; ++i) qp[i] = query[qpn - 1 - i];
        xstate s; int8_t* bufs[2];
        x_run(tp, tpn, qp, qpn, X, &s, bufs);
        x_one_direction(&s);
        bH = eH - s.endH; bV = eV - s.endV;                                   /* :345-348 */
        best1 = s.best; flagged |= s.flagged;
        steps += (s.hoff - XLW) + (s.voff - XLW);
        free(bufs[0]); free(bufs[1]); free(tp); free(qp);
    }
    /* right: suffixes AFTER the seed (:351-352) */
    uint32_t tsn = tlen - (uint32_t)eH, qsn = qlen - (uint32_t)eV;
    if (tsn < XW || qsn < XW) { bH = eH + (int)tsn; bV = eV + (int)qsn; }      /* :356-360 writes BEGIN (sic) */
    else {
        xstate s; int8_t* bufs[2];
        x_run(target + eH, tsn, query + eV, qsn, X, &s, bufs);
        x_one_direction(&s);
        eH += s.endH; eV += s.endV;                                           /* :365-366 */
        best2 = s.best; flagged |= s.flagged;
        steps += (s.hoff - XLW) + (s.voff - XLW);
        free(bufs[0]); free(bufs[1]);
    }
    out->score = (int32_t)(best1 + best2);                                    /* simdutils.h:333-337 */
    out->begH = bH; out->endH = eH; out->begV = bV; out->endV = eV;
    out->flagged = flagged; out->steps = steps;
}

/* include/align.hpp:152-202 xavierAlign: row = H read (rid), col = V read (cid). */
void oracle_xavier_align(const char* row, uint32_t rowLen, const char* col, uint32_t colLen, int i, int j, int X,
                         int k, oracle_aln* out) {
    int rc = 1;                                                               /* :171-176 */
    for (int t = 0; t < k; ++t) if (comp(row[i + k - 1 - t]) != col[j + t]) { rc = 0; break; }
    if (rc) {
        char* cpy = (char*)malloc(rowLen);
        for (uint32_t t = 0; t < rowLen; ++t) cpy[t] = comp(row[rowLen - 1 - t]);     /* :178-179 */
        oracle_xavier_xdrop(cpy, rowLen, col, colLen, (int)rowLen - i - k, j, k, X, out);  /* :181-185 */
        out->strand = 1;
        free(cpy);
    } else {
        oracle_xavier_xdrop(row, rowLen, col, colLen, i, j, k, X, out);       /* :191 */
        out->strand = 0;
    }
}

/* include/overlap.hpp:413-497 PostAlignDecision (fixedThreshold == -1): returns pass/fail and the u16
 * overlap estimate `ov`.  ratiophi = slope(e) (align.hpp:72-80), delta = deltaChernoff. */
int oracle_post_align(int score, int begV, int endV, int begH, int endH, uint32_t len1_H, uint32_t len2_V,
                      double ratiophi, double delta, uint16_t* ov_out) {
    uint16_t read1len = (uint16_t)len1_H, read2len = (uint16_t)len2_V;        /* :441-442 */
    uint16_t overlapLenV = (uint16_t)(endV - begV), overlapLenH = (uint16_t)(endH - begH);   /* :444-445 */
    uint16_t minLeft = (uint16_t)(begV < begH ? begV : begH);                 /* :447 */
    int r2 = read2len - endV, r1 = read1len - endH;
    uint16_t minRight = (uint16_t)(r2 < r1 ? r2 : r1);                        /* :448 */
    uint16_t ov = (uint16_t)(minLeft + minRight + (overlapLenV + overlapLenH) / 2);   /* :449 */
    float thr = (float)((1 - delta) * (ratiophi * (float)ov));                /* :456 */
    *ov_out = ov;
    return (float)score >= thr;                                               /* :457 */
}

/* ---------------------------------------------------------------------------------------------------
 * LOGAN: the exact (growing band) gapped X-drop of the reference's CUDA build (loganGPU/functions.cuh), linear gap, +1/-1/-1.
 * Pinned on SeqAn's extendSeed(GappedXDrop) compiled from the reference (oracle/_ref: bella_ref_seqan_align), the CPU algorithm
 * the CUDA kernel ports line by line.  Where the two differ the CPU function decides (one place: an empty prefix / suffix, where
 * the CUDA kernel returns without writing its result, functions.cuh:270-271, and SeqAn returns 0).
 * ------------------------------------------------------------------------------------------------- */
#define LG_UNDEF (-32767)                                                     /* functions.cuh:18 */
#define LG_GAP (-1)                                                           /* functions.cuh:16 */

/* functions.cuh:223-408 extendSeedLGappedXDropOneDirectionGlobal: query read forward, database read BACKWARD
 * (dbPos = col + rows - antiDiagNo - 1, :119-120).  Returns the extension score; *extCol / *extRow = bases of query / database
 * covered (updateExtendedSeedL :67-100 adds / subtracts them). */
int oracle_logan_one_direction(const char* q, int qlen, const char* db, int dlen, int xdrop, int* extCol, int* extRow, int* steps) {
    const int cols = qlen + 1, rows = dlen + 1;                               /* :260-267 */
    *extCol = 0; *extRow = 0;
    if (rows == 1 || cols == 1) return 0;                                     /* :270 (SeqAn: return 0) */
    const int cap = (qlen < dlen ? qlen : dlen) + 4;
    int* b0 = (int*)malloc(sizeof(int) * cap);
    int* b1 = (int*)malloc(sizeof(int) * cap);
    int* b2 = (int*)malloc(sizeof(int) * cap);
    int *ad1 = b0, *ad2 = b1, *ad3 = b2;
    int a1size = 0, a2size = 0, a3size = 0;
    int minCol = 1, maxCol = 2;                                               /* :275-276 */
    int offset1 = 0, offset2 = 0, offset3 = 0;
    a2size = 1; ad2[0] = 0;                                                   /* initAntiDiags :186-216 */
    a3size = 2; ad3[0] = LG_GAP; ad3[1] = LG_GAP;
    int antiDiagNo = 1, best = 0;                                             /* :283-285 */
    while (minCol < maxCol) {                                                 /* :290 */
        ++antiDiagNo;
        int* t = ad1; ad1 = ad2; ad2 = ad3; ad3 = t;                          /* :299-311 */
        int tl = a1size; a1size = a2size; a2size = a3size; a3size = tl;
        offset1 = offset2; offset2 = offset3; offset3 = minCol - 1;
        a3size = maxCol + 1 - offset3;                                        /* initAntiDiag3 :160-184 */
        ad3[0] = LG_UNDEF;
        ad3[maxCol - offset3] = LG_UNDEF;
        if (antiDiagNo * LG_GAP > best - xdrop) {
            if (offset3 == 0) ad3[0] = antiDiagNo * LG_GAP;
            if (antiDiagNo - maxCol == 0) ad3[maxCol - offset3] = antiDiagNo * LG_GAP;
        }
        for (int col = minCol; col < maxCol; ++col) {                         /* computeAntidiag :102-148 */
            const int queryPos = col - 1, dbPos = col + rows - antiDiagNo - 1;
            int a = ad2[col - offset2], b = ad2[col - offset2 - 1];
            int tmp = (a > b ? a : b) + LG_GAP;
            const int sc = (q[queryPos] == db[dbPos]) ? 1 : -1;
            const int dg = ad1[col - offset1 - 1] + sc;
            if (dg > tmp) tmp = dg;
            ad3[col - minCol + 1] = (tmp < best - xdrop) ? LG_UNDEF : tmp;
        }
        int adb = LG_UNDEF;                                                   /* :318-333 */
        for (int x = 0; x < a3size; ++x) if (ad3[x] > adb) adb = ad3[x];
        if (adb > best) best = adb;
        while (minCol - offset3 < a3size && ad3[minCol - offset3] == LG_UNDEF &&          /* :339-343 */
               minCol - offset2 - 1 < a2size && ad2[minCol - offset2 - 1] == LG_UNDEF)
            ++minCol;
        while (maxCol - offset3 > 0 && ad3[maxCol - offset3 - 1] == LG_UNDEF && ad2[maxCol - offset2 - 1] == LG_UNDEF)   /* :346-350 */
            --maxCol;
        ++maxCol;
        if (minCol < antiDiagNo + 2 - rows) minCol = antiDiagNo + 2 - rows;   /* :358 end of the database segment */
        if (maxCol > cols) maxCol = cols;                                     /* :360 end of the query segment */
    }
    if (steps) *steps = antiDiagNo - 1;
    int lcol = a3size + offset3 - 2;                                          /* :364-366 */
    int lrow = antiDiagNo - lcol;
    int lscore = ad3[lcol - offset3];
    if (lscore == LG_UNDEF) {
        if (ad2[a2size - 2] != LG_UNDEF) {                                    /* :370-376 reached the end of the query segment */
            lcol = a2size + offset2 - 2; lrow = antiDiagNo - 1 - lcol; lscore = ad2[lcol - offset2];
        } else if (a2size > 2 && ad2[a2size - 3] != LG_UNDEF) {               /* :378-384 end of the database segment */
            lcol = a2size + offset2 - 3; lrow = antiDiagNo - 1 - lcol; lscore = ad2[lcol - offset2];
        }
    }
    if (lscore == LG_UNDEF)                                                   /* :389-401 general case */
        for (int x = 0; x < a1size; ++x)
            if (ad1[x] > lscore) { lscore = ad1[x]; lcol = x + offset1; lrow = antiDiagNo - 2 - lcol; }
    if (lscore != LG_UNDEF) { *extCol = lcol; *extRow = lrow; }               /* :403-404 */
    free(b0); free(b1); free(b2);
    return lscore;
}

/* The pair as RunPairWiseAlignmentsGPU prepares it (include/overlap.hpp:918-944: strand test, reverse complement of the H read,
 * seed remap) and extendSeedL runs it (functions.cuh:505-547 prefixes / suffixes, :680-682 score = left + right + k and the end
 * positions; the begin positions come back with the left kernel's seeds :631).  row = H read (rid), col = V read (cid). */
void oracle_logan_align(const char* row, uint32_t rowLen, const char* col, uint32_t colLen, int i, int j, int X, int k, oracle_aln* out) {
    int rc = 1;
    for (int t = 0; t < k; ++t) if (comp(row[i + k - 1 - t]) != col[j + t]) { rc = 0; break; }
    char* h = (char*)malloc(rowLen + 1);
    if (rc) { for (uint32_t t = 0; t < rowLen; ++t) h[t] = comp(row[rowLen - 1 - t]); i = (int)rowLen - i - k; }
    else memcpy(h, row, rowLen);
    const int begH = i, begV = j, endH = i + k, endV = j + k;
    /* left: query prefix reversed (reverse_copy :539), target prefix as is (memcpy :541; the kernel reads it backward) */
    char* pq = (char*)malloc(begV + 1);
    for (int t = 0; t < begV; ++t) pq[t] = col[begV - 1 - t];
    int lc = 0, lr = 0, st1 = 0, st2 = 0;
    const int left = oracle_logan_one_direction(pq, begV, h, begH, X, &lc, &lr, &st1);
    /* right: query suffix as is (memcpy :542), target suffix reversed (reverse_copy :543; read backward = forward) */
    const int sl = (int)rowLen - endH;
    char* st = (char*)malloc(sl + 1);
    for (int t = 0; t < sl; ++t) st[t] = h[rowLen - 1 - t];
    int rcol = 0, rrow = 0;
    const int right = oracle_logan_one_direction(col + endV, (int)colLen - endV, st, sl, X, &rcol, &rrow, &st2);
    out->score = left + right + k;                                            /* :680 */
    out->begH = begH - lr; out->begV = begV - lc;                             /* updateExtendedSeedL :81-83 */
    out->endH = endH + rrow; out->endV = endV + rcol;                         /* :96-97 */
    out->strand = rc;
    out->flagged = 0;
    out->steps = st1 + st2;
    free(h); free(pq); free(st);
}

/* include/overlap.hpp:797-871 PostAlignDecisionGPU (fixedThreshold == -1): as PostAlignDecision, but the threshold test is made
 * in double (:826-827), not in float */
int oracle_post_align_gpu(int score, int begV, int endV, int begH, int endH, uint32_t len1_H, uint32_t len2_V,
                          double ratiophi, double delta, uint16_t* ov_out) {
    uint16_t read1len = (uint16_t)len1_H, read2len = (uint16_t)len2_V;        /* :814-815 */
    uint16_t overlapLenV = (uint16_t)(endV - begV), overlapLenH = (uint16_t)(endH - begH);   /* :818-819 */
    uint16_t minLeft = (uint16_t)(begV < begH ? begV : begH);                 /* :821 */
    int r2 = read2len - endV, r1 = read1len - endH;
    uint16_t minRight = (uint16_t)(r2 < r1 ? r2 : r1);                        /* :822 */
    uint16_t ov = (uint16_t)(minLeft + minRight + (overlapLenV + overlapLenH) / 2);   /* :823 */
    double thr = (1 - delta) * (ratiophi * (double)ov);                       /* :830 */
    *ov_out = ov;
    return (double)score >= thr;                                              /* :831 */
}

/* overlap.hpp:149-154 toOriginalCoordinates (PAF, "-" strand) */
void oracle_to_original(int* begpH, int* endpH, int lenH) {
    unsigned int tmp = (unsigned int)*begpH;
    *begpH = lenH - *endpH;
    *endpH = lenH - (int)tmp;
}

double oracle_slope(double error) {                                          /* align.hpp:72-80 */
    double p_mat = (1 - error) * (1 - error);   /* pow(1-error,2) */
    double p_mis = 1 - p_mat;
    return 1.0 * p_mat - 1.0 * p_mis;
}
