// writer.hpp -- host side of the boundary: the output writer and the staged host <-> device copies.
//
// bella_hip_write_output replaces the formatting half of RunPairWiseAlignments and its writer (include/overlap.hpp:531-590 lines,
// :603-642 per-thread buffers -> one file through offset writes; PostAlignDecision's two formats :472-473 / :476-489): every host
// thread measures a contiguous share of the records, the sizes are prefix-summed, and every thread formats its share piecewise and
// writes the pieces at its offsets of the file.  No stringstream, no locale: decimal digits straight into place.
#pragma once
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bella_hip.h"

namespace bella {

inline unsigned dec_digits(uint64_t x) {
    if (x < 10) return 1;
    if (x < 100) return 2;
    if (x < 1000) return 3;
    if (x < 10000) return 4;
    if (x < 100000) return 5;
    unsigned n = 5;
    for (x /= 100000; x; x /= 10) ++n;
    return n;
}

// the two sinks a record is formatted into: one that only measures, one that writes at a pointer.  Same record code for both
// (format_record below), so the measured size is the written size by construction.
struct CountSink {
    uint64_t n = 0;
    inline void str(const char*, size_t len) { n += len; }
    inline void ch(char) { ++n; }
    inline void u(uint64_t x) { n += dec_digits(x); }                  // decimal, as operator<< prints an unsigned integer
    inline void i(int64_t x) { if (x < 0) { ++n; u((uint64_t)(-x)); } else u((uint64_t)x); }
};
struct PtrSink {
    char* p;
    inline void str(const char* s, size_t len) { std::memcpy(p, s, len); p += len; }
    inline void ch(char c) { *p++ = c; }
    inline void u(uint64_t x) {
        char* e = p + dec_digits(x);
        p = e;
        do { *--e = (char)('0' + x % 10); x /= 10; } while (x);
    }
    inline void i(int64_t x) { if (x < 0) { *p++ = '-'; u((uint64_t)(-x)); } else u((uint64_t)x); }
};

// chain.hpp:47-71 overlapop on the chosen seed, with the strand test delivered in bella_pair::flags bit0
inline int writer_seed_overlap(const bella_pair& p, int len1, int len2, unsigned short k) {
    unsigned short begpH = p.seedH, begpV = p.seedV;
    if (!(p.flags & 1)) begpH = (unsigned short)(len1 - begpH - k);
    unsigned short endpH = (unsigned short)(begpH + k), endpV = (unsigned short)(begpV + k);
    const int margin1 = std::min(begpH, begpV);
    const int margin2 = std::min(len1 - endpH, len2 - endpV);
    return margin1 + margin2 + k;
}

// one record -> its line (or nothing: an alignment that did not pass); returns whether a line was produced
template <typename Sink>
inline bool format_record(Sink& b, const bella_params& p, int paf, const char* const* names, const uint32_t* name_len, const uint32_t* lens,
                          const bella_pair& q, const bella_aln* a) {
    const uint32_t r1 = q.rid, r2 = q.cid;
    const unsigned short l1 = (unsigned short)lens[r1], l2 = (unsigned short)lens[r2];       // overlap.hpp:539-540 (unsigned short)
    if (p.skip_alignment) {                                                      // overlap.hpp:577-588
        b.str(names[r2], name_len[r2]); b.ch('\t'); b.str(names[r1], name_len[r1]); b.ch('\t');
        b.u(q.count); b.ch('\t'); b.i(writer_seed_overlap(q, (int)lens[r1], (int)lens[r2], p.kmer_size)); b.ch('\t');
        b.u(l2); b.ch('\t'); b.u(l1); b.ch('\n');
        return true;
    }
    if (!a->passed) return false;                                                // PostAlignDecision, overlap.hpp:413-497
    if (!paf) {
        b.str(names[r2], name_len[r2]); b.ch('\t'); b.str(names[r1], name_len[r1]); b.ch('\t');
        b.u(q.count); b.ch('\t'); b.i(a->score); b.ch('\t'); b.u(a->ov); b.ch('\t'); b.ch(a->strand ? 'c' : 'n'); b.ch('\t');
        b.i(a->begV); b.ch('\t'); b.i(a->endV); b.ch('\t'); b.u(l2); b.ch('\t'); b.i(a->begH); b.ch('\t'); b.i(a->endH); b.ch('\t'); b.u(l1); b.ch('\n');
    } else {
        int begH = a->begH, endH = a->endH;
        if (a->strand) { const unsigned int tmp = (unsigned int)begH; begH = l1 - endH; endH = (int)(l1 - tmp); }   // toOriginalCoordinates :149-154
        b.str(names[r2], name_len[r2]); b.ch('\t'); b.u(l2); b.ch('\t'); b.i(a->begV); b.ch('\t'); b.i(a->endV); b.ch('\t'); b.ch(a->strand ? '-' : '+'); b.ch('\t');
        b.str(names[r1], name_len[r1]); b.ch('\t'); b.u(l1); b.ch('\t'); b.i(begH); b.ch('\t'); b.i(endH); b.ch('\t'); b.i(a->score); b.ch('\t'); b.u(a->ov); b.ch('\t');
        b.u(255); b.ch('\n');
    }
    return true;
}

struct WriteShare {
    bella_write_stats st{};
    bool bad_id = false;
    bool io_fail = false;
    bool size_mismatch = false;
    std::vector<char> buf;                                                       // the piece being formatted
};

// pass 1 of a share: validate, measure, and the statistics of overlap.hpp:531-590
inline void measure_share(WriteShare& w, const bella_params& p, int paf, uint32_t nreads, const char* const* names, const uint32_t* name_len,
                          const uint32_t* lens, const bella_pair* pairs, const bella_aln* alns, uint64_t lo, uint64_t hi) {
    CountSink c;
    for (uint64_t n = lo; n < hi; ++n) {
        const bella_pair& q = pairs[n];
        if (q.rid >= nreads || q.cid >= nreads) { w.bad_id = true; return; }
        if (!p.skip_alignment) {
            const bella_aln& a = alns[n];
            ++w.st.aligned_pairs;
            w.st.total_read_len += (uint64_t)(unsigned short)lens[q.rid] + (unsigned short)lens[q.cid];
            w.st.aligned_bases += (uint64_t)(int64_t)(a.endV - a.begV);
            if (!a.passed) w.st.bases_failed += (uint64_t)(int64_t)(a.endV - a.begV);
            else w.st.bases_passed += (uint64_t)(int64_t)(a.endV - a.begV);
        }
        if (format_record(c, p, paf, names, name_len, lens, q, alns ? alns + n : nullptr)) ++w.st.lines;
    }
    w.st.bytes = c.n;
}

// Two passes over the records on T host threads: measure every share (exact bytes, no formatting), prefix-sum the sizes, grow the
// file, then every thread formats its share about a megabyte at a time into a buffer that stays in its cache and writes each piece
// at its running offset: formatting overlaps the writes, and no share is ever held in memory whole.  Buffered writes to one file
// take the file's inode lock in the kernel, so the call is bound by ONE stream into the page cache (10-12 GB/s on the MI355X host);
// 16 threads saturate it and more only queue on the lock (profiles/r03_writer_sweep.txt: 3.3 GB in 267-323 ms at 16 threads,
// 690-1180 ms at 64-256; formatting into a shared mapping of the file instead collapses on page-fault contention: 1.0 s at 8
// threads, 7.0 s at 256).  So nthreads = 0 means min(hardware threads, 16).
inline int write_output_impl(const char* path, const bella_params* p, int paf, uint32_t nreads, const char* const* names, const uint32_t* lens,
                             const bella_pair* pairs, const bella_aln* alns, uint64_t npairs, int nthreads, bella_write_stats* out, std::string& err) {
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    if (!path || !p || (nreads && (!names || !lens)) || (npairs && !pairs) || (npairs && !p->skip_alignment && !alns)) { err = "null argument"; return BELLA_ERR_BAD_ARG; }
    int T = nthreads > 0 ? nthreads : std::min(16, (int)std::thread::hardware_concurrency());
    if (T < 1) T = 1;
    if (T > 256) T = 256;
    if ((uint64_t)T > npairs / 32768 + 1) T = (int)(npairs / 32768 + 1);       // (a thread per ~2 MB of text: below that its start-up costs more)
    std::vector<uint32_t> name_len(nreads);
    for (uint32_t r = 0; r < nreads; ++r) name_len[r] = (uint32_t)std::strlen(names[r]);
    std::vector<WriteShare> W((size_t)T);
    auto on_threads = [&](auto&& f) {
        std::vector<std::thread> th;
        for (int t = 1; t < T; ++t) th.emplace_back(f, t);
        f(0);
        for (auto& x : th) x.join();
    };
    auto lo_of = [&](int t) { return npairs * (uint64_t)t / (uint64_t)T; };
    on_threads([&](int t) { measure_share(W[(size_t)t], *p, paf, nreads, names, name_len.data(), lens, pairs, alns, lo_of(t), lo_of(t + 1)); });
    for (auto& w : W)
        if (w.bad_id) { err = "pair record with a read id out of range"; return BELLA_ERR_BAD_ARG; }
    const auto t1 = clk::now();
    // sizes -> offsets -> every thread puts its lines at its place (overlap.hpp:603-642), after what the file already holds
    // (the file is opened in append mode there, :613: a later stage goes behind an earlier one)
    const int fd = ::open(path, O_WRONLY | O_CREAT, 0644);
    if (fd < 0) { err = std::string("cannot open ") + path; return BELLA_ERR_BAD_ARG; }
    struct stat sb;
    uint64_t at = 0;
    if (::fstat(fd, &sb) == 0) at = (uint64_t)sb.st_size;
    std::vector<uint64_t> off((size_t)T + 1, at);
    for (int t = 0; t < T; ++t) off[(size_t)t + 1] = off[(size_t)t] + W[(size_t)t].st.bytes;
    const uint64_t end = off[(size_t)T];
    bool ok = end == at || ::ftruncate(fd, (off_t)end) == 0;
    if (ok && end > at) {
        const size_t kPiece = (size_t)1 << 20;
        size_t slack = 256;
        for (uint32_t r = 0; r < nreads; ++r) slack = std::max<size_t>(slack, 2 * (size_t)name_len[r] + 256);           // (one line at most)
        {
            // every thread: format about a megabyte into a buffer that stays in its cache, write it at its running offset, again
            on_threads([&](int t) {
                WriteShare& w = W[(size_t)t];
                w.buf.resize(kPiece + slack);
                uint64_t o = off[(size_t)t];
                PtrSink s{w.buf.data()};
                auto flush = [&]() {
                    const char* q = w.buf.data();
                    uint64_t left = (uint64_t)(s.p - q);
                    while (left) {
                        const ssize_t n = ::pwrite(fd, q, left, (off_t)o);
                        if (n <= 0) { w.io_fail = true; return false; }
                        q += n; o += (uint64_t)n; left -= (uint64_t)n;
                    }
                    s.p = w.buf.data();
                    return true;
                };
                for (uint64_t n = lo_of(t), hi = lo_of(t + 1); n < hi; ++n) {
                    format_record(s, *p, paf, names, name_len.data(), lens, pairs[n], alns ? alns + n : nullptr);
                    if ((size_t)(s.p - w.buf.data()) >= kPiece && !flush()) return;
                }
                if (!flush()) return;
                if (o != off[(size_t)t + 1]) w.size_mismatch = true;
            });
        }
    }
    ::close(fd);
    for (auto& w : W) {
        if (w.size_mismatch) { err = "writer: a share's formatted size differs from its measured size"; return BELLA_ERR_STATE; }
        if (w.io_fail) ok = false;
    }
    if (!ok) { err = std::string("short write to ") + path; return BELLA_ERR_BAD_ARG; }
    const auto t2 = clk::now();
    if (out) {
        bella_write_stats s{};
        for (auto& w : W) {
            s.lines += w.st.lines; s.bytes += w.st.bytes; s.aligned_pairs += w.st.aligned_pairs; s.aligned_bases += w.st.aligned_bases;
            s.total_read_len += w.st.total_read_len; s.bases_passed += w.st.bases_passed; s.bases_failed += w.st.bases_failed;
        }
        s.format_seconds = std::chrono::duration<double>(t1 - t0).count();
        s.seconds = std::chrono::duration<double>(t2 - t0).count();
        s.threads = (uint32_t)T;
        *out = s;
    }
    return 0;
}

// ---- staged copies: pageable host memory <-> device through two pinned buffers, the host side of a chunk copied by several
// threads while the previous chunk is on the wire (a plain hipMemcpy from pageable memory runs at ~1 GB/s on these boxes) ----
struct Stager {
    static constexpr size_t kChunk = (size_t)64 << 20;
    void* pin[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    bool ok = false;
    bool init() {
        if (ok) return true;
        for (int i = 0; i < 2; ++i) {
            if (hipHostMalloc(&pin[i], kChunk, hipHostMallocDefault) != hipSuccess) return false;
            if (hipEventCreateWithFlags(&done[i], hipEventDisableTiming) != hipSuccess) return false;
        }
        ok = true;
        return true;
    }
    void destroy() {
        for (int i = 0; i < 2; ++i) {
            if (pin[i]) (void)hipHostFree(pin[i]);
            if (done[i]) (void)hipEventDestroy(done[i]);
            pin[i] = nullptr; done[i] = nullptr;
        }
        ok = false;
    }
    static void par_memcpy(void* dst, const void* src, size_t n) {
        const size_t kSlice = (size_t)8 << 20;
        if (n <= kSlice) { std::memcpy(dst, src, n); return; }
        const size_t parts = (n + kSlice - 1) / kSlice;
        std::vector<std::thread> th;
        for (size_t q = 1; q < parts; ++q)
            th.emplace_back([=]() { const size_t o = q * kSlice; std::memcpy((char*)dst + o, (const char*)src + o, std::min(kSlice, n - o)); });
        std::memcpy(dst, src, kSlice);
        for (auto& x : th) x.join();
    }
    hipError_t h2d(void* dst, const void* src, size_t bytes, hipStream_t st) {
        if (bytes < ((size_t)1 << 20) || !init()) return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
        return h2d_fill(dst, bytes, st, [src](void* pinned, size_t o, size_t n) { par_memcpy(pinned, (const char*)src + o, n); });
    }
    // the same with the source produced piecewise: fill(pinned, o, n) writes bytes [o, o + n) of the payload into the pinned
    // buffer while the previous chunk is on the wire (FASTQ ingest gathers the bases out of the mapped file this way)
    template <typename Fill>
    hipError_t h2d_fill(void* dst, size_t bytes, hipStream_t st, Fill&& fill) {
        if (bytes < ((size_t)1 << 20) || !init()) {               // small, or no pinned memory to be had: pageable chunks, synchronous copies
            std::vector<char> tmp(std::min(kChunk, bytes));
            for (size_t o = 0; o < bytes; o += kChunk) {
                const size_t n = std::min(kChunk, bytes - o);
                fill(tmp.data(), o, n);
                hipError_t e = hipMemcpyAsync((char*)dst + o, tmp.data(), n, hipMemcpyHostToDevice, st);
                if (e == hipSuccess) e = hipStreamSynchronize(st);
                if (e != hipSuccess) return e;
            }
            return hipSuccess;
        }
        int b = 0;
        for (size_t o = 0; o < bytes; o += kChunk, b ^= 1) {
            const size_t n = std::min(kChunk, bytes - o);
            hipError_t e = hipEventSynchronize(done[b]);          // the copy that last used this buffer has left it
            if (e != hipSuccess) return e;
            fill(pin[b], o, n);
            e = hipMemcpyAsync((char*)dst + o, pin[b], n, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipEventRecord(done[b], st);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    // returns with the data in dst (synchronises the stream)
    hipError_t d2h(void* dst, const void* src, size_t bytes, hipStream_t st) {
        if (bytes < ((size_t)1 << 20) || !init()) {
            hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st);
            return e == hipSuccess ? hipStreamSynchronize(st) : e;
        }
        const size_t nchunks = (bytes + kChunk - 1) / kChunk;
        for (size_t q = 0; q <= nchunks; ++q) {
            if (q < nchunks) {
                const size_t o = q * kChunk, n = std::min(kChunk, bytes - o);
                hipError_t e = hipMemcpyAsync(pin[q & 1], (const char*)src + o, n, hipMemcpyDeviceToHost, st);
                if (e == hipSuccess) e = hipEventRecord(done[q & 1], st);
                if (e != hipSuccess) return e;
            }
            if (q > 0) {                                            // chunk q-1 has landed (or lands now) while chunk q is on the wire
                const size_t o = (q - 1) * kChunk, n = std::min(kChunk, bytes - o);
                hipError_t e = hipEventSynchronize(done[(q - 1) & 1]);
                if (e != hipSuccess) return e;
                par_memcpy((char*)dst + o, pin[(q - 1) & 1], n);
            }
        }
        return hipSuccess;
    }
};

}  // namespace bella
