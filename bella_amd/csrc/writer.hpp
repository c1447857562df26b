// writer.hpp -- host side of the boundary: the output writer and the staged host <-> device copies.
//
// bella_hip_write_output replaces the formatting half of RunPairWiseAlignments and its writer (include/overlap.hpp:531-590 lines,
// :603-642 per-thread buffers -> one file through offset writes; PostAlignDecision's two formats :472-473 / :476-489): every host
// thread formats a contiguous share of the records into its own buffer, the sizes are prefix-summed, and every thread writes its
// buffer at its offset of the file.  No stringstream, no locale: decimal digits straight into the buffer.
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

struct OutBuf {
    std::vector<char> v;
    size_t n = 0;
    inline void need(size_t more) { if (n + more > v.size()) v.resize(std::max(v.size() * 2, n + more + (1u << 16))); }
    inline void str(const char* s, size_t len) { std::memcpy(v.data() + n, s, len); n += len; }
    inline void ch(char c) { v[n++] = c; }
    inline void u(uint64_t x) {                       // decimal, as operator<< prints an unsigned integer
        char t[24];
        int k = 0;
        do { t[k++] = (char)('0' + x % 10); x /= 10; } while (x);
        while (k) v[n++] = t[--k];
    }
    inline void i(int64_t x) { if (x < 0) { v[n++] = '-'; u((uint64_t)(-x)); } else u((uint64_t)x); }
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

struct WriteShare {
    OutBuf buf;
    bella_write_stats st{};
};

inline void format_share(WriteShare& w, const bella_params& p, int paf, const char* const* names, const uint32_t* name_len, const uint32_t* lens,
                         const bella_pair* pairs, const bella_aln* alns, uint64_t lo, uint64_t hi) {
    OutBuf& b = w.buf;
    b.v.resize((size_t)(hi - lo) * 72 + (1u << 16));
    for (uint64_t n = lo; n < hi; ++n) {
        const bella_pair& q = pairs[n];
        const uint32_t r1 = q.rid, r2 = q.cid;
        const unsigned short l1 = (unsigned short)lens[r1], l2 = (unsigned short)lens[r2];   // overlap.hpp:539-540 (unsigned short)
        b.need((size_t)name_len[r1] + name_len[r2] + 160);
        if (p.skip_alignment) {                                                  // overlap.hpp:577-588
            b.str(names[r2], name_len[r2]); b.ch('\t'); b.str(names[r1], name_len[r1]); b.ch('\t');
            b.u(q.count); b.ch('\t'); b.i(writer_seed_overlap(q, (int)lens[r1], (int)lens[r2], p.kmer_size)); b.ch('\t');
            b.u(l2); b.ch('\t'); b.u(l1); b.ch('\n');
            ++w.st.lines;
            continue;
        }
        const bella_aln& a = alns[n];                                            // PostAlignDecision, overlap.hpp:413-497
        ++w.st.aligned_pairs;
        w.st.total_read_len += (uint64_t)l1 + l2;
        w.st.aligned_bases += (uint64_t)(int64_t)(a.endV - a.begV);
        if (!a.passed) { w.st.bases_failed += (uint64_t)(int64_t)(a.endV - a.begV); continue; }
        w.st.bases_passed += (uint64_t)(int64_t)(a.endV - a.begV);
        if (!paf) {
            b.str(names[r2], name_len[r2]); b.ch('\t'); b.str(names[r1], name_len[r1]); b.ch('\t');
            b.u(q.count); b.ch('\t'); b.i(a.score); b.ch('\t'); b.u(a.ov); b.ch('\t'); b.ch(a.strand ? 'c' : 'n'); b.ch('\t');
            b.i(a.begV); b.ch('\t'); b.i(a.endV); b.ch('\t'); b.u(l2); b.ch('\t'); b.i(a.begH); b.ch('\t'); b.i(a.endH); b.ch('\t'); b.u(l1); b.ch('\n');
        } else {
            int begH = a.begH, endH = a.endH;
            if (a.strand) { const unsigned int tmp = (unsigned int)begH; begH = l1 - endH; endH = (int)(l1 - tmp); }   // toOriginalCoordinates :149-154
            b.str(names[r2], name_len[r2]); b.ch('\t'); b.u(l2); b.ch('\t'); b.i(a.begV); b.ch('\t'); b.i(a.endV); b.ch('\t'); b.ch(a.strand ? '-' : '+'); b.ch('\t');
            b.str(names[r1], name_len[r1]); b.ch('\t'); b.u(l1); b.ch('\t'); b.i(begH); b.ch('\t'); b.i(endH); b.ch('\t'); b.i(a.score); b.ch('\t'); b.u(a.ov); b.ch('\t');
            b.u(255); b.ch('\n');
        }
        ++w.st.lines;
    }
    w.st.bytes = b.n;
}

inline int write_output_impl(const char* path, const bella_params* p, int paf, uint32_t nreads, const char* const* names, const uint32_t* lens,
                             const bella_pair* pairs, const bella_aln* alns, uint64_t npairs, int nthreads, bella_write_stats* out, std::string& err) {
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    if (!path || !p || (nreads && (!names || !lens)) || (npairs && !pairs) || (npairs && !p->skip_alignment && !alns)) { err = "null argument"; return BELLA_ERR_BAD_ARG; }
    for (uint64_t n = 0; n < npairs; ++n)
        if (pairs[n].rid >= nreads || pairs[n].cid >= nreads) { err = "pair record with a read id out of range"; return BELLA_ERR_BAD_ARG; }
    int T = nthreads > 0 ? nthreads : (int)std::thread::hardware_concurrency();
    if (T < 1) T = 1;
    if (T > 256) T = 256;
    if ((uint64_t)T > npairs / 32768 + 1) T = (int)(npairs / 32768 + 1);       // (a thread per ~2 MB of text: below that its start-up costs more)
    std::vector<uint32_t> name_len(nreads);
    for (uint32_t r = 0; r < nreads; ++r) name_len[r] = (uint32_t)std::strlen(names[r]);
    std::vector<WriteShare> W((size_t)T);
    auto share = [&](int t) {
        const uint64_t lo = npairs * (uint64_t)t / (uint64_t)T, hi = npairs * (uint64_t)(t + 1) / (uint64_t)T;
        format_share(W[(size_t)t], *p, paf, names, name_len.data(), lens, pairs, alns, lo, hi);
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < T; ++t) th.emplace_back(share, t);
        share(0);
        for (auto& x : th) x.join();
    }
    const auto t1 = clk::now();
    // sizes -> offsets -> every thread writes its buffer at its place (overlap.hpp:603-642), after what the file already holds
    // (the file is opened in append mode there, :613: a later stage goes behind an earlier one)
    const int fd = ::open(path, O_WRONLY | O_CREAT, 0644);
    if (fd < 0) { err = std::string("cannot open ") + path; return BELLA_ERR_BAD_ARG; }
    struct stat sb;
    uint64_t at = 0;
    if (::fstat(fd, &sb) == 0) at = (uint64_t)sb.st_size;
    std::vector<uint64_t> off((size_t)T + 1, at);
    for (int t = 0; t < T; ++t) off[(size_t)t + 1] = off[(size_t)t] + W[(size_t)t].buf.n;
    bool ok = ::ftruncate(fd, (off_t)off[(size_t)T]) == 0;
    auto put = [&](int t) {
        const char* q = W[(size_t)t].buf.v.data();
        uint64_t left = W[(size_t)t].buf.n, o = off[(size_t)t];
        while (left) {
            const ssize_t w = ::pwrite(fd, q, left > (1u << 30) ? (1u << 30) : left, (off_t)o);
            if (w <= 0) { ok = false; return; }
            q += w; o += (uint64_t)w; left -= (uint64_t)w;
        }
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < T; ++t) th.emplace_back(put, t);
        put(0);
        for (auto& x : th) x.join();
    }
    ::close(fd);
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
        int b = 0;
        for (size_t o = 0; o < bytes; o += kChunk, b ^= 1) {
            const size_t n = std::min(kChunk, bytes - o);
            hipError_t e = hipEventSynchronize(done[b]);          // the copy that last used this buffer has left it
            if (e != hipSuccess) return e;
            par_memcpy(pin[b], (const char*)src + o, n);
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
