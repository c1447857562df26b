// fastq.hpp -- FASTQ ingest on the host side of the library.  Plain C++, no device code: usable from the
// CPU test harness as well.
//
// Reference: ParallelFASTQ::fill_block -> get_next_fq_record (kmercode/fq_reader.c:540-610): records of exactly four lines
// (header, bases, '+' line, qualities), parsing stops at the first empty line or at a short record; the header is turned into
// the read name by get_fq_name (fq_reader.c:88-130: trailing blanks stripped, cut at the first TAB -- else the first blank --
// unless the name already ends in "/x"; "name 1:Y:..:.." / "name 2:..." Illumina comments become "name/1", "name/2") and
// src/main.cpp:357 drops the leading '@'.  The bases are taken verbatim (the device packer rejects anything but ACGT).
//
// Two implementations of the same rules:
//   parse_fastq_serial  one pass over the lines in file order, the direct restatement (kept as the cross-check of the other)
//   FastqIndex          what the library uses: the file is mapped, never copied; T threads count the newlines of their slice and
//                       find its first empty line, a prefix sum turns that into the line number at every slice start, and
//                       because a record is exactly four lines every slice then knows which of its lines are headers without
//                       any guessing about '@' (a quality string may start with one).  A second pass per slice collects
//                       (where the bases are, how many, the name); gather() copies any range of the concatenated base stream
//                       out of the mapping, so the caller can stream it through a pinned buffer to the device (bella_hip.hip)
//                       without ever holding the concatenation on the host.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace bella {

struct FastqData {
    std::vector<uint8_t> bases;        // all reads concatenated
    std::vector<uint64_t> offsets;     // nreads + 1
    std::vector<std::string> names;    // without the '@'
};

// get_fq_name (fq_reader.c:88-130) on a header line that still has its '@'; returns the name without the '@' (main.cpp:357)
inline std::string fastq_read_name(std::string h) {
    while (!h.empty() && std::isspace((unsigned char)h.back())) h.pop_back();       // strip_trailing_spaces
    const size_t len = h.size();
    if (len >= 2 && h[len - 2] != '/') {
        size_t end = h.find('\t');
        if (end == std::string::npos) end = h.find(' ');
        if (end != std::string::npos) {
            const bool slash_pair = end > 3 && h[end - 2] == '/' && (h[end - 1] == '1' || h[end - 1] == '2');
            const bool illumina = !(len < end + 7) && h[end + 2] == ':' && h[end + 4] == ':' && h[end + 6] == ':' &&
                                  (h[end + 1] == '1' || h[end + 1] == '2');
            if (slash_pair || !illumina) h.resize(end);                              // truncate the name at the comment
            else { h[end] = '/'; h.resize(end + 2); }                                // "@pair 1:Y:.." -> "@pair/1"
        }
    }
    return h.empty() ? h : h.substr(1);
}

// a file mapped read-only (or, where it cannot be mapped -- a pipe --, read into memory)
struct MappedFile {
    const char* p = nullptr;
    size_t n = 0;
    bool mapped = false;
    std::vector<char> held;
    MappedFile() = default;
    MappedFile(const MappedFile&) = delete;
    MappedFile& operator=(const MappedFile&) = delete;
    ~MappedFile() { close(); }
    void close() {
        if (mapped && p) ::munmap((void*)p, n);
        p = nullptr; n = 0; mapped = false;
        std::vector<char>().swap(held);
    }
    bool open(const char* path, std::string& err) {
        close();
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) { err = std::string("cannot open ") + path; return false; }
        struct stat st;
        if (::fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) {
            n = (size_t)st.st_size;
            if (n == 0) { ::close(fd); p = ""; return true; }
            void* m = ::mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) {
                (void)::madvise(m, n, MADV_WILLNEED);
                p = (const char*)m; mapped = true;
                ::close(fd);
                return true;
            }
        }
        char tmp[1 << 16];                                                           // not a regular file: read it all
        ssize_t got;
        while ((got = ::read(fd, tmp, sizeof(tmp))) > 0) held.insert(held.end(), tmp, tmp + got);
        ::close(fd);
        p = held.empty() ? "" : held.data();
        n = held.size();
        return true;
    }
};

// Whole-buffer serial parse.  Returns 0, or -1 with `err` set (a record that does not start with '@').
inline int parse_fastq_buffer_serial(const char* buf, size_t size, FastqData& out, std::string& err) {
    out.bases.clear(); out.offsets.assign(1, 0); out.names.clear();
    const char* p = buf;
    const char* const end = p + size;
    auto next_line = [&](const char*& b, const char*& e) -> bool {                   // [b, e) without the newline
        if (p >= end) return false;
        b = p;
        const char* nl = (const char*)std::memchr(p, '\n', (size_t)(end - p));
        e = nl ? nl : end;
        p = nl ? nl + 1 : end;
        return true;
    };
    uint64_t line_no = 0;
    for (;;) {
        const char *hb, *he, *sb, *se, *pb, *pe, *qb, *qe;
        if (!next_line(hb, he) || hb == he) break;                                  // end of file or an empty line: done
        ++line_no;
        if (*hb != '@') { err = "invalid FASTQ at line " + std::to_string(line_no) + ": expected a read name (@)"; return -1; }
        if (!next_line(sb, se) || sb == se) break;                                  // short record: the reference stops here
        if (!next_line(pb, pe) || pb == pe) break;
        if (!next_line(qb, qe) || qb == qe) break;
        line_no += 3;
        while (se > sb && (se[-1] == '\r' || se[-1] == '\n')) --se;                 // chompBuffer
        out.names.push_back(fastq_read_name(std::string(hb, he)));
        out.bases.insert(out.bases.end(), (const uint8_t*)sb, (const uint8_t*)se);
        out.offsets.push_back(out.bases.size());
    }
    return 0;
}

inline int parse_fastq_serial(const char* path, FastqData& out, std::string& err) {
    MappedFile f;
    if (!f.open(path, err)) return -1;
    return parse_fastq_buffer_serial(f.p, f.n, out, err);
}

// The multi-threaded index of a FASTQ file (see the head of this file).
struct FastqIndex {
    MappedFile file;
    std::vector<uint64_t> where;       // nreads: byte offset of the read's bases in the file
    std::vector<uint64_t> offsets;     // nreads + 1: offsets into the concatenated base stream
    std::vector<std::string> names;
    unsigned threads = 1;

    uint32_t nreads() const { return (uint32_t)where.size(); }
    uint64_t nbases() const { return offsets.empty() ? 0 : offsets.back(); }

    static unsigned pick_threads(size_t bytes, unsigned want) {
        unsigned hw = want ? want : std::thread::hardware_concurrency();
        if (hw == 0) hw = 1;
        const size_t by_size = bytes / ((size_t)4 << 20) + 1;                        // a thread is not worth less than 4 MB
        return (unsigned)std::min<size_t>(std::min<size_t>(hw, 64), by_size);
    }

    template <typename F>
    static void run(unsigned T, F&& f) {                                             // f(t) on T threads, the caller being thread 0
        std::vector<std::thread> th;
        for (unsigned t = 1; t < T; ++t) th.emplace_back([&f, t]() { f(t); });
        f(0);
        for (auto& x : th) x.join();
    }

    // Returns 0, or -1 with `err` set.  want_threads 0: one per core (at most 64, at least 4 MB of file each).
    int build(const char* path, std::string& err, unsigned want_threads = 0, size_t slice_bytes = 0) {
        where.clear(); offsets.assign(1, 0); names.clear();
        if (!file.open(path, err)) return -1;
        return build_from(file.p, file.n, err, want_threads, slice_bytes);
    }

    // same over a caller's buffer; the buffer has to outlive gather().  slice_bytes != 0 (tests): slices of exactly that size on
    // exactly want_threads threads, however small the input.
    int build_from(const char* buf, size_t size, std::string& err, unsigned want_threads = 0, size_t slice_bytes = 0) {
        where.clear(); offsets.assign(1, 0); names.clear();
        base_ = buf;
        const unsigned T = threads = slice_bytes ? std::max(1u, want_threads) : pick_threads(size, want_threads);
        if (size == 0) return 0;
        // slices; more of them than threads so that one slow slice (page faults) does not hold the others up
        const size_t nsl = slice_bytes ? (size + slice_bytes - 1) / slice_bytes
                                       : std::max<size_t>(1, std::min<size_t>((size_t)T * 8, size / ((size_t)1 << 20) + 1));
        const size_t sl = (size + nsl - 1) / nsl;
        auto lo = [&](size_t s) { return std::min(size, s * sl); };
        // pass 1: newlines of every slice up to its first empty line ('\n' at the file start or straight after another '\n')
        std::vector<uint64_t> nl(nsl + 1, 0), empty_at(nsl, UINT64_MAX);
        run(T, [&](unsigned t) {
            for (size_t s = t; s < nsl; s += T) {
                const char* p = buf + lo(s);
                const char* const e = buf + lo(s + 1);
                uint64_t cnt = 0;
                while (p < e) {
                    const char* q = (const char*)std::memchr(p, '\n', (size_t)(e - p));
                    if (!q) break;
                    if (q == buf || q[-1] == '\n') { empty_at[s] = (uint64_t)(q - buf); break; }
                    ++cnt;
                    p = q + 1;
                }
                nl[s + 1] = cnt;
            }
        });
        size_t stop = size;                                                          // the serial parser ends at the first empty line
        size_t live = nsl;
        for (size_t s = 0; s < nsl; ++s)
            if (empty_at[s] != UINT64_MAX) { stop = (size_t)empty_at[s]; live = s + 1; break; }
        for (size_t s = 0; s < live; ++s) nl[s + 1] += nl[s];                        // nl[s] = newlines before slice s
        uint64_t nlines = nl[live];
        if (stop == size && buf[size - 1] != '\n') ++nlines;                         // a last line without its newline still counts
        const uint64_t nrec = nlines / 4;                                            // short last record: dropped
        // pass 2: every slice takes the records whose header line begins in it
        struct Part { std::vector<uint64_t> where; std::vector<uint32_t> len; std::vector<std::string> names; uint64_t bad = UINT64_MAX; };
        std::vector<Part> part(live);
        run(T, [&](unsigned t) {
            for (size_t s = t; s < live; s += T) {
                Part& pt = part[s];
                const char* const send = buf + std::min(lo(s + 1), stop);
                const char* const fend = buf + stop;
                const char* p = buf + lo(s);
                uint64_t line = nl[s];                                              // number of the line that contains p
                if (p != buf && p[-1] != '\n') {                                    // p is inside a line of the previous slice
                    const char* q = (const char*)std::memchr(p, '\n', (size_t)(fend - p));
                    if (!q) continue;
                    p = q + 1; ++line;
                }
                auto eol = [&](const char* a) { const char* q = (const char*)std::memchr(a, '\n', (size_t)(fend - a)); return q ? q : fend; };
                while (p < send && (line & 3)) { p = std::min(eol(p) + 1, fend); ++line; }   // on to the first header line
                while (p < send && line < nlines) {
                    if (*p != '@') { pt.bad = line + 1; break; }                    // checked on the header of a short last record too
                    if (line / 4 >= nrec) break;
                    const char* he = eol(p);
                    const char* sb = he + 1;
                    const char* se = eol(sb);
                    const char* pe = eol(se + 1);
                    const char* qe = eol(pe + 1);
                    const char* sx = se;
                    while (sx > sb && (sx[-1] == '\r' || sx[-1] == '\n')) --sx;     // chompBuffer
                    pt.where.push_back((uint64_t)(sb - buf));
                    pt.len.push_back((uint32_t)std::min<uint64_t>((uint64_t)(sx - sb), UINT32_MAX));
                    pt.names.push_back(fastq_read_name(std::string(p, he)));
                    p = std::min(qe + 1, fend);
                    line += 4;
                }
            }
        });
        uint64_t bad = UINT64_MAX;
        for (const Part& pt : part) bad = std::min(bad, pt.bad);
        if (bad != UINT64_MAX) { err = "invalid FASTQ at line " + std::to_string(bad) + ": expected a read name (@)"; return -1; }
        where.reserve((size_t)nrec); offsets.reserve((size_t)nrec + 1); names.reserve((size_t)nrec);
        for (Part& pt : part) {
            for (size_t i = 0; i < pt.where.size(); ++i) {
                where.push_back(pt.where[i]);
                offsets.push_back(offsets.back() + pt.len[i]);
                names.push_back(std::move(pt.names[i]));
            }
        }
        return 0;
    }

    // bytes [off, off + n) of the concatenated base stream -> dst; safe to call from several threads on disjoint ranges
    void gather(uint8_t* dst, uint64_t off, uint64_t n) const {
        if (n == 0) return;
        size_t r = (size_t)(std::upper_bound(offsets.begin(), offsets.end(), off) - offsets.begin()) - 1;
        const uint64_t end = off + n;
        while (off < end) {
            const uint64_t in = off - offsets[r], take = std::min(end, offsets[r + 1]) - off;
            std::memcpy(dst, base_ + where[r] + in, (size_t)take);
            dst += take; off += take; ++r;
        }
    }

    // the same on `threads` threads (slices of 4 MB)
    void gather_parallel(uint8_t* dst, uint64_t off, uint64_t n) const {
        const uint64_t kSlice = (uint64_t)4 << 20;
        const uint64_t parts = (n + kSlice - 1) / kSlice;
        if (parts <= 1 || threads <= 1) { gather(dst, off, n); return; }
        const unsigned T = (unsigned)std::min<uint64_t>(threads, parts);
        run(T, [&](unsigned t) {
            for (uint64_t q = t; q < parts; q += T) gather(dst + q * kSlice, off + q * kSlice, std::min(kSlice, n - q * kSlice));
        });
    }

private:
    const char* base_ = nullptr;
};

// Whole-file parse through the index.  Returns 0, or -1 with `err` set (cannot open / a record that does not start with '@').
inline int parse_fastq(const char* path, FastqData& out, std::string& err, unsigned threads = 0) {
    out.bases.clear(); out.offsets.assign(1, 0); out.names.clear();
    FastqIndex ix;
    if (ix.build(path, err, threads)) return -1;
    out.bases.resize((size_t)ix.nbases());
    ix.gather_parallel(out.bases.data(), 0, ix.nbases());
    out.offsets = std::move(ix.offsets);
    out.names = std::move(ix.names);
    return 0;
}

}  // namespace bella
