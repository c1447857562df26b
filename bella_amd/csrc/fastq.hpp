// fastq.hpp -- FASTQ ingest on the host side of the library.  Plain C++, no device code: usable from the
// CPU test harness as well.
//
// Reference: ParallelFASTQ::fill_block -> get_next_fq_record (kmercode/fq_reader.c:540-610): records of exactly four lines
// (header, bases, '+' line, qualities), parsing stops at the first empty line or at a short record; the header is turned into
// the read name by get_fq_name (fq_reader.c:88-130: trailing blanks stripped, cut at the first TAB -- else the first blank --
// unless the name already ends in "/x"; "name 1:Y:..:.." / "name 2:..." Illumina comments become "name/1", "name/2") and
// src/main.cpp:357 drops the leading '@'.  The bases are taken verbatim (the device packer rejects anything but ACGT).
#pragma once
#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
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

// Whole-file parse.  Returns 0, or -1 with `err` set (cannot open / a record that does not start with '@').
inline int parse_fastq(const char* path, FastqData& out, std::string& err) {
    out.bases.clear(); out.offsets.assign(1, 0); out.names.clear();
    FILE* f = std::fopen(path, "rb");
    if (!f) { err = std::string("cannot open ") + path; return -1; }
    std::vector<char> buf;
    {
        char tmp[1 << 16];
        size_t n;
        while ((n = std::fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    }
    std::fclose(f);
    const char* p = buf.data();
    const char* const end = p + buf.size();
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

}  // namespace bella
