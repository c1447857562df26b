// bella_hip_shim.hpp -- the reference-side binding: BELLA's own call site, MI355X underneath.
//
// Include this header in the reference's src/main.cpp right AFTER its `#include "../include/align.hpp"` (main.cpp:55) and
// link libbella_hip.so.  It declares an overload of HashSpGEMM that is more specialised than the reference's template
// (include/overlap.hpp:650-652: IT=uint32_t [KMERINDEX, main.cpp:60], NT=unsigned short, FT=spmatPtr_), so the unchanged
// call at main.cpp:498-525 resolves to it.  Argument meaning, ownership and observable behaviour are the reference's:
//   A (spmat), B (transpmat): caller-owned CSC, read only.  Only B is read: A = B.Transpose() is rebuilt on the device with
//       ascending read ids per k-mer, i.e. the reference's own layout at OMP_NUM_THREADS=1 (transpose.h:26-50).
//   multop / addop: ignored -- the position-binning semiring (chain.hpp:74-150) runs inside the HIP kernels.
//   reads: caller-owned readVector_ (common.h:98-109), must be upper-case ACGT (align.hpp:40-55 asserts the same).
//   filename: the output file (already remove()d by main.cpp:130); BELLA or PAF lines, or 6-column --skip-alignment lines.
//   stdout: nnz(C) (overlap.hpp:686) and, when aligning, the number of lines written (overlap.hpp:771).
//   errors: the reference returns void and prints; so does this (message on stderr, then abort(), as CSC.cpp:269 does).
// The Kmer length is bpars.kmerSize (Kmer::set_k(bpars.kmerSize), main.cpp:183).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "bella_hip.h"

namespace bella_hip_detail {
inline void check(bella_ctx* c, int rc, const char* what) {
    if (rc == 0) return;
    std::cerr << "bella_hip: " << what << " failed: " << (c ? bella_hip_last_error(c) : bella_hip_strerror(rc)) << " (" << rc << ")"
              << std::endl;
    std::abort();
}
// chain.hpp:47-71 overlapop on the chosen seed, with the strand test delivered in bella_pair::flags bit0
inline int seed_overlap(const bella_pair& p, int len1, int len2, unsigned short k) {
    unsigned short begpH = p.seedH, begpV = p.seedV;
    if (!(p.flags & 1)) begpH = (unsigned short)(len1 - begpH - k);
    unsigned short endpH = begpH + k, endpV = begpV + k;
    int margin1 = std::min(begpH, begpV);
    int margin2 = std::min(len1 - endpH, len2 - endpV);
    return margin1 + margin2 + k;
}
}  // namespace bella_hip_detail

template <typename MultiplyOperation, typename AddOperation>
void HashSpGEMM(const CSC<uint32_t, unsigned short>& A, const CSC<uint32_t, unsigned short>& B, MultiplyOperation, AddOperation,
                const readVector_& reads, spmatPtr_& getvaluetype, char* filename, const BELLApars& bpars, const double& ratiophi) {
    using namespace bella_hip_detail;
    (void)A; (void)getvaluetype; (void)ratiophi;   // ratiophi = slope(bpars.errorRate) is recomputed from bpars
    bella_ctx* ctx = nullptr;
    check(nullptr, bella_hip_init(0, &ctx), "bella_hip_init");

    const uint32_t nreads = (uint32_t)reads.size();
    std::vector<uint64_t> offs(nreads + 1, 0);
    for (uint32_t r = 0; r < nreads; ++r) offs[r + 1] = offs[r] + reads[r].seq.size();
    std::string flat;
    flat.reserve(offs[nreads]);
    for (uint32_t r = 0; r < nreads; ++r) flat += reads[r].seq;
    check(ctx, bella_hip_set_reads(ctx, (const uint8_t*)flat.data(), offs.data(), nreads), "bella_hip_set_reads");
    std::string().swap(flat);
    check(ctx, bella_hip_set_B(ctx, bpars.kmerSize, (uint32_t)B.rows, B.colptr, B.rowids, B.values), "bella_hip_set_B");

    bella_params p;
    p.kmer_size = bpars.kmerSize;
    p.bin_size = bpars.binSize;
    p.xdrop = bpars.xDrop;
    p.skip_alignment = bpars.skipAlignment;
    p.error_rate = bpars.errorRate;
    p.delta_chernoff = bpars.deltaChernoff;
    uint64_t nnzc = 0, flops = 0;
    check(ctx, bella_hip_overlap(ctx, &p, &nnzc, &flops), "bella_hip_overlap");
    std::cout << nnzc << std::endl;                                           // overlap.hpp:686
    std::vector<bella_pair> pairs(nnzc);
    check(ctx, bella_hip_get_pairs(ctx, pairs.data(), nullptr, nullptr), "bella_hip_get_pairs");

    std::stringstream ss;
    size_t outputted = 0;
    if (bpars.skipAlignment) {                                                // overlap.hpp:577-588
        for (const bella_pair& q : pairs) {
            const readType_& r1 = reads[q.rid];
            const readType_& r2 = reads[q.cid];
            unsigned short l1 = r1.seq.length(), l2 = r2.seq.length();
            ss << r2.nametag << '\t' << r1.nametag << '\t' << q.count << '\t'
               << seed_overlap(q, (int)r1.seq.length(), (int)r2.seq.length(), bpars.kmerSize) << '\t' << l2 << '\t' << l1 << '\n';
            ++outputted;
        }
    } else {
        uint64_t npass = 0;
        check(ctx, bella_hip_align_pairs(ctx, &p, &npass), "bella_hip_align_pairs");
        std::vector<bella_aln> al(nnzc);
        if (nnzc) check(ctx, bella_hip_get_alignments(ctx, al.data()), "bella_hip_get_alignments");
        for (size_t n = 0; n < pairs.size(); ++n) {                            // PostAlignDecision, overlap.hpp:462-491
            const bella_aln& a = al[n];
            if (!a.passed) continue;
            const readType_& r1 = reads[pairs[n].rid];
            const readType_& r2 = reads[pairs[n].cid];
            unsigned short l1 = r1.seq.length(), l2 = r2.seq.length();
            if (!bpars.outputPaf) {
                ss << r2.nametag << '\t' << r1.nametag << '\t' << pairs[n].count << '\t' << a.score << '\t' << a.ov << '\t'
                   << (a.strand ? "c" : "n") << '\t' << a.begV << '\t' << a.endV << '\t' << l2 << '\t' << a.begH << '\t' << a.endH
                   << '\t' << l1 << '\n';
            } else {
                int begH = a.begH, endH = a.endH;
                if (a.strand) { unsigned int tmp = begH; begH = l1 - endH; endH = l1 - tmp; }   // toOriginalCoordinates :149-154
                ss << r2.nametag << '\t' << l2 << '\t' << a.begV << '\t' << a.endV << '\t' << (a.strand ? "-" : "+") << '\t'
                   << r1.nametag << '\t' << l1 << '\t' << begH << '\t' << endH << '\t' << a.score << '\t' << a.ov << '\t' << 255 << '\n';
            }
            ++outputted;
        }
        std::cout << outputted << std::endl;                                  // overlap.hpp:771
    }
    std::ofstream ofs(filename, std::ios::binary | std::ios::app);            // overlap.hpp:613
    const std::string text = ss.str();
    ofs.write(text.data(), (std::streamsize)text.size());
    ofs.close();
    bella_hip_destroy(ctx);
}
