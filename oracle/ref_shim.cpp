// TEST INFRASTRUCTURE ONLY (oracle).  Never linked, imported or executed by the product path.
//
// A thin C-ABI around the *reference's own* code (PASSIONLab/BELLA, CPU path), compiled by
// oracle/build_ref.sh from the sources where they lie under /root/reference into
// oracle/_ref/libbella_ref.so.  This file is ours; it only #includes the reference headers the way
// src/main.cpp:32-55 does and calls the reference's functions -- no reference source is copied here.
//
// Entry points (used by tests/, oracle/make_golden.py and bench.py's cpu_baseline leg):
//   bella_ref_build_B       CSC tuple ctor + MergeDuplicates, as main.cpp:476-480 -> CSC.cpp:422-479,301-420
//   bella_ref_hashspgemm    main.cpp:476-525: CSC(tuples) -> Transpose -> HashSpGEMM(...) (overlap.hpp:650)
//   bella_ref_xavier_align  align.hpp:152 xavierAlign
//   bella_ref_seqan_align   align.hpp:93 alignSeqAn (SeqAn extendSeed, GappedXDrop)
//   bella_ref_slope         align.hpp:72
#include <iostream>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <istream>
#include <vector>
#include <string>
#include <stdlib.h>
#include <algorithm>
#include <utility>
#include <array>
#include <typeinfo>
#include <tuple>
#include <queue>
#include <memory>
#include <stack>
#include <functional>
#include <cstring>
#include <string.h>
#include <math.h>
#include <cassert>
#include <ios>
#include <chrono>
#include <thread>
#include <sys/stat.h>
#include <sys/types.h>
#include <map>
#include <unordered_map>
#include <sstream>
#include <omp.h>

// same include set and order as the reference's src/main.cpp:32-55 (paths relative to -I$REF)
#include "include/cxxopts.hpp"
#include "libcuckoo/cuckoohash_map.hh"
#include "include/kmercount.hpp"
#include "include/chain.hpp"
#include "include/common/bellaio.h"
#include "include/minimizer.hpp"
#include "include/syncmer.hpp"
#include "kmercode/hash_funcs.h"
#include "kmercode/Kmer.hpp"
#include "kmercode/Buffer.h"
#include "kmercode/common.h"
#include "kmercode/fq_reader.h"
#include "kmercode/ParallelFASTQ.h"
#include "kmercode/bound.hpp"
#include "include/common/utility.h"
#include "include/common/CSC.h"
#include "include/common/CSR.h"
#include "include/common/common.h"
#include "include/common/IO.h"
#include "include/overlap.hpp"
#include "include/align.hpp"

typedef uint32_t KIDX;   // KMERINDEX, main.cpp:60

namespace {
struct StreamCapture {     // the reference talks on cout/cerr (stdout protocol, printLog); keep it
    std::streambuf *oc, *oe;
    std::stringstream sc, se;
    StreamCapture() { oc = std::cout.rdbuf(sc.rdbuf()); oe = std::cerr.rdbuf(se.rdbuf()); }
    ~StreamCapture() { std::cout.rdbuf(oc); std::cerr.rdbuf(oe); }
};
void copy_log(const std::string& s, char* log, size_t cap) {
    if (!log || !cap) return;
    size_t n = std::min(cap - 1, s.size());
    memcpy(log, s.data(), n);
    log[n] = 0;
}
std::vector<std::tuple<KIDX, KIDX, unsigned short>> make_tuples(uint64_t n, const uint32_t* tk, const uint32_t* tr,
                                                                const uint16_t* tp) {
    std::vector<std::tuple<KIDX, KIDX, unsigned short>> t(n);
    for (uint64_t i = 0; i < n; ++i) t[i] = std::make_tuple(tk[i], tr[i], tp[i]);
    return t;
}
}  // namespace

extern "C" {

// B = transpmat exactly as main.cpp:476-480 builds it.  colptr has nreads+1 entries; rowids/values must
// hold ntuples entries (upper bound); returns nnz after MergeDuplicates, or <0 on error.
int64_t bella_ref_build_B(uint32_t nreads, uint32_t nkmers, uint64_t ntuples, const uint32_t* t_kmer,
                          const uint32_t* t_read, const uint16_t* t_pos, uint32_t* colptr, uint32_t* rowids,
                          uint16_t* values) {
    StreamCapture cap;
    auto tuples = make_tuples(ntuples, t_kmer, t_read, t_pos);
    CSC<KIDX, unsigned short> B(tuples, nkmers, nreads,
                                [](unsigned short& p1, unsigned short& p2) { return p1; }, false);
    memcpy(colptr, B.colptr, sizeof(uint32_t) * (nreads + 1));
    memcpy(rowids, B.rowids, sizeof(uint32_t) * B.nnz);
    memcpy(values, B.values, sizeof(uint16_t) * B.nnz);
    return (int64_t)B.nnz;
}

// A = B.Transpose() (transpose.h:13); exposes the reference's A so tests can pin "ascending at 1 thread".
int64_t bella_ref_transpose_B(uint32_t nreads, uint32_t nkmers, uint64_t nnz, const uint32_t* Bcolptr,
                              const uint32_t* Browids, const uint16_t* Bvalues, uint32_t* Acolptr,
                              uint32_t* Arowids, uint16_t* Avalues) {
    StreamCapture cap;
    CSC<KIDX, unsigned short> B(nnz, nkmers, nreads);
    memcpy(B.colptr, Bcolptr, sizeof(uint32_t) * (nreads + 1));
    memcpy(B.rowids, Browids, sizeof(uint32_t) * nnz);
    memcpy(B.values, Bvalues, sizeof(uint16_t) * nnz);
    CSC<KIDX, unsigned short> A = B.Transpose();
    memcpy(Acolptr, A.colptr, sizeof(uint32_t) * (nkmers + 1));
    memcpy(Arowids, A.rowids, sizeof(uint32_t) * nnz);
    memcpy(Avalues, A.values, sizeof(uint16_t) * nnz);
    return (int64_t)nnz;
}

// The whole reference path from the tuple list on: main.cpp:476-525.  Writes `outfile` (removed first,
// main.cpp:130).  stdout_log / stderr_log receive what the reference printed (stdout protocol and the
// printLog lines incl. "OverlapTime", overlap.hpp:727 and "AlignmentTime", :759).
int bella_ref_hashspgemm(uint32_t nreads, uint32_t nkmers, uint64_t ntuples, const uint32_t* t_kmer,
                         const uint32_t* t_read, const uint16_t* t_pos, const char* const* seqs,
                         const char* const* names, int kmerSize, int binSize, int xDrop, int skipAlignment,
                         int outputPaf, double errorRate, double deltaChernoff, double totalMemoryMB,
                         const char* outfile, char* stdout_log, size_t stdout_cap, char* stderr_log,
                         size_t stderr_cap) {
    StreamCapture cap;
    BELLApars bpars;
    bpars.kmerSize = kmerSize;
    bpars.binSize = binSize;
    bpars.xDrop = xDrop;
    bpars.skipAlignment = skipAlignment != 0;
    bpars.outputPaf = outputPaf != 0;
    bpars.errorRate = errorRate;
    bpars.deltaChernoff = deltaChernoff;
    bpars.totalMemory = totalMemoryMB;
    bpars.userDefMem = true;
    double ratiophi = slope(bpars.errorRate);  // main.cpp:323

    readVector_ reads(nreads);
    for (uint32_t r = 0; r < nreads; ++r) {
        reads[r].nametag = names[r];
        reads[r].seq = seqs[r];
        reads[r].readid = r;
    }
    auto transtuples = make_tuples(ntuples, t_kmer, t_read, t_pos);
    std::cout << nkmers << std::endl;  // main.cpp:473
    CSC<KIDX, unsigned short> transpmat(transtuples, nkmers, nreads,
                                        [](unsigned short& p1, unsigned short& p2) { return p1; }, false);
    std::vector<std::tuple<KIDX, KIDX, unsigned short>>().swap(transtuples);
    CSC<KIDX, unsigned short> spmat = transpmat.Transpose();

    std::string of(outfile);
    remove(of.c_str());
    std::vector<char> ofc(of.begin(), of.end());
    ofc.push_back(0);

    spmatPtr_ getvaluetype(std::make_shared<spmatType_>());
    HashSpGEMM(
        spmat, transpmat,
        [&bpars, &reads](const unsigned short int& begpH, const unsigned short int& begpV, const unsigned int& id1,
                         const unsigned int& id2) {
            spmatPtr_ value(std::make_shared<spmatType_>());
            std::string& read1 = reads[id1].seq;
            std::string& read2 = reads[id2].seq;
            multiop(value, read1, read2, begpH, begpV, bpars.kmerSize);
            return value;
        },
        [&bpars, &reads](spmatPtr_& m1, spmatPtr_& m2, const unsigned int& id1, const unsigned int& id2) {
            std::string& readname1 = reads[id1].nametag;
            std::string& readname2 = reads[id2].nametag;
            chainop(m1, m2, bpars, readname1, readname2);
            return m1;
        },
        reads, getvaluetype, ofc.data(), bpars, ratiophi);
    copy_log(cap.sc.str(), stdout_log, stdout_cap);
    copy_log(cap.se.str(), stderr_log, stderr_cap);
    return 0;
}

// align.hpp:152.  out[0..4] = score, beginH, endH, beginV, endV ; strand = 'n' / 'c'.
int bella_ref_xavier_align(const char* row, const char* col, int rowLen, int i, int j, int xDrop, int kmerSize,
                           int* out, char* strand) {
    std::string r(row), c(col);
    xavierResult res = xavierAlign(r, c, rowLen, i, j, xDrop, kmerSize);
    out[0] = res.score;
    out[1] = getBeginPositionH(res.seed);
    out[2] = getEndPositionH(res.seed);
    out[3] = getBeginPositionV(res.seed);
    out[4] = getEndPositionV(res.seed);
    *strand = res.strand[0];
    return 0;
}

// xavier.h:276 on raw (already oriented) strings: lets tests hit XavierXDrop edge cases directly.
int bella_ref_xavier_xdrop(const char* target, const char* query, int begH, int begV, int kmerSize, int xDrop,
                           int* out) {
    std::string t(target), q(query);
    ScoringSchemeX sc(1, -1, -1);
    SeedX seed(begH, begV, kmerSize);
    std::pair<int, int> r = XavierXDrop(seed, XAVIER_EXTEND_BOTH, t, q, sc, xDrop);
    out[0] = r.first;
    out[1] = getBeginPositionH(seed);
    out[2] = getEndPositionH(seed);
    out[3] = getBeginPositionV(seed);
    out[4] = getEndPositionV(seed);
    out[5] = r.second;
    return 0;
}

// include/align.hpp:93 alignSeqAn = SeqAn's gapped X-drop extendSeed (seqan/seeds/seeds_extension.h:789-848), the CPU algorithm that
// loganGPU/functions.cuh:223-408 ports to CUDA: the pin of the oracle's LOGAN restatement.  out = {score, begH, endH, begV, endV}
int bella_ref_seqan_align(const char* row, const char* col, int rowLen, int i, int j, int xDrop, int kmerSize, int* out, char* strand) {
    std::string r(row), c(col);
    seqAnResult res = alignSeqAn(r, c, rowLen, i, j, xDrop, kmerSize, false, false, false);
    out[0] = res.score;
    out[1] = (int)beginPositionH(res.seed);
    out[2] = (int)endPositionH(res.seed);
    out[3] = (int)beginPositionV(res.seed);
    out[4] = (int)endPositionV(res.seed);
    *strand = res.strand[0];
    return 0;
}

double bella_ref_slope(double e) { return slope(e); }

}  // extern "C"
