// TEST INFRASTRUCTURE ONLY.  Proves the drop-in claim of INTEGRATION.md: this translation unit includes the REFERENCE's
// headers exactly as src/main.cpp:32-55 does, then bella_amd/host/bella_hip_shim.hpp, and makes the reference's own call
// (main.cpp:476-525, same lambdas).  Overload resolution must pick the shim's HashSpGEMM, i.e. the work runs in
// libbella_hip.so on the GPU.  Built by oracle/build_ref.sh into oracle/_ref/libbella_dropin.so (needs /root/reference).
#include <iostream>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <istream>
#include <vector>
#include <string>
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
// ---- the one line a BELLA maintainer adds to src/main.cpp (after its line 55) ----
#include "bella_hip_shim.hpp"

typedef uint32_t KIDX;

// numGPU = BELLApars::numGPU (-g), totalMemoryMB = BELLApars::totalMemory (-m): the shim derives contexts and stages from them
extern "C" int bella_dropin_hashspgemm2(uint32_t nreads, uint32_t nkmers, uint64_t ntuples, const uint32_t* t_kmer,
                                        const uint32_t* t_read, const uint16_t* t_pos, const char* const* seqs,
                                        const char* const* names, int kmerSize, int binSize, int xDrop, int skipAlignment,
                                        int outputPaf, double errorRate, double deltaChernoff, int numGPU, double totalMemoryMB,
                                        const char* outfile, char* stdout_log, size_t stdout_cap) {
    std::stringstream sc;
    std::streambuf* oc = std::cout.rdbuf(sc.rdbuf());
    BELLApars bpars;
    bpars.kmerSize = kmerSize; bpars.binSize = binSize; bpars.xDrop = xDrop;
    bpars.skipAlignment = skipAlignment != 0; bpars.outputPaf = outputPaf != 0;
    bpars.errorRate = errorRate; bpars.deltaChernoff = deltaChernoff;
    bpars.totalMemory = totalMemoryMB; bpars.userDefMem = true; bpars.numGPU = (unsigned short)numGPU;
    double ratiophi = slope(bpars.errorRate);
    readVector_ reads(nreads);
    for (uint32_t r = 0; r < nreads; ++r) { reads[r].nametag = names[r]; reads[r].seq = seqs[r]; reads[r].readid = r; }
    std::vector<std::tuple<KIDX, KIDX, unsigned short>> transtuples(ntuples);
    for (uint64_t i = 0; i < ntuples; ++i) transtuples[i] = std::make_tuple(t_kmer[i], t_read[i], t_pos[i]);
    std::cout << nkmers << std::endl;
    CSC<KIDX, unsigned short> transpmat(transtuples, nkmers, nreads, [](unsigned short& p1, unsigned short& p2) { return p1; }, false);
    CSC<KIDX, unsigned short> spmat = transpmat.Transpose();
    std::string of(outfile);
    remove(of.c_str());
    std::vector<char> ofc(of.begin(), of.end());
    ofc.push_back(0);
    spmatPtr_ getvaluetype(std::make_shared<spmatType_>());
    HashSpGEMM(                                                   // verbatim call shape of src/main.cpp:498-525
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
    std::cout.rdbuf(oc);
    std::string s = sc.str();
    if (stdout_log && stdout_cap) { size_t n = std::min(stdout_cap - 1, s.size()); memcpy(stdout_log, s.data(), n); stdout_log[n] = 0; }
    return 0;
}

extern "C" int bella_dropin_hashspgemm(uint32_t nreads, uint32_t nkmers, uint64_t ntuples, const uint32_t* t_kmer,
                                       const uint32_t* t_read, const uint16_t* t_pos, const char* const* seqs,
                                       const char* const* names, int kmerSize, int binSize, int xDrop, int skipAlignment,
                                       int outputPaf, double errorRate, double deltaChernoff, const char* outfile,
                                       char* stdout_log, size_t stdout_cap) {
    return bella_dropin_hashspgemm2(nreads, nkmers, ntuples, t_kmer, t_read, t_pos, seqs, names, kmerSize, binSize, xDrop, skipAlignment,
                                    outputPaf, errorRate, deltaChernoff, 1, 400000.0, outfile, stdout_log, stdout_cap);
}

// what the shim's last HashSpGEMM did: {numeric columns, numeric passes, symbolic passes, nreads, stages, contexts, B' bytes max, B' bytes sum, host upload bytes}
extern "C" int bella_dropin_last_stats(uint64_t* out) {
    const bella_hip_detail::CallStats& s = bella_hip_detail::last_call_stats();
    out[0] = s.numeric_columns; out[1] = s.numeric_passes; out[2] = s.symbolic_passes; out[3] = s.nreads; out[4] = (uint64_t)s.stages;
    out[5] = (uint64_t)s.contexts; out[6] = s.layout_B_bytes_max; out[7] = s.layout_B_bytes_sum; out[8] = s.host_upload_bytes;
    return 0;
}

// the per-pair form on RESIDENT reads (bella_hip::use_reads once, then read ids): n calls on the same read set
extern "C" int bella_dropin_xavier_align_resident(uint32_t nreads, const char* const* seqs, int n, const uint32_t* rids, const uint32_t* cids, const int* is,
                                                  const int* js, int xDrop, int kmerSize, int* out, char* strands) {
    readVector_ reads(nreads);
    for (uint32_t r = 0; r < nreads; ++r) { reads[r].seq = seqs[r]; reads[r].readid = r; }
    bella_hip::use_reads(reads);
    for (int t = 0; t < n; ++t) {
        xavierResult res = bella_hip::xavierAlign(rids[t], cids[t], is[t], js[t], xDrop, kmerSize);
        out[5 * t] = res.score; out[5 * t + 1] = res.seed.beginPositionH; out[5 * t + 2] = res.seed.endPositionH;
        out[5 * t + 3] = res.seed.beginPositionV; out[5 * t + 4] = res.seed.endPositionV;
        strands[t] = res.strand[0];
    }
    return 0;
}

// the align.hpp call surface through the shim: out = {score, begH, endH, begV, endV}, strand = "n" / "c"
#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Wdeprecated-declarations"
extern "C" int bella_dropin_xavier_align(const char* row, const char* col, int i, int j, int xDrop, int kmerSize, int* out, char* strand) {
    const std::string r(row), c(col);
    xavierResult res = bella_hip::xavierAlign(r, c, (int)r.size(), i, j, xDrop, kmerSize);
    out[0] = res.score; out[1] = res.seed.beginPositionH; out[2] = res.seed.endPositionH; out[3] = res.seed.beginPositionV; out[4] = res.seed.endPositionV;
    strand[0] = res.strand[0]; strand[1] = 0;
    return 0;
}
#pragma GCC diagnostic pop

// the batched form (alignLogan's shape, align.hpp:210-211): n pairs at once
extern "C" int bella_dropin_align_batch(int n, const char* const* rows, const char* const* cols, const int* is, const int* js, int xDrop,
                                        int kmerSize, int* out, char* strands) {
    std::vector<std::string> target, query;
    std::vector<SeedX> seeds;
    for (int t = 0; t < n; ++t) { target.emplace_back(rows[t]); query.emplace_back(cols[t]); seeds.emplace_back(is[t], js[t], kmerSize); }
    BELLApars bp;
    bp.kmerSize = (unsigned short)kmerSize; bp.xDrop = (unsigned short)xDrop; bp.errorRate = 0.15;
    std::vector<xavierResult> res;
    bella_hip::alignXavier(target, query, seeds, bp, res);
    for (int t = 0; t < n; ++t) {
        out[5 * t] = res[t].score; out[5 * t + 1] = res[t].seed.beginPositionH; out[5 * t + 2] = res[t].seed.endPositionH;
        out[5 * t + 3] = res[t].seed.beginPositionV; out[5 * t + 4] = res[t].seed.endPositionV;
        strands[t] = res[t].strand[0];
    }
    return 0;
}
