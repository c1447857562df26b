// TEST INFRASTRUCTURE ONLY.  Compiles the per-lane functions the gfx950 kernels inline (bella_amd/csrc/core.hpp,
// xdrop.hpp) as plain host C++ so tests/test_core_host.py can check them against the oracle without a GPU.
// libbella_hip.so does not contain or call any of this; it is not a CPU fallback.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../bella_amd/csrc/core.hpp"
#include "../../bella_amd/csrc/fastq.hpp"
#include "../../bella_amd/csrc/xdrop.hpp"

using namespace bella;

extern "C" {

// out: count, nbins, support, binov, seedH, seedV, many_bins.  `shuffle_seed` != 0 permutes the list first
// (as the scatter phase may) and lets sort_products_by_index restore the order.
void h_fold_pair(const uint32_t* hv, const uint16_t* ov, uint32_t m, int k, int binSize, int use_scratch,
                 uint32_t shuffle_seed, uint32_t* out) {
    std::vector<uint32_t> P(hv, hv + m), Bm(m);
    for (uint32_t t = 0; t < m; ++t) Bm[t] = (t << 16) | ov[t];
    if (shuffle_seed) {
        uint64_t s = shuffle_seed;
        for (uint32_t t = m; t > 1; --t) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            uint32_t j = (uint32_t)((s >> 33) % t);
            std::swap(P[t - 1], P[j]);
            std::swap(Bm[t - 1], Bm[j]);
        }
    }
    std::vector<uint16_t> scr(m + 1);
    FoldResult fr;
    if (m == 1) { fr.count = 1; fr.nbins = 1; fr.support = 1; fr.binov = (uint16_t)(Bm[0] & 0xFFFF); fr.seed = P[0]; fr.many_bins = 0; }
    else {
        sort_products_by_index(P.data(), Bm.data(), m);
        fold_pair(P.data(), Bm.data(), m, k, binSize, use_scratch ? scr.data() : (uint16_t*)nullptr, fr);
    }
    out[0] = fr.count; out[1] = fr.nbins; out[2] = fr.support; out[3] = fr.binov; out[4] = fr.seed & 0xFFFF; out[5] = fr.seed >> 16;
    out[6] = fr.many_bins;
}

uint32_t h_choose(const uint32_t* sup, uint32_t n) {
    std::vector<uint16_t> ids(n);
    for (uint32_t i = 0; i < n; ++i) ids[i] = (uint16_t)i;
    auto f = [&](uint16_t id) -> uint32_t { return sup[id]; };
    IdSorter<decltype(f)> s{f, ids.data()};
    s.sort((long)n);
    return ids[0];
}

int h_overlap_estimate(uint32_t posH, uint32_t posV, uint32_t lenH, uint32_t lenV, int oriented, uint32_t k) {
    return overlap_estimate(posH, posV, lenH, lenV, oriented != 0, k);
}

void h_kmer_words(const uint32_t* packed, uint64_t g, uint32_t k, uint64_t* out) {
    const uint64_t le = kmer_le(packed, g, k);
    out[0] = le; out[1] = kmer_fw_from_le(le, k); out[2] = kmer_rc_from_le(le, k);
}

// the whole per-pair computation of k_xdrop (both lanes + finish_pair), serially
void h_xdrop_pair(const uint32_t* packed, uint64_t goffH, uint32_t lenH, uint64_t goffV, uint32_t lenV, uint32_t seedH,
                  uint32_t seedV, uint32_t k, int X, double phi, double delta, bella_aln* out) {
    PairGeom g;
    make_geom(packed, goffH, lenH, goffV, lenV, seedH, seedV, k, g);
    XRes res[2];
    bool ran[2] = {false, false};
    int8_t dp[132];
    for (int which = 0; which < 2; ++which) {
        SeqAcc H, V;
        make_accessors(packed, goffH, goffV, g, which, H, V);
        res[which].best = 0; res[which].endH = 0; res[which].endV = 0; res[which].flagged = 0; res[which].steps = 0;
        if (H.len >= (uint32_t)kXW && V.len >= (uint32_t)kXW) {
            ran[which] = true;
            xavier_one_direction(H, V, X, dp, 1, res[which]);
        }
    }
    finish_pair(g, ran[0], res[0], ran[1], res[1], phi, delta, *out);
}

// FASTQ ingest (fastq.hpp): the parse of a file, flattened.  Returns the read count or -1; names are joined with '\n'.
long h_parse_fastq(const char* path, uint8_t* bases, uint64_t bases_cap, uint64_t* offsets, uint64_t offsets_cap, char* names,
                   uint64_t names_cap, uint64_t* nbases) {
    FastqData fq;
    std::string err;
    if (parse_fastq(path, fq, err)) return -1;
    std::string joined;
    for (const auto& n : fq.names) { joined += n; joined += '\n'; }
    *nbases = fq.bases.size();
    if (fq.bases.size() > bases_cap || fq.offsets.size() > offsets_cap || joined.size() + 1 > names_cap) return -2;
    std::memcpy(bases, fq.bases.data(), fq.bases.size());
    std::memcpy(offsets, fq.offsets.data(), 8 * fq.offsets.size());
    std::memcpy(names, joined.c_str(), joined.size() + 1);
    return (long)fq.names.size();
}

// the same through the serial restatement (mode 0) or the index with `threads` threads over slices of `slice` bytes (mode 1);
// -1 with the message in `names` on a parse error.  gather_off/gather_n != 0: bases = that range of the base stream only.
long h_parse_fastq2(const char* path, int mode, unsigned threads, uint64_t slice, uint8_t* bases, uint64_t bases_cap, uint64_t* offsets,
                    uint64_t offsets_cap, char* names, uint64_t names_cap, uint64_t* nbases, uint64_t gather_off, uint64_t gather_n) {
    FastqData fq;
    std::string err;
    if (mode == 0) {
        if (parse_fastq_serial(path, fq, err)) { std::strncpy(names, err.c_str(), names_cap - 1); names[names_cap - 1] = 0; return -1; }
    } else {
        FastqIndex ix;
        if (ix.build(path, err, threads, (size_t)slice)) { std::strncpy(names, err.c_str(), names_cap - 1); names[names_cap - 1] = 0; return -1; }
        if (gather_n) {
            if (gather_n > bases_cap || gather_off + gather_n > ix.nbases()) return -2;
            ix.gather_parallel(bases, gather_off, gather_n);
            *nbases = gather_n;
            return (long)ix.nreads();
        }
        fq.bases.resize((size_t)ix.nbases());
        ix.gather_parallel(fq.bases.data(), 0, ix.nbases());
        fq.offsets = ix.offsets;
        fq.names = ix.names;
    }
    std::string joined;
    for (const auto& n : fq.names) { joined += n; joined += '\n'; }
    *nbases = fq.bases.size();
    if (fq.bases.size() > bases_cap || fq.offsets.size() > offsets_cap || joined.size() + 1 > names_cap) return -2;
    std::memcpy(bases, fq.bases.data(), fq.bases.size());
    std::memcpy(offsets, fq.offsets.data(), 8 * fq.offsets.size());
    std::memcpy(names, joined.c_str(), joined.size() + 1);
    return (long)fq.names.size();
}

void h_fastq_name(const char* header, char* out, uint64_t cap) {
    const std::string n = fastq_read_name(header);
    std::strncpy(out, n.c_str(), cap - 1);
    out[cap - 1] = 0;
}

}  // extern "C"
