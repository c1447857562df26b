// bella_hip_main.cpp -> bella_amd/bin/bella-hip: BELLA's command line with the WHOLE pipeline on the MI355X.
//
// The flags, the output file and the stdout protocol are those of the reference's src/main.cpp (:65-175 options, :130 output name,
// :472-473 nkmer, src/CSC.cpp:405 nnz(A), include/overlap.hpp:686 nnz(C), :771 lines written, main.cpp:532 run time); nothing of the
// reference runs: FASTQ list -> bella_hip_load_fastq_list (replaces ParallelFASTQ, main.cpp:339-360) -> bella_hip_count_kmers /
// _syncmers / _minimizers (SplitCount / SyncmerCount / MinimizerCount, include/kmercount.hpp:467-986, and the tuple loop,
// main.cpp:363-416) -> bella_hip_assemble_counted (CSC constructor + Transpose, main.cpp:476-489) -> the stage plan, HashSpGEMM,
// RunPairWiseAlignments and the writer of include/overlap.hpp:650-789 (bella_hip_driver.hpp).  One host thread and one context per
// GPU (-g): reads replicated, the dictionary counted across the contexts, B assembled by row blocks and exchanged device to device,
// output columns i % N == g per context.
//
// K-mer ids are labels: the reference numbers the reliable k-mers in libcuckoo's iteration order, this program in ascending order of
// the canonical word.  The ids decide the order of a column's products and through it the order of the output lines and, for pairs
// with several shared k-mers, count / seed (the reference's own output changes the same way with its thread
// count).  --tuples FILE takes the reference's `readbykmers.mtx` dump (include/common/bellaio.h:2-47: "nreads nkmers ntuples", then
// "read+1 kmer+1 pos" per line) instead of counting: same ids as that reference run, same output file byte for byte.
//
// Not built (rejected loudly): --hopc, --estimate, --split-count > 1.
#include <sys/stat.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "bella_hip.h"
#include "bella_hip_driver.hpp"

namespace {

struct Options {
    std::string fastq_list, output, tuples;
    int kmer = 17, xdrop = 7, memory = 8000, bin_size = 500, gpus = 1, split_count = 1, window = 0, upper = 8, lower = 2;
    double error = 0.15, deviation = 0.1;
    bool estimate = false, skip_alignment = false, paf = false, hopc = false, syncmer = false, help = false, exact = false;
};

const char* kHelp =
    "Long Read to Long Read Aligner and Overlapper (MI355X)\n"
    "Usage:\n"
    "  bella-hip [OPTION...]\n\n"
    "  -f, --fastq arg            List of Fastq(s) (required)\n"
    "  -o, --output arg           Output Filename (required)\n"
    "  -k, --kmer arg             K-mer Length (default: 17)\n"
    "  -x, --xdrop arg            X-Drop (default: 7)\n"
    "  -e, --error arg            Error Rate (default: 0.15)\n"
    "      --estimate             Estimate Error Rate from Data (not built)\n"
    "      --skip-alignment       Overlap Only\n"
    "  -m, --memory arg           Total RAM of the System in MB (default: 8000)\n"
    "      --score-deviation arg  Deviation from the Mean Alignment Score [0,1] (default: 0.1)\n"
    "  -b, --bin-size arg         Bin Size for Binning Algorithm (default: 500)\n"
    "      --paf                  Output in PAF format\n"
    "  -g, --gpus arg             GPUs Available (default: 1)\n"
    "      --split-count arg      K-mer Counting Split Count (only 1)\n"
    "      --hopc                 Use HOPC representation (not built)\n"
    "  -w, --window arg           Window Size for Minimizer Selection (default: 0)\n"
    "  -s, --syncmer              Enable Syncmer Selection\n"
    "  -u, --upper-freq arg       K-mer Frequency Upper Bound (default: 8)\n"
    "  -l, --lower-freq arg       K-mer Frequency Lower Bound (default: 2)\n"
    "      --tuples arg           readbykmers.mtx of a reference run: its k-mer ids instead of counting\n"
    "      --exact-xdrop          the exact (growing band) X-drop of the reference's GPU build instead of Xavier\n"
    "  -h, --help                 Usage\n";

[[noreturn]] void die(const std::string& msg) {
    std::cerr << "bella-hip: " << msg << std::endl;
    std::exit(1);
}

// cxxopts' forms (the reference's parser): --name value, --name=value, -n value, -nvalue; booleans take no value
Options parse(int argc, char** argv) {
    Options o;
    struct Spec { const char* lng; char sht; int kind; void* dst; };   // kind 0 bool, 1 int, 2 double, 3 string
    const Spec specs[] = {
        {"fastq", 'f', 3, &o.fastq_list}, {"output", 'o', 3, &o.output}, {"kmer", 'k', 1, &o.kmer}, {"xdrop", 'x', 1, &o.xdrop},
        {"error", 'e', 2, &o.error}, {"estimate", 0, 0, &o.estimate}, {"skip-alignment", 0, 0, &o.skip_alignment}, {"memory", 'm', 1, &o.memory},
        {"score-deviation", 0, 2, &o.deviation}, {"bin-size", 'b', 1, &o.bin_size}, {"paf", 0, 0, &o.paf}, {"gpus", 'g', 1, &o.gpus},
        {"split-count", 0, 1, &o.split_count}, {"hopc", 0, 0, &o.hopc}, {"window", 'w', 1, &o.window}, {"syncmer", 's', 0, &o.syncmer},
        {"upper-freq", 'u', 1, &o.upper}, {"lower-freq", 'l', 1, &o.lower}, {"tuples", 0, 3, &o.tuples}, {"exact-xdrop", 0, 0, &o.exact},
        {"help", 'h', 0, &o.help}};
    auto assign = [&](const Spec& s, const char* v, const std::string& shown) {
        char* end = nullptr;
        if (s.kind == 1) { const long x = std::strtol(v, &end, 10); if (!*v || *end) die("option " + shown + ": '" + v + "' is not an integer"); *(int*)s.dst = (int)x; }
        else if (s.kind == 2) { const double x = std::strtod(v, &end); if (!*v || *end) die("option " + shown + ": '" + v + "' is not a number"); *(double*)s.dst = x; }
        else *(std::string*)s.dst = v;
    };
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        const Spec* sp = nullptr;
        const char* val = nullptr;
        std::string holder;
        if (a.size() > 2 && a[0] == '-' && a[1] == '-') {
            const size_t eq = a.find('=');
            const std::string name = a.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
            for (const Spec& s : specs) if (name == s.lng) sp = &s;
            if (!sp) die("option '" + a + "' does not exist");
            if (eq != std::string::npos) { holder = a.substr(eq + 1); val = holder.c_str(); }
        } else if (a.size() >= 2 && a[0] == '-') {
            for (const Spec& s : specs) if (s.sht && a[1] == s.sht) sp = &s;
            if (!sp) die("option '" + a + "' does not exist");
            if (a.size() > 2) { holder = a.substr(2); val = holder.c_str(); }
        } else die("unexpected argument '" + a + "'");
        if (sp->kind == 0) {
            if (val) die("option " + a + " takes no value");
            *(bool*)sp->dst = true;
            continue;
        }
        if (!val) {
            if (i + 1 >= argc) die("option " + a + " is missing an argument");
            val = argv[++i];
        }
        assign(*sp, val, a);
    }
    return o;
}

// GetFiles (include/kmercount.hpp:82-105): one path per line; a last line without '\n' is not read by the reference either
std::vector<std::string> read_list(const std::string& path) {
    std::ifstream f(path);
    if (!f.is_open()) die("Could not open " + path);
    std::vector<std::string> out;
    std::string line;
    while (std::getline(f, line)) {
        if (f.eof()) break;                                         // (no trailing newline: getline hit the end of the file in this line)
        if (!line.empty()) out.push_back(line);
    }
    return out;
}

struct MtxTuples { uint32_t nreads = 0, nkmers = 0; std::vector<uint32_t> kmer, read; std::vector<uint16_t> pos; };
MtxTuples read_mtx(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) die("Could not open " + path);
    MtxTuples t;
    unsigned long long nr = 0, nk = 0, nt = 0;
    if (std::fscanf(f, "%llu %llu %llu", &nr, &nk, &nt) != 3) die(path + ": not a readbykmers.mtx (header: nreads nkmers ntuples)");
    t.nreads = (uint32_t)nr; t.nkmers = (uint32_t)nk;
    t.kmer.reserve(nt); t.read.reserve(nt); t.pos.reserve(nt);
    unsigned long long r = 0, k = 0, p = 0;
    while (std::fscanf(f, "%llu %llu %llu", &r, &k, &p) == 3) {
        if (!r || !k || r > nr || k > nk || p > 65535) die(path + ": tuple out of range");
        t.read.push_back((uint32_t)(r - 1)); t.kmer.push_back((uint32_t)(k - 1)); t.pos.push_back((uint16_t)p);
    }
    std::fclose(f);
    if (t.kmer.size() != nt) die(path + ": " + std::to_string(t.kmer.size()) + " tuples, header says " + std::to_string(nt));
    for (size_t i = 1; i < t.read.size(); ++i)
        if (t.read[i] < t.read[i - 1]) die(path + ": tuples must be grouped by non-decreasing read (the reference's 1-thread dump is)");
    return t;
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

int main(int argc, char** argv) {
    using namespace bella_hip_detail;
    const Options o = parse(argc, argv);
    if (o.help || o.fastq_list.empty() || o.output.empty()) { std::cout << kHelp << std::endl; return 0; }     // main.cpp:97-145
    if (o.deviation > 1.0 || o.deviation < 0.0) { std::cout << kHelp << std::endl; return 0; }                 // main.cpp:159-163
    if (o.hopc) die("--hopc is not built (the HOPC representation, include/kmercount.hpp: outside this engine's parity contract)");
    if (o.estimate) die("--estimate is not built (error-rate estimation from the quality strings): pass -e");
    if (o.split_count != 1) die("--split-count > 1 is not built: the device counter takes the whole k-mer space in one call (or in passes of its own)");
    if (o.kmer < 1 || o.kmer > 32) die("-k must be in [1,32] (one 64-bit word per k-mer, Kmer.hpp:27-28)");
    const std::string outfile = o.output + ".out";                  // main.cpp:113-130
    std::remove(outfile.c_str());
    const char* const tag = "bella_hip_main.cpp";
    const double all = now_s();

    const std::vector<std::string> files = read_list(o.fastq_list);
    if (files.empty()) die("no FASTQ file in " + o.fastq_list + " (every line must end in a newline, kmercount.hpp:96)");
    uint64_t file_bytes = 0;
    for (const std::string& f : files) {
        struct stat st;
        if (stat(f.c_str(), &st) != 0) die("Could not open " + f);
        file_bytes += (uint64_t)st.st_size;
        const std::string InputFile = f;
        BELLA_HIP_LOGT(tag, InputFile);
    }
    const int ndev = bella_hip_device_count();
    if (ndev <= 0) check(nullptr, BELLA_ERR_NO_DEVICE, "bella_hip_device_count");
    int N = o.gpus > 1 ? o.gpus : 1;
    if (N > ndev && !std::getenv("BELLA_HIP_OVERSUBSCRIBE")) N = ndev;   // (tests on a one-GPU box: contexts share the device)
    const int GPUs = N;
    BELLA_HIP_LOGT(tag, GPUs);

    // selector of the k-mers that make tuples: syncmers win over minimizers as in main.cpp:170-176
    const bool use_sync = o.syncmer, use_min = !o.syncmer && o.window != 0;
    MtxTuples mtx;
    if (!o.tuples.empty()) mtx = read_mtx(o.tuples);

    std::vector<Worker> W((size_t)N);
    std::vector<const char*> cpaths;
    for (const std::string& f : files) cpaths.push_back(f.c_str());
    uint8_t comm_id[BELLA_HIP_COMM_ID_BYTES];
    if (N > 1) check(nullptr, bella_hip_comm_id_local(comm_id), "bella_hip_comm_id_local");
    std::vector<uint32_t> nk_of((size_t)N, 0), nreads_of((size_t)N, 0);
    std::vector<double> t_ingest((size_t)N, 0), t_count((size_t)N, 0), t_asm((size_t)N, 0), t_init((size_t)N, 0);
    on_all(N, [&](int g) {
        Worker& w = W[(size_t)g];
        double t0 = now_s();
        check(nullptr, bella_hip_init(g % ndev, &w.ctx), "bella_hip_init");
        reserve_for(w.ctx, file_bytes / 2, (N + ndev - 1) / ndev);   // (a FASTQ file is half bases, half qualities)
        if (const char* e = std::getenv("BELLA_HIP_KCOUNT_BUDGET")) {
            const uint64_t v = std::strtoull(e, nullptr, 10);
            check(w.ctx, bella_hip_set_tuning(w.ctx, BELLA_TUNE_KCOUNT_BUDGET, &v, 1), "bella_hip_set_tuning");
        }
        t_init[(size_t)g] = now_s() - t0;
        t0 = now_s();
        uint32_t nreads = 0;
        uint64_t nbases = 0;
        check(w.ctx, bella_hip_load_fastq_list(w.ctx, cpaths.data(), (uint32_t)cpaths.size(), &nreads, &nbases), "bella_hip_load_fastq_list");
        nreads_of[(size_t)g] = nreads;
        t_ingest[(size_t)g] = now_s() - t0;
        t0 = now_s();
        const uint32_t lo = (uint32_t)((uint64_t)nreads * (uint64_t)g / (uint64_t)N), hi = (uint32_t)((uint64_t)nreads * (uint64_t)(g + 1) / (uint64_t)N);
        uint32_t nk = 0;
        if (N > 1) {
            check(w.ctx, bella_hip_set_partition(w.ctx, (uint32_t)g, (uint32_t)N), "bella_hip_set_partition");
            check(w.ctx, bella_hip_comm_init_local(w.ctx, N, g, comm_id), "bella_hip_comm_init_local");
        }
        if (!o.tuples.empty()) {
            if (mtx.nreads != nreads) die("--tuples: the dump holds " + std::to_string(mtx.nreads) + " reads, the FASTQ list " + std::to_string(nreads));
            nk = mtx.nkmers;
            t_count[(size_t)g] = 0;
            t0 = now_s();
            if (N == 1) {
                check(w.ctx, bella_hip_assemble_tuples(w.ctx, (uint16_t)o.kmer, nk, mtx.kmer.size(), mtx.kmer.data(), mtx.read.data(), mtx.pos.data()), "bella_hip_assemble_tuples");
            } else {
                const size_t a = (size_t)(std::lower_bound(mtx.read.begin(), mtx.read.end(), lo) - mtx.read.begin());
                const size_t b = (size_t)(std::lower_bound(mtx.read.begin(), mtx.read.end(), hi) - mtx.read.begin());
                check(w.ctx, bella_hip_assemble_panel(w.ctx, (uint16_t)o.kmer, nk, lo, hi - lo, b - a, mtx.kmer.data() + a, mtx.read.data() + a, mtx.pos.data() + a), "bella_hip_assemble_panel");
                check(w.ctx, bella_hip_allgather_panels(w.ctx), "bella_hip_allgather_panels");
            }
        } else {
            uint64_t nt = 0, nd = 0;
            if (N > 1)
                check(w.ctx, bella_hip_count_kmers_dist(w.ctx, (uint16_t)o.kmer, (uint32_t)o.lower, (uint32_t)o.upper, use_sync ? 1u : use_min ? 2u : 0u,
                                                        (uint32_t)o.window, lo, hi - lo, &nk, &nt, &nd), "bella_hip_count_kmers_dist");
            else if (use_sync) check(w.ctx, bella_hip_count_syncmers(w.ctx, (uint16_t)o.kmer, (uint32_t)o.lower, (uint32_t)o.upper, &nk, &nt, &nd), "bella_hip_count_syncmers");
            else if (use_min) check(w.ctx, bella_hip_count_minimizers(w.ctx, (uint16_t)o.kmer, (uint32_t)o.window, (uint32_t)o.lower, (uint32_t)o.upper, &nk, &nt, &nd), "bella_hip_count_minimizers");
            else check(w.ctx, bella_hip_count_kmers(w.ctx, (uint16_t)o.kmer, (uint32_t)o.lower, (uint32_t)o.upper, &nk, &nt, &nd), "bella_hip_count_kmers");
            t_count[(size_t)g] = now_s() - t0;
            t0 = now_s();
            if (N == 1) check(w.ctx, bella_hip_assemble_counted(w.ctx), "bella_hip_assemble_counted");
            else {
                check(w.ctx, bella_hip_assemble_counted_panel(w.ctx, lo, hi - lo), "bella_hip_assemble_counted_panel");
                check(w.ctx, bella_hip_allgather_panels(w.ctx), "bella_hip_allgather_panels");
            }
        }
        t_asm[(size_t)g] = now_s() - t0;
        nk_of[(size_t)g] = nk;
    });
    const uint32_t nreads = nreads_of[0];
    const std::string ContextAndReservationTime = std::to_string(t_init[0]) + " seconds";
    BELLA_HIP_LOGT(tag, ContextAndReservationTime);
    const std::string fastqParsingTime = std::to_string(t_ingest[0]) + " seconds";
    BELLA_HIP_LOGT(tag, fastqParsingTime);
    const uint32_t numReads = nreads;
    BELLA_HIP_LOGT(tag, numReads);
    const std::string KmerCountingTime = std::to_string(t_count[0]) + " seconds";
    BELLA_HIP_LOGT(tag, KmerCountingTime);
    std::cout << nk_of[0] << std::endl;                              // main.cpp:472-473 ("to help the parsing script")
    uint64_t nnzA = 0;
    check(W[0].ctx, bella_hip_get_B(W[0].ctx, &nnzA, nullptr, nullptr, nullptr), "bella_hip_get_B");
    std::cout << nnzA << std::endl;                                  // src/CSC.cpp:405 (nnz after MergeDuplicates)
    const std::string SparseMatrixCreationTime = std::to_string(t_asm[0]) + " seconds";
    BELLA_HIP_LOGT(tag, SparseMatrixCreationTime);

    // names and lengths for the writer
    uint64_t need = 0;
    check(W[0].ctx, bella_hip_get_read_names(W[0].ctx, nullptr, 0, nullptr, &need), "bella_hip_get_read_names");
    std::vector<char> namebuf((size_t)need + 1);
    std::vector<uint64_t> noffs((size_t)nreads + 1, 0);
    check(W[0].ctx, bella_hip_get_read_names(W[0].ctx, namebuf.data(), need, noffs.data(), &need), "bella_hip_get_read_names");
    std::vector<const char*> names(nreads);
    for (uint32_t r = 0; r < nreads; ++r) names[r] = namebuf.data() + noffs[r];
    std::vector<uint32_t> lens(nreads);
    if (nreads) check(W[0].ctx, bella_hip_get_read_lengths(W[0].ctx, lens.data()), "bella_hip_get_read_lengths");

    StageOpts so;
    so.N = N;
    so.nreads = nreads;
    so.p.kmer_size = (uint16_t)o.kmer;
    so.p.bin_size = (uint16_t)o.bin_size;
    so.p.xdrop = (uint16_t)o.xdrop;
    so.p.skip_alignment = o.skip_alignment ? 1 : 0;
    so.p.error_rate = o.error;
    so.p.delta_chernoff = o.deviation;
    so.paf = o.paf ? 1 : 0;
    so.total_memory_mb = (double)o.memory;
    so.per_nnz = 20.0;                                               // sizeof(spmatPtr_) + sizeof(uint32_t) in the reference (overlap.hpp:365-404)
    so.filename = outfile.c_str();
    so.tag = tag;
    so.exact = o.exact ? 1 : 0;
    const double t_stages = now_s();
    run_stages(W, so, names.data(), lens.data());
    {
        const CallStats& cs = last_call_stats();
        const std::string OverlapTime = std::to_string(cs.overlap_seconds) + " seconds";            // (overlap.hpp:714-727's bracket)
        BELLA_HIP_LOGT(tag, OverlapTime);
        const std::string RecordsAndAlignmentTime = std::to_string(cs.align_seconds) + " seconds";    // records to the host (+ X-drop + alignments to the host)
        BELLA_HIP_LOGT(tag, RecordsAndAlignmentTime);
        const std::string WriterTime = std::to_string(cs.write_seconds) + " seconds";
        BELLA_HIP_LOGT(tag, WriterTime);
        const std::string StagesTime = std::to_string(now_s() - t_stages) + " seconds";
        BELLA_HIP_LOGT(tag, StagesTime);
    }
    const double t_close = now_s();
    for (auto& w : W) bella_hip_destroy(w.ctx);
    const std::string TeardownTime = std::to_string(now_s() - t_close) + " seconds";
    BELLA_HIP_LOGT(tag, TeardownTime);

    const double totaltime = now_s() - all;
    const std::string TotalRuntime = std::to_string(totaltime) + " seconds";
    BELLA_HIP_LOGT(tag, TotalRuntime);
    std::cout << totaltime << std::endl;                             // main.cpp:532
    return 0;
}
