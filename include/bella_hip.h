/*
 * bella_hip.h -- C ABI of the MI355X-native BELLA overlap engine (libbella_hip.so).
 *
 * The reference (PASSIONLab/BELLA) has no FFI layer: its boundary for this path is two header-level
 * C++ call sites.  Each entry point below names the reference interface it replaces (paths relative
 * to the reference tree).  INTEGRATION.md shows the shim a BELLA maintainer adds so that
 * src/main.cpp:498-525 calls these instead of include/overlap.hpp's HashSpGEMM.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 or a negative BELLA_ERR_*;
 * no exceptions or exit() cross the ABI; inputs are borrowed for the duration of the call; results
 * live in the context (device memory) until overwritten or destroyed and are copied out by the
 * bella_hip_get_* calls into caller-owned buffers.  One host thread per context.
 * The library never falls back to a CPU implementation: without a gfx950 device
 * bella_hip_init returns BELLA_ERR_NO_DEVICE.
 */
#ifndef BELLA_HIP_H
#define BELLA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BELLA_HIP_ABI_VERSION 6

enum {
    BELLA_OK = 0,
    BELLA_ERR_NO_DEVICE = -1,     /* no HIP device / not gfx950                                         */
    BELLA_ERR_HIP = -2,           /* a HIP runtime call failed (bella_hip_last_error has the text)       */
    BELLA_ERR_BAD_ARG = -3,       /* incl. a k-mer present in more than 16,383 reads                     */
    BELLA_ERR_BAD_BASE = -4,      /* read contains a character other than ACGT (align.hpp:40-55 asserts) */
    BELLA_ERR_READ_TOO_LONG = -5, /* read >= 65,536 bases: u16 positions (common.h:122-126)             */
    BELLA_ERR_TUPLE_ORDER = -6,   /* tuples not grouped by non-decreasing read id                       */
    BELLA_ERR_STATE = -7,         /* call order: reads -> matrix -> overlap -> align                    */
    BELLA_ERR_ROW_TOO_LARGE = -8, /* one output column has >= 2^31 products                             */
    BELLA_ERR_BINS = -9,          /* a pair ended with > 16 overlap bins: std::sort tie order path      */
    BELLA_ERR_NOMEM = -10
};

typedef struct bella_ctx bella_ctx;

/* BELLApars (include/common/common.h:46-74), the fields the hot path reads. */
typedef struct {
    uint16_t kmer_size;       /* -k, kmerSize      (<= 32)                       */
    uint16_t bin_size;        /* -b, binSize       (chain.hpp:114)               */
    uint16_t xdrop;           /* -x, xDrop         (align.hpp:152)               */
    uint16_t skip_alignment;  /* --skip-alignment  (overlap.hpp:542,577)         */
    double error_rate;        /* -e : ratiophi = slope(e) (align.hpp:72-80)      */
    double delta_chernoff;    /* --score-deviation (overlap.hpp:456)             */
} bella_params;

/* One nonzero of C = A*A^T after the semiring fold, i.e. what RunPairWiseAlignments
 * (overlap.hpp:531-585) reads from a spmatType_ (common.h:119-183): count, and choose()'s seed.
 * Order of the array = the reference's 1-thread output order: column (cid) ascending, hash-slot
 * order inside a column (overlap.hpp:343-361). */
typedef struct {
    uint32_t rid;     /* row of C: the larger read id; "read1"/H in chain.hpp            */
    uint32_t cid;     /* column of C: read i; "read2"/V                                   */
    uint16_t count;   /* spmatType_::count (u16 wrap kept)                                */
    uint16_t seedH;   /* choose().first  : seed k-mer position on read rid               */
    uint16_t seedV;   /* choose().second : seed k-mer position on read cid               */
    uint16_t flags;   /* bit0: seed k-mers identical (checkstrand true, chain.hpp:35-44)
                         bit1: revcomp(seedH)==seedV (strand "c", align.hpp:171)          */
} bella_pair;

/* Diagnostics of the final semiring value (tests compare them with the oracle). */
typedef struct {
    uint16_t nbins;    /* pos.size() at the end                                           */
    uint16_t support;  /* chain(): support of the winning bin (common.h:142-150)          */
    uint16_t binov;    /* overlap[] of the winning bin                                    */
    uint16_t pad;
} bella_pair_ext;

/* xavierResult (common.h:83-87) + what PostAlignDecision (overlap.hpp:413-497) derives from it. */
typedef struct {
    int32_t score;            /* best1 + best2 (simdutils.h:333-337)                      */
    int32_t begH, endH;       /* SeedX positions on (possibly reverse-complemented) read rid */
    int32_t begV, endV;       /* on read cid                                              */
    uint16_t ov;              /* overlap estimate `ov` (overlap.hpp:449)                   */
    uint8_t strand;           /* 0 = "n", 1 = "c"                                          */
    uint8_t passed;           /* (float)score >= (1-delta)*phi*ov (overlap.hpp:456-460)    */
    uint32_t steps;           /* anti-diagonal steps taken (both directions): GCUPS = 31*steps */
    uint32_t flagged;         /* 1 if an extension began with no positive lane: the reference reads an
                                 uninitialised `maxpos` there (xavier.h:165); we use 0 */
} bella_aln;

/* Explicit seed for the batched xavierAlign (the alignLogan-shaped entry, align.hpp:210-211). */
typedef struct {
    uint32_t rid, cid;        /* reads previously given to bella_hip_set_reads             */
    uint16_t seedH, seedV;
} bella_seed;

typedef struct {
    float assemble_ms;        /* tuples/B -> device CSR layout (all assembly kernels)      */
    float symbolic_ms;        /* per-row flops + tiering (estimateFLOP, overlap.hpp:157)   */
    float spgemm_ms;          /* row kernels: symbolic + expansion + slot order (overlap.hpp:205,281) */
    float fold_ms;            /* semiring fold kernels (chain.hpp:74-150)                    */
    float compact_ms;         /* pair compaction to the dense output                        */
    float xdrop_ms;           /* X-drop kernel                                              */
    float overlap_total_ms;   /* bella_hip_overlap, stream time start to end                */
    uint32_t spgemm_launches; /* row-kernel launches (one per non-empty LDS tier)            */
    float kcount_ms;          /* bella_hip_count_kmers: counting + dictionary + tuples      */
    uint32_t retry_columns;   /* last pass: columns an LDS tier handed to the global-workspace path (key table too small for
                                 the column's pairs, or a product list out of order -- see DESIGN 4.1, phase S).  Normally 0 or a
                                 handful; a large value is a performance cliff worth reporting                              */
    uint32_t overflow_pairs;  /* last pass: pairs that ended with > 16 bins (serial fold with libstdc++'s sort order)       */
    float layout_ms;          /* part of assemble_ms: CSR of B -> device layout B' / A' (sort + segmented passes)           */
    float rows_ms;            /* part of assemble_ms: tuples -> rows of B in MergeDuplicates slot order (CSC.cpp:301-420)    */
    uint32_t lane_order;      /* init-time self-test of the LDS-atomic lane order the LDS tiers rely on (phase S):
                                 1 = holds; 2 = does not hold on this device/driver: every column takes the repairing
                                 global-workspace path (correct, slower)                                                 */
    float expand_ms;          /* part of layout_ms: the product expansion done at assembly time (the row lists of
                                 BELLA_TUNE_ROW_LISTS; 0 in the default layout, where every pass expands B' x A' itself)   */
    uint32_t numeric_passes;  /* since bella_hip_init: calls of bella_hip_overlap that ran the numeric phase ...          */
    uint32_t symbolic_passes; /* ... and calls of bella_hip_count_pairs (symbolic phase only)                            */
    uint32_t pad;
    uint64_t numeric_columns; /* since bella_hip_init: output columns the numeric phase computed (a staged run that computes
                                 every column once ends at nreads)                                                        */
} bella_timings;

/* ---- lifecycle ---------------------------------------------------------------------------------- */
int bella_hip_abi_version(void);
int bella_hip_device_count(void);
int bella_hip_init(int device, bella_ctx** out);
void bella_hip_destroy(bella_ctx* ctx);
const char* bella_hip_strerror(int code);
const char* bella_hip_last_error(const bella_ctx* ctx);

/* ---- reads: readVector_ (common.h:98-109) -------------------------------------------------------- */
/* `bases` = all reads concatenated, upper-case ASCII ACGT; offsets has nreads+1 entries. */
int bella_hip_set_reads(bella_ctx* ctx, const uint8_t* bases, const uint64_t* offsets, uint32_t nreads);
/* FASTQ ingest: replaces ParallelFASTQ::fill_block / get_next_fq_record (kmercode/fq_reader.c:540-610) and the
 * name handling of get_fq_name (fq_reader.c:88-130) + src/main.cpp:352-360 for one plain (not compressed, as in the
 * reference's NO_GZIP build) 4-line FASTQ file: the file is mapped and indexed on all host cores (records are found by line
 * number, not by looking for '@'), the bases stream mapping -> pinned buffer -> device in 64 MB chunks (gather of chunk i+1 under
 * the transfer of chunk i) and are packed to 2 bit/base on the device.  Same effect as bella_hip_set_reads; the names (without
 * '@', cut at the comment as the reference does) stay in the context. */
int bella_hip_load_fastq(bella_ctx* ctx, const char* path, uint32_t* nreads, uint64_t* nbases);
/* the same for the reference's LIST of FASTQ files (-f: include/kmercount.hpp:82-105 GetFiles reads one path per line; src/main.cpp:339-423
 * numbers the reads through the files in list order): read ids continue from file to file, every file is mapped and indexed for itself. */
int bella_hip_load_fastq_list(bella_ctx* ctx, const char* const* paths, uint32_t nfiles, uint32_t* nreads, uint64_t* nbases);
/* what the last bella_hip_load_fastq / _list did (file_bytes, threads: over all files) */
typedef struct bella_ingest_stats {
    uint64_t file_bytes;
    uint64_t bases;
    uint32_t reads;
    uint32_t threads;     /* host threads of the index and of the gather */
    double index_ms;      /* map + both passes over the file */
    double upload_ms;     /* gather + host->device + pack, until the packed reads are on the device */
} bella_ingest_stats;
int bella_hip_get_ingest_stats(bella_ctx* ctx, bella_ingest_stats* out);
/* names back to back, NUL-terminated; offsets[nreads+1] into buf; *needed = bytes required (call with buf = NULL first) */
int bella_hip_get_read_names(bella_ctx* ctx, char* buf, uint64_t buflen, uint64_t* offsets, uint64_t* needed);
int bella_hip_get_read_lengths(bella_ctx* ctx, uint32_t* lens);

/* ---- k-mer counting, reliable dictionary, tuple generation ---------------------------- */
/* Replaces SplitCount (include/kmercount.hpp:467-677) and the tuple loop of src/main.cpp:393-416 on the reads given to
 * bella_hip_set_reads: every position j <= len-k contributes Kmer::rep() (kmercode/Kmer.cpp:314-317); a k-mer is reliable
 * when lower <= (occurrences mod 65536) <= upper (the reference counts in an unsigned short, kmercount.hpp:632-655;
 * lower >= 2 as in the reference's defaults: its table only holds k-mers seen twice).  K-mer ids are labels: the reference
 * numbers in libcuckoo's iteration order, this library in ascending order of the canonical word (first base most
 * significant, A<C<G<T, the order of Kmer::operator<).  The tuples (id, read, position) are generated in the reference's
 * order -- read by read, positions ascending -- and stay on the device for bella_hip_assemble_counted.
 * Out: *nkmers = dictionary size, *ntuples, *ndistinct = distinct canonical k-mers seen (the reference's HyperLogLog
 * estimates this number, kmercount.hpp:585-590; here it is exact).  Any of them may be NULL. */
int bella_hip_count_kmers(bella_ctx* ctx, uint16_t kmer_size, uint32_t lower, uint32_t upper, uint32_t* nkmers,
                          uint64_t* ntuples, uint64_t* ndistinct);
/* The same for the reference's syncmer mode (-s): SyncmerCount (include/kmercount.hpp:845-985) with isSyncmer
 * (include/syncmer.hpp:47-79, s = 5, Kmer::hash = MurmurHash3_x64_64 seed 313) + the tuple loop of src/main.cpp:393-416.
 * Counted: the strand-specific word of every syncmer position, saturating at 65535; the dictionary holds those words; tuples:
 * every position whose CANONICAL k-mer is a dictionary key (the reference's tuple loop has no syncmer branch).  k > 5. */
int bella_hip_count_syncmers(bella_ctx* ctx, uint16_t kmer_size, uint32_t lower, uint32_t upper, uint32_t* nkmers,
                             uint64_t* ntuples, uint64_t* ndistinct);
/* The reference's minimizer mode (-w window): MinimizerCount (include/kmercount.hpp:691-835) with getMinimizers
 * (include/minimizer.hpp:49-79: monotone deque on rep().hash(), robust winnowing, the first `window` k-mers of a read are
 * never sampled -- the reference's size_t range test) + the minimizer branch of the tuple loop (src/main.cpp:363-388).
 * Counted: rep() of the minimizer positions, saturating at 65535; tuples: the minimizer positions whose rep() is reliable. */
int bella_hip_count_minimizers(bella_ctx* ctx, uint16_t kmer_size, uint32_t window, uint32_t lower, uint32_t upper,
                               uint32_t* nkmers, uint64_t* ntuples, uint64_t* ndistinct);
/* codes[nkmers]: the dictionary's words (canonical for count_kmers, strand-specific for count_syncmers), ascending (id = index),
 * right-aligned in 2k bits; counts[nkmers].  Either may be NULL. */
int bella_hip_get_dictionary(bella_ctx* ctx, uint64_t* codes, uint16_t* counts);
/* the tuple list of the last bella_hip_count_kmers (what the reference's alltranstuples holds, main.cpp:339-423) */
int bella_hip_get_tuples(bella_ctx* ctx, uint32_t* t_kmer, uint32_t* t_read, uint16_t* t_pos);
/* bella_hip_assemble_tuples on the device-resident tuples of bella_hip_count_kmers (no host copy) */
int bella_hip_assemble_counted(bella_ctx* ctx);
/* bella_hip_assemble_panel on the device-resident tuples of the reads [first_read, first_read + nreads_panel) (multi-GPU:
 * every rank counts, each assembles the rows of its own read block; then the all-gather of bella_hip_panel_device_ptrs) */
int bella_hip_assemble_counted_panel(bella_ctx* ctx, uint32_t first_read, uint32_t nreads_panel);

/* ---- operands ------------------------------------------------------------------------------------ */
/* From the (kmer, read, pos) tuple list: replaces the CSC tuple constructor + MergeDuplicates +
 * Transpose of src/main.cpp:476-489 (src/CSC.cpp:422-479,301-420; include/common/transpose.h:13).
 * Tuples must be grouped by non-decreasing read id, in generation order inside a read.
 * kmer_size = Kmer::set_k (main.cpp:183): needed here because the strand test of multiop
 * (chain.hpp:35-44) is precomputed as one orientation bit per nonzero. */
int bella_hip_assemble_tuples(bella_ctx* ctx, uint16_t kmer_size, uint32_t nkmers, uint64_t ntuples,
                              const uint32_t* t_kmer, const uint32_t* t_read, const uint16_t* t_pos);
/* From the reference's own B = transpmat CSC arrays (the HashSpGEMM boundary, overlap.hpp:650):
 * colptr[nreads+1], rowids = k-mer ids in MergeDuplicates slot order, values = positions.
 * A = spmat is derived on device (ascending read ids per k-mer = the reference's 1-thread Transpose).
 * Every call that installs operands (this one, the assemble_* calls, bella_hip_allgather_panels) also lays them out for the passes:
 * A' (whole) and B' (the columns of the context's partition, bella_hip_set_partition); every pass expands B' x A' itself.  With
 * BELLA_TUNE_ROW_LISTS (callers that run several passes over the same columns) also the row lists -- the two operand entries of every
 * product side by side in product order, 10 bytes per product -- when they fit.  Same results either way. */
int bella_hip_set_B(bella_ctx* ctx, uint16_t kmer_size, uint32_t nkmers, const uint32_t* colptr,
                    const uint32_t* rowids, const uint16_t* values);
/* Multi-GPU assembly: rank r builds only the rows of B of ITS reads (a row-block panel) from their tuples (global read ids,
 * same rules as above); the panels' device arrays are exchanged with one all-gather (RCCL over xGMI, done by the caller:
 * bella_amd/dist.py) and the full matrix comes back through bella_hip_set_B_device.  Rows = per-read entry counts (u32),
 * rowids = k-mer ids in MergeDuplicates slot order (u32), values = positions (u16). */
int bella_hip_assemble_panel(bella_ctx* ctx, uint16_t kmer_size, uint32_t nkmers, uint32_t first_read, uint32_t nreads_panel,
                             uint64_t ntuples, const uint32_t* t_kmer, const uint32_t* t_read, const uint16_t* t_pos);
/* The same panel from the reference's CSC arrays of B (colptr[nreads + 1], rowids, values: the WHOLE matrix on the host; only the
 * block's slice is uploaded).  A host program that holds B and drives several GPUs gives every context its block this way and lets
 * bella_hip_allgather_panels move the blocks device to device, instead of uploading the whole matrix once per GPU. */
int bella_hip_set_B_panel(bella_ctx* ctx, uint16_t kmer_size, uint32_t nkmers, uint32_t first_read, uint32_t nreads_panel,
                          const uint32_t* colptr, const uint32_t* rowids, const uint16_t* values);
int bella_hip_panel_device_ptrs(bella_ctx* ctx, uint32_t* first_read, uint32_t* nreads_panel, uint64_t* nnz,
                                const void** d_rowcnt, const void** d_rowids, const void** d_values);
/* As bella_hip_set_B, from DEVICE pointers (colptr[nreads+1], rowids[nnz], values[nnz]); copied, the caller keeps ownership. */
int bella_hip_set_B_device(bella_ctx* ctx, uint16_t kmer_size, uint32_t nkmers, const uint32_t* d_colptr, const uint32_t* d_rowids,
                           const uint16_t* d_values, uint64_t nnz);
/* Copies B back in the reference's layout (tests: compare with CSC.cpp's result). Any pointer may be NULL. */
int bella_hip_get_B(bella_ctx* ctx, uint64_t* nnz, uint32_t* colptr, uint32_t* rowids, uint16_t* values);

/* Multi-GPU (one context per GPU/process): this context computes output columns i with
 * i % stride == first.  Default (0,1) = all columns.  The device layout follows the partition: operands installed AFTER this call
 * get B' entries (and row lists) for the OWNED columns only -- per-column layout work and memory are 1/stride of the whole; A' (the
 * k-mer -> reads lists every column gathers from) and the exchanged reference-layout B stay whole.  Changing the partition of a
 * context whose layout was built for another one rebuilds the layout from the resident B at the next pass. */
int bella_hip_set_partition(bella_ctx* ctx, uint32_t first, uint32_t stride);
/* Stages (the reference forms the output in stages of consecutive columns when it does not fit the -m budget:
 * estimateMemory, overlap.hpp:365-404, stage loop :682-789): the next passes compute only the output columns
 * [first, first + count) (intersected with the partition above); default = all.  Running the ranges in ascending order and
 * appending their outputs reproduces the single-stage output, and every pass only holds its own columns' records. */
int bella_hip_set_column_range(bella_ctx* ctx, uint32_t first, uint32_t count);

/* ---- HashSpGEMM (overlap.hpp:650-789): estimateFLOP + estimateNNZ_Hash + LocalSpGEMM ---------------- */
int bella_hip_overlap(bella_ctx* ctx, const bella_params* p, uint64_t* npairs, uint64_t* flops);
/* The symbolic phase alone: estimateFLOP (overlap.hpp:157-202) + estimateNNZ_Hash (:205-276) + prefixsum (:110-146) over the
 * columns of the current partition and column range -- distinct row ids per output column, no product lists, no fold, no records,
 * and no product-sized buffers (a bitmap over the reads per workgroup).  colptrC[nreads + 1] (host, nullable): exclusive prefix sums
 * of the per-column pair counts (columns outside the partition / range count 0); *npairs = nnz(C), *flops = products.  This is what
 * the reference knows BEFORE its numeric phase and sizes its stages from (overlap.hpp:674-710); the shim's stage planner uses it
 * so that no column is computed twice.  With colptrC == NULL and npairs == NULL only estimateFLOP runs (*flops; sums over the
 * count stream, no gather): an upper bound of nnz(C) that lets a planner skip the symbolic phase when one stage is certain. */
int bella_hip_count_pairs(bella_ctx* ctx, const bella_params* p, uint64_t* colptrC, uint64_t* npairs, uint64_t* flops);
/* pairs[npairs]; ext (nullable) [npairs]; colptrC (nullable) [nreads+1] */
int bella_hip_get_pairs(bella_ctx* ctx, bella_pair* pairs, bella_pair_ext* ext, uint64_t* colptrC);

/* ---- RunPairWiseAlignments (overlap.hpp:499-645): xavierAlign + PostAlignDecision on every pair ----- */
int bella_hip_align_pairs(bella_ctx* ctx, const bella_params* p, uint64_t* npassed);
int bella_hip_get_alignments(bella_ctx* ctx, bella_aln* out);
/* xavierAlign (align.hpp:152) on explicit seeds; out[n] index-aligned with seeds[n]. */
int bella_hip_xdrop_batch(bella_ctx* ctx, const bella_seed* seeds, uint64_t n, const bella_params* p,
                          bella_aln* out);

/* ---- output writer (overlap.hpp:531-590 formatting, :603-642 per-thread buffers + offset writes) -------------------------
 * Formats pairs[npairs] (in the reference's order, as bella_hip_get_pairs delivers them) -- 6-column lines when
 * p->skip_alignment (overlap.hpp:577-588), else the passed alignments as BELLA's 12 columns (:472-473) or PAF (paf != 0,
 * :476-489) -- and APPENDS the text to `path` (the reference opens its file in append mode, :613).  names[nreads]: NUL-terminated
 * read names (readType_::nametag); lens[nreads]: read lengths; alns may be NULL when p->skip_alignment.  nthreads host threads
 * (0 = min(hardware threads, 16): more threads into ONE file are slower, the writes serialise on the inode) first measure contiguous
 * shares (exact bytes, no formatting), then format pieces of about 1 MB into private buffers and pwrite() each at its offset behind
 * the file's size at entry (fstat, not O_APPEND: ONE writer per file at a time is assumed; mapping the file was measured and
 * rejected, page faults).  Plain host code: no context, usable from any thread.  stats (nullable): what RunPairWiseAlignments
 * returns (:644) + timings. */
typedef struct {
    uint64_t lines;            /* outputted                                                       */
    uint64_t bytes;
    uint64_t aligned_pairs;    /* numAlignmentsThread (overlap.hpp:544)                           */
    uint64_t aligned_bases;    /* sum of endV - begV (:569)                                       */
    uint64_t total_read_len;   /* sum of both read lengths (:545)                                 */
    uint64_t bases_passed;     /* numBasesAlignedTrue (:491)                                      */
    uint64_t bases_failed;     /* numBasesAlignedFalse (:495)                                     */
    double seconds;            /* whole call                                                       */
    double format_seconds;     /* of which: validation + measuring pass (before the file is touched) */
    uint32_t threads;
    uint32_t pad;
} bella_write_stats;
int bella_hip_write_output(const char* path, const bella_params* p, int paf, uint32_t nreads, const char* const* names, const uint32_t* lens,
                           const bella_pair* pairs, const bella_aln* alns, uint64_t npairs, int nthreads, bella_write_stats* stats);

/* The exact (growing band) gapped X-drop of the reference's CUDA build instead of Xavier's 32-cell adaptive band: the scores and
 * seed positions of loganGPU/functions.cuh:223-408,505-547,680-682 (= SeqAn's extendSeed(GappedXDrop), include/align.hpp:93-139)
 * and the pass test of PostAlignDecisionGPU (include/overlap.hpp:797-871).  Same inputs and outputs as the two calls above;
 * bella_aln::steps = anti-diagonals computed, flagged = 0.  Extension scores live in int16 rings as in the CUDA kernel (`short`
 * anti-diagonals, loganGPU/functions.cuh:236-240): a one-direction score above 32,767 wraps there and here. */
int bella_hip_align_pairs_exact(bella_ctx* ctx, const bella_params* p, uint64_t* npassed);
int bella_hip_xdrop_batch_exact(bella_ctx* ctx, const bella_seed* seeds, uint64_t n, const bella_params* p, bella_aln* out);

/* ---- multi-GPU: one context per GPU, RCCL over xGMI ------------------------------------------------
 * The reference's multi-GPU path hands alignment batches to the devices inside one call (loganGPU/functions.cuh:441-443,
 * 498-637; include/align.hpp:226-229) and has no collective.  Here reads are 1D row-block partitioned: context r assembles the
 * rows of B of its read block (bella_hip_assemble_panel / _counted_panel), bella_hip_allgather_panels exchanges the panels with
 * ONE grouped point-to-point all-gather (every peer pair uses its own xGMI link; the blocks land directly at their offsets of
 * the full arrays, no padding, no staging) and builds the device layout; bella_hip_set_partition then gives every context its
 * output columns.  The communicator is RCCL's: rank 0 makes the 128-byte id (bella_hip_comm_id), the host program hands it to
 * every rank (MPI, torch.distributed, a file ...), every rank calls bella_hip_comm_init (collective). */
#define BELLA_HIP_COMM_ID_BYTES 128
int bella_hip_comm_id(uint8_t id[BELLA_HIP_COMM_ID_BYTES]);
int bella_hip_comm_init(bella_ctx* ctx, int nranks, int rank, const uint8_t id[BELLA_HIP_COMM_ID_BYTES]);
int bella_hip_comm_destroy(bella_ctx* ctx);
/* 1 if librccl can be loaded in this process (every rank can check this BEFORE the collective bella_hip_comm_init) */
int bella_hip_comm_available(void);
/* The same communicator interface over an in-process transport: the "ranks" are contexts of ONE process (each driven by its own
 * host thread, on one GPU or several), rendezvous on the host, bytes moved with device-to-device copies.  Runs the N > 1 logic of
 * the two collective entry points below without RCCL (tests on a single GPU; several contexts inside one host program). */
int bella_hip_comm_id_local(uint8_t id[BELLA_HIP_COMM_ID_BYTES]);
int bella_hip_comm_init_local(bella_ctx* ctx, int nranks, int rank, const uint8_t id[BELLA_HIP_COMM_ID_BYTES]);
/* collective: needs a panel (rank r: rows [first_r, first_r + rows_r), the blocks in rank order covering all reads) */
int bella_hip_allgather_panels(bella_ctx* ctx);
/* collective k-mer counting (kmercount.hpp:467-677 + main.cpp:393-416 across the ranks; the reference's relative is --split-count,
 * kmercount.hpp:534, which partitions the k-mer space into sequential passes): rank r counts the canonical k-mers of ITS range of
 * the code space over all reads, the partial dictionaries are exchanged once, tuples are made for the reads of the rank's own
 * block only (the block it will pass to bella_hip_assemble_counted_panel).  selector: 0 all k-mers, 1 syncmers (-s), 2 minimizers (-w). */
int bella_hip_count_kmers_dist(bella_ctx* ctx, uint16_t kmer_size, uint32_t lower, uint32_t upper, uint32_t selector, uint32_t window,
                               uint32_t first_read, uint32_t nreads_block, uint32_t* nkmers, uint64_t* ntuples, uint64_t* ndistinct);

/* ---- measurement --------------------------------------------------------------------------------- */
int bella_hip_get_timings(bella_ctx* ctx, bella_timings* t);
/* Device memory the context holds right now, by role (multi-GPU: what is replicated and what follows the partition). */
typedef struct {
    uint64_t reads_bytes;      /* packed reads + offsets: replicated                                                     */
    uint64_t matrix_bytes;     /* B in the reference's layout (colptr / rowids / values): what the all-gather delivers     */
    uint64_t layout_A_bytes;   /* A' (k-mer -> reads lists): whole on every context                                        */
    uint64_t layout_B_bytes;   /* B' entries + the rows' product counts + row pointers: owned columns only                 */
    uint64_t rowlist_bytes;    /* row lists (BELLA_TUNE_ROW_LISTS) + row pointers: owned columns only                      */
    uint64_t pass_bytes;       /* buffers of the passes (records, product lists, workspaces): follow the pass's products   */
    uint64_t other_bytes;      /* counting / assembly / alignment buffers still held, and released ones kept for reuse      */
    uint64_t owned_nnz;        /* nnz of the owned columns (B' keeps those of them that have a later read: BELLA_TUNE_COMPACT_B) */
    uint64_t layout_shared;    /* 1: the current layout was formed shared over the ranks (BELLA_TUNE_DIST_LAYOUT), else 0  */
    uint64_t live_nnz;         /* B' entries the layout holds: those of the owned columns that have a later read (ABI 6)  */
} bella_memory;
int bella_hip_get_memory(bella_ctx* ctx, bella_memory* m);
/* the same, writing at most struct_size bytes (pass sizeof(bella_memory) of the header the caller was built against: the struct has
 * grown between ABI versions -- 64 bytes in version 4, 72 in 5, 80 in 6 -- and will again) */
int bella_hip_get_memory_sized(bella_ctx* ctx, void* m, uint64_t struct_size);
/* 0 = default; bit0 = force the global-memory row path (tests); bit1 = no pair_ext output; bit2 = tests: treat every fifth
 * column as if its product lists had come out of order (the LDS tiers verify the order and fall back to the repairing path);
 * bit3 = tests: 512-thread workgroups in every LDS class (default: 1024 threads where a CU holds one or two columns);
 * bit4 = tests: key tables of cap/2 slots (the layout of pair-rich inputs) on any input; bit5 = tests: every column above the
 * LDS tiers takes the sort-based path of the wide columns (default: from 16 such columns in a pass on); bit6 = tests: that path
 * sorts on 64-bit keys on any input (default: 32-bit keys when column bits + read-id bits fit); bit8 = tests: the exact X-drop mode
 * launches its extensions in chunks of 1,000 (default 2^24: grid x block stays below 2^32 threads); bit10 (read when the operands are
 * assembled) = tests: the lists of A' in order of first appearance in B' (default: k-mer order), no row lists; bit11 (same moment) =
 * tests: as if the row lists BELLA_TUNE_ROW_LISTS asks for did not fit in memory (the layout stands without them);
 * bit12 = tests: the columns above the LDS tiers are grouped by the radix sort also when the row lists would allow grouping in LDS;
 * bit13 = tests: the symbolic phase (bella_hip_count_pairs) keeps its bitmaps in global memory also when they fit in LDS;
 * bit14 = tests: k-mer counting looks every position up in a hash table over the dictionary (the path of syncmer mode, of k-mers too
 * long to share a 64-bit sort key with their position, and of the distributed count) also where the sorted words carry their positions;
 * bit15 (read when the operands are assembled) = tests: every entry of B' in the plain form (default, when A' is larger than the last-level
 * cache: an entry whose k-mer has exactly one later read carries that read instead of an index into A'; the plain form is also that of
 * inputs with 2^30 reads or 2^31 nonzeros and more); bit16 (same moment) = tests: that inline form on inputs of any size;
 * bit18 = tests: inside bella_hip_allgather_panels this rank fails after the ranks agreed on the shared formation of A' and before its own
 * share begins (every rank must leave the call with an error, none may wait) */
int bella_hip_set_debug(bella_ctx* ctx, uint32_t flags);
/* Reserve device memory up front: ONE slab of `bytes` taken from the driver (and touched) now, from which the stages' buffers are cut
 * afterwards.  The reference has no counterpart (its vectors grow on the host); here the first hipMalloc of a multi-GB buffer costs
 * tens of ms per GB (the driver maps and wipes the pages), which a process that runs the pipeline should pay once, not inside its first
 * k-mer count, assembly and pass.  *ms (optional) = what the reservation took.  bytes == 0 gives an unused slab back.  What does not fit
 * the slab later is allocated as before; the slab goes with the context. */
int bella_hip_reserve(bella_ctx* ctx, uint64_t bytes, double* ms);
/* Device buffers of >= 128 MB that a stage released stay with the context for the next stage that fits them (a hipMalloc right after a
 * hipFree of tens of GB waits for the driver to wipe the pages); they are counted as free wherever the library sizes something by free
 * memory and are given up when one of its own allocations fails.  bella_hip_trim gives them back to the driver NOW: for hosts whose
 * other allocators (another context, RCCL, a framework) need the memory. */
int bella_hip_trim(bella_ctx* ctx);
/* Per-context tuning parameters (tests and A/B measurements; nothing here changes results).  what:
 *   BELLA_TUNE_LDS_TIERS      values = ascending product capacities of the row kernels' LDS tiers, each in [64, 11008] (n = 0: defaults)
 *   BELLA_TUNE_KCOUNT_BUDGET  values[0] = k-mers per pass of the counting sort (default 2^30)
 *   BELLA_TUNE_WIDE_BUDGET    values[0] = products per batch of the sort-based path of the wide columns (default 2^30)
 *   BELLA_TUNE_XDROP_VARIANT  values[0] = 0 one launch in length-sorted order, 1 (default; n = 0) slices of 512 steps with compaction of
 *                             the live extensions between launches, 2 packed kernel in pair order, 3 the scalar statement of xavier.h.
 *                             Same results in every variant.
 *   BELLA_TUNE_ROW_LISTS      values[0] = 1: operands installed from now on also get ROW LISTS -- the two operand entries of every product
 *                             of the owned columns side by side in product order (10 bytes per product, written once at assembly time)
 *                             -- when they fit next to what a pass needs; the numeric phase then streams its products instead of
 *                             expanding B' x A'.  For callers that run SEVERAL passes over the same columns (parameter sweeps): the
 *                             expansion is paid once instead of per pass.  Default 0 (n = 0): a one-shot call is faster without them
 *                             (layout + first pass); bella_timings.expand_ms reports the expansion when they are built.
 *   BELLA_TUNE_XDROP_CLASS_MIN values[0] = extensions of a batch from which on the slices of variant 1 run the batch as four classes by step
 *                             estimate, the three classes of long extensions on streams of their own at a higher priority (default 2^20:
 *                             four times the wavefronts the device holds at once; tests set it low, UINT64_MAX = never)
 *   BELLA_TUNE_LAYOUT_ORDER   values[0] = 0 (default): the lists of A' in k-mer order; 1: in order of first appearance in B' (the owner row of a
 *                             list streams it; three more random-access passes at layout time).  Read when operands are installed.
 *   BELLA_TUNE_INLINE_ENTRIES values[0] = 0 (default): an entry of B' whose k-mer has exactly one later read carries that read (util.hpp)
 *                             when A' is larger than BELLA_TUNE_CACHE_BYTES; 1: never (plain entries); 2: always.  Read at layout time.
 *   BELLA_TUNE_ROW_PATH       values[0] = 0 (default): columns in the LDS tiers; 1: every column on the global-workspace path (the
 *                             repairing path a device that fails the lane-order self-test gets)
 *   BELLA_TUNE_CACHE_BYTES    values[0] = size of A' from which on it counts as larger than the last-level cache (default 192 MB: three
 *                             quarters of the 256 MB Infinity Cache of an MI355X; the HIP runtime does not report that cache)
 *   BELLA_TUNE_DIST_LAYOUT    values[0] = 0: bella_hip_allgather_panels forms A' on every rank (each sorts ALL entries by k-mer).  1: the
 *                             formation is SHARED over the ranks of the communicator (2 = default: shared on the in-process transport, where
 *                             it is measured; replicated over RCCL until a run on several devices has shown the win) -- rank g sorts the entries whose k-mer id lies in the
 *                             g-th N-th of the id space, emits its slice of A' and the B' entries of all rows for those k-mers; slices and
 *                             entries travel in one more grouped exchange.  The same layout entry for entry.  Taken when EVERY rank can: the
 *                             partition (first, stride) = (rank, ranks) set by bella_hip_set_partition before the call, at most 64 ranks,
 *                             BELLA_TUNE_LAYOUT_ORDER 0 (the call asks all ranks and falls back to the replicated formation on all of them).
 *   BELLA_TUNE_COMPACT_B      values[0] = 0 (default): the device layout drops the entries of B' whose k-mer has no later read (they have
 *                             no product in the strict lower triangle, overlap.hpp:157-202); 1: every entry stays.  Read at layout time.
 * (bella_hip_set_debug bits 0, 10, 15 and 16 of earlier rounds still select the same things: aliases, no longer needed)
 */
enum { BELLA_TUNE_LDS_TIERS = 0, BELLA_TUNE_KCOUNT_BUDGET = 1, BELLA_TUNE_WIDE_BUDGET = 2, BELLA_TUNE_XDROP_VARIANT = 3, BELLA_TUNE_ROW_LISTS = 4,
       BELLA_TUNE_XDROP_CLASS_MIN = 5, BELLA_TUNE_LAYOUT_ORDER = 6, BELLA_TUNE_INLINE_ENTRIES = 7, BELLA_TUNE_ROW_PATH = 8, BELLA_TUNE_CACHE_BYTES = 9, BELLA_TUNE_DIST_LAYOUT = 10, BELLA_TUNE_COMPACT_B = 11 };
int bella_hip_set_tuning(bella_ctx* ctx, uint32_t what, const uint64_t* values, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif /* BELLA_HIP_H */
