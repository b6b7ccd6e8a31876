/* gasm.h — C ABI of libgasm, the MI355X (gfx950) implementation of the de Bruijn graph + k-meric breakage scoring
 * hot path of SahakyanLab/GenomeAssembler_dev.
 *
 * This header is the drop-in boundary.  The reference crosses exactly one FFI: R -> C++ through Rcpp::sourceCpp
 * (`// [[Rcpp::export]]` in lib/DeNovoAssembler.cpp:85,214,316 and lib/BreakageScorer.cpp:79,185).  Every entry
 * point below names the exported reference function it replaces; integration/DeNovoAssemblerHIP.cpp (shown in
 * INTEGRATION.md) is the Rcpp glue a maintainer would source instead of the reference file.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ or torch types cross this boundary;
 *   - every function returns GASM_OK (0) or a negative gasm_status; gasm_last_error() (thread-local) has the text;
 *     no exception ever leaves the library (the reference's C++ exceptions become R errors through Rcpp's
 *     BEGIN_RCPP/END_RCPP; the glue re-raises a non-zero status as Rcpp::stop);
 *   - string lists are one byte buffer + n+1 uint64 offsets (string i = data[off[i] .. off[i+1]));
 *   - results are library-allocated objects read through accessors and released with their *_free;
 *   - bases are upper-case ACGT.  2-bit packing cannot hold anything else: other bytes return GASM_ERR_NON_ACGT
 *     (documented divergence: the reference would silently create new hash keys);
 *   - a gasm_ctx owns one GPU (device ordinal given at creation), one HIP stream and its scratch memory; calls on one
 *     ctx must not overlap in time, different ctxs are independent.  There is no CPU fallback: without a usable
 *     gfx950 device gasm_ctx_create fails with GASM_ERR_NO_DEVICE.
 */
#ifndef GASM_H
#define GASM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum gasm_status {
    GASM_OK = 0,
    GASM_ERR_INVALID = -1,    /* bad argument (null pointer, k out of range, inconsistent sizes) */
    GASM_ERR_NON_ACGT = -2,   /* a base outside upper-case ACGT */
    GASM_ERR_NO_DEVICE = -3,  /* no usable gfx950 GPU / HIP runtime error at start-up */
    GASM_ERR_HIP = -4,        /* HIP runtime error during a call */
    GASM_ERR_CAPACITY = -5,   /* input exceeds a documented limit */
    GASM_ERR_RANGE = -6,      /* the reference would throw std::out_of_range here (substr past the end) */
    GASM_ERR_STATE = -7       /* call order violated (e.g. score before build) */
} gasm_status;

#define GASM_MAX_K 63          /* k-mer keys are 64-bit for k <= 31 and 128-bit for 32 <= k <= 63 */
#define GASM_TABLE_ROWS 69904  /* 4^2 + 4^4 + 4^6 + 4^8 rows of the breakage table, in that order, each lexicographic */

typedef struct gasm_ctx gasm_ctx;
typedef struct gasm_contigs gasm_contigs;
typedef struct gasm_strlist gasm_strlist;
typedef struct gasm_scores gasm_scores;
typedef struct gasm_batch gasm_batch;

const char* gasm_last_error(void);
const char* gasm_version(void);

int gasm_ctx_create(int device, gasm_ctx** out);
void gasm_ctx_destroy(gasm_ctx* ctx);
/* waits for everything queued on the ctx stream */
int gasm_ctx_sync(gasm_ctx* ctx);
/* the ctx's hipStream_t as an opaque pointer (for callers that time with their own HIP events) */
void* gasm_ctx_stream(gasm_ctx* ctx);

/* ------------------------------------------------------------------------------------------------------------------
 * get_contigs(read_kmers, dbg_kmer, seed)                       replaces lib/DeNovoAssembler.cpp:86-206
 *   kmers/n_kmers : the exploded k-mers, each exactly dbg_kmer characters, concatenated (what
 *                   lib/DeNovoAssembler.R:109-130 builds); duplicates allowed, order irrelevant.
 *   matrix_rows   : the reference's baked-in 10 000 (lib/DeNovoAssembler.cpp:195); 0 skips the shuffle.
 * Result: the sorted unique contigs (the value of `contigs` after :192) and the shuffle matrix of :195-203 as
 * matrix_rows x count indices into them (row-major).  The shuffle is std::shuffle with std::mt19937(seed) on the
 * host, exactly as the reference draws it.
 * Graph by-products kept for parity checks: the distinct k-mers (sorted, = the distinct edge list) with their
 * multiplicities, 2-bit packed big-endian in `words` 64-bit words per k-mer (1 for k<=31, else 2; base 0 is the most
 * significant pair of the k-mer's 2k bits).
 * ---------------------------------------------------------------------------------------------------------------- */
int gasm_get_contigs(gasm_ctx* ctx, const char* kmers, uint64_t n_kmers, int dbg_kmer, int seed, int matrix_rows,
                     gasm_contigs** out);
/* The same from the reads themselves — get_kmers_from_reads (lib/DeNovoAssembler.R:109-130: substring() of every read at every
 * offset, 31 characters per k-mer handed across the R / C++ boundary) and get_contigs in one call: the k-mers are taken on the
 * GPU from the 2-bit packed reads.  reads: the reads' characters back to back, read_off[n_reads + 1] their offsets; a read
 * shorter than dbg_kmer has no k-mers.  Same result as gasm_get_contigs on the exploded k-mers (the k-mers' order never
 * mattered: lib/DeNovoAssembler.cpp:91-122 counts them). */
int gasm_get_contigs_from_reads(gasm_ctx* ctx, const char* reads, const uint64_t* read_off, uint64_t n_reads, int dbg_kmer, int seed,
                                int matrix_rows, gasm_contigs** out);
uint64_t gasm_contigs_count(const gasm_contigs* c);
const char* gasm_contigs_data(const gasm_contigs* c);
const uint64_t* gasm_contigs_offsets(const gasm_contigs* c);       /* count+1 */
uint64_t gasm_contigs_rows(const gasm_contigs* c);
const uint32_t* gasm_contigs_perm(const gasm_contigs* c);          /* rows*count */
uint64_t gasm_contigs_distinct_count(const gasm_contigs* c);
int gasm_contigs_key_words(const gasm_contigs* c);
const uint64_t* gasm_contigs_distinct_keys(const gasm_contigs* c); /* distinct_count*words */
const uint32_t* gasm_contigs_distinct_mult(const gasm_contigs* c); /* distinct_count */
void gasm_contigs_free(gasm_contigs* c);

/* ------------------------------------------------------------------------------------------------------------------
 * assemble_contigs(contig_matrix, dbg_kmer)                      replaces lib/DeNovoAssembler.cpp:215-305
 *   the matrix is `rows` rows of `row_len` entries (perm, row-major indices into the `n` distinct strings); for
 *   get_contigs' output every row is a permutation of all n contigs, so row_len == n.
 * assemble_contigs(velvet_contigs, dbg_kmer, seed)               replaces lib/BreakageScorer.cpp:80-174
 *   rows = the reference's baked-in 20 000 (lib/BreakageScorer.cpp:86).
 * Result: distinct scaffolds, longest first (same std::sort call as the reference).  GASM_ERR_RANGE where the
 * reference's substr would throw (a contig shorter than the overlap being tried).
 * ---------------------------------------------------------------------------------------------------------------- */
int gasm_assemble_contigs(gasm_ctx* ctx, const char* contigs, const uint64_t* off, uint64_t n, const uint32_t* perm,
                          uint64_t rows, uint64_t row_len, int dbg_kmer, gasm_strlist** out);
int gasm_assemble_contigs_velvet(gasm_ctx* ctx, const char* contigs, const uint64_t* off, uint64_t n, int dbg_kmer,
                                 int seed, int rows, gasm_strlist** out);
/* The same two functions with the scaffolds left on the GPU (2-bit, the reference's order: longest first) behind a handle
 * that gasm_calc_breakscore_dev takes as its `path` argument: the text of the scaffolds (3.8e8 characters for one 50 kb
 * experiment) is only made when gasm_scaffolds_fetch asks for it.  Same results as the string forms. */
typedef struct gasm_scaffolds gasm_scaffolds;
int gasm_assemble_contigs_dev(gasm_ctx* ctx, const char* contigs, const uint64_t* off, uint64_t n, const uint32_t* perm,
                              uint64_t rows, uint64_t row_len, int dbg_kmer, gasm_scaffolds** out);
int gasm_assemble_contigs_velvet_dev(gasm_ctx* ctx, const char* contigs, const uint64_t* off, uint64_t n, int dbg_kmer, int seed,
                                     int rows, gasm_scaffolds** out);
uint64_t gasm_scaffolds_count(const gasm_scaffolds* s);
/* who ran the greedy merge behind these scaffolds: 1 the GPU (k_asm_merge), 2 host threads, 3 both (*rows_on_host of the
 * permutations went back to the host routine: two chains of equal length need the reference's full-string test) */
int gasm_scaffolds_merge_device(const gasm_scaffolds* s, uint64_t* rows_on_host);
const uint64_t* gasm_scaffolds_offsets(const gasm_scaffolds* s);   /* count + 1 base offsets: lengths without a fetch */
int gasm_scaffolds_fetch(const gasm_scaffolds* s, gasm_strlist** out);
void gasm_scaffolds_free(gasm_scaffolds* s);
uint64_t gasm_strlist_count(const gasm_strlist* s);
const char* gasm_strlist_data(const gasm_strlist* s);
const uint64_t* gasm_strlist_offsets(const gasm_strlist* s);
void gasm_strlist_free(gasm_strlist* s);

/* ------------------------------------------------------------------------------------------------------------------
 * calc_breakscore(path, sequencing_reads, true_solution, kmer, bp_kmer, bp_prob)
 *   variant GASM_SCORE_OWN    replaces lib/DeNovoAssembler.cpp:317-477  (returns path_freq)
 *   variant GASM_SCORE_VELVET replaces lib/BreakageScorer.cpp:186-353   (returns path_prob_dist(+_startpos),
 *                                                                         Levenshtein in infix mode)
 * flags: GASM_WANT_LEV computes lev_dist_vs_true (else zeros) — global distance for the own variant, infix for the velvet
 *        one, as the reference's two calc_levenshtein do; on the GPU when that is the quicker place (many or long paths),
 *        on the host for a handful of short ones or a true_solution with bytes outside ACGT (same numbers either way;
 *        GASM_LEV_GPU / GASM_LEV_HOST in the environment force one or the other); GASM_WANT_FREQ materialises the dense path_freq
 * (n_paths x n_table doubles, in bp_kmer order — the reference emits them in hash-iteration order, so only the
 * multiset per path is defined there).
 * bp_kmer keys must be distinct ACGT strings of length 1..8 (the reference tables hold lengths 2,4,6,8).
 * Scores are FP64, summed in a fixed order (deterministic run to run); the reference sums in hash-iteration order,
 * hence the 1e-9 absolute tolerance of the parity tests.
 * ---------------------------------------------------------------------------------------------------------------- */
#define GASM_SCORE_OWN 0
#define GASM_SCORE_VELVET 1
#define GASM_WANT_LEV 1
#define GASM_WANT_FREQ 2
#define GASM_WANT_KS 4       /* stat_test_KS per path: lib/DeNovoAssembler.R:414-424 (own variant; true_solution must be ACGT) */
int gasm_calc_breakscore(gasm_ctx* ctx, const char* paths, const uint64_t* path_off, uint64_t n_paths,
                         const char* reads, const uint64_t* read_off, uint64_t n_reads, const char* true_solution,
                         uint64_t true_len, int kmer, const char* bp_kmer, const uint64_t* bp_off, uint64_t n_table,
                         const double* bp_prob, int variant, int flags, gasm_scores** out);
/* calc_breakscore of device-resident scaffolds (gasm_assemble_contigs_dev); everything else as gasm_calc_breakscore */
int gasm_calc_breakscore_dev(gasm_ctx* ctx, const gasm_scaffolds* paths, const char* reads, const uint64_t* read_off, uint64_t n_reads,
                             const char* true_solution, uint64_t true_len, int kmer, const char* bp_kmer, const uint64_t* bp_off,
                             uint64_t n_table, const double* bp_prob, int variant, int flags, gasm_scores** out);
uint64_t gasm_scores_count(const gasm_scores* s);
const int32_t* gasm_scores_sequence_len(const gasm_scores* s);
const double* gasm_scores_bp_score(const gasm_scores* s);
const double* gasm_scores_norm_by_break_freqs(const gasm_scores* s);
const double* gasm_scores_norm_by_len(const gasm_scores* s);
const int32_t* gasm_scores_kmer_breaks(const gasm_scores* s);
const int32_t* gasm_scores_lev_dist(const gasm_scores* s);
const double* gasm_scores_path_freq(const gasm_scores* s);          /* count*n_table or NULL */
const int32_t* gasm_scores_startpos(const gasm_scores* s);          /* velvet variant, else NULL */
const double* gasm_scores_prob_dist(const gasm_scores* s);          /* velvet: concatenated, see offsets */
const uint64_t* gasm_scores_prob_dist_offsets(const gasm_scores* s);/* count+1 */
/* two-sample Kolmogorov-Smirnov statistic D of the path's path_freq (NaN entries dropped) against the probabilities of
 * the true solution's kmer-long windows (kmer_from_seq, lib/GenerateReads.R:243-259): what ks.test(...)$statistic gives in
 * lib/DeNovoAssembler.R:419-424; NaN where no read matched (R stops with an error there).  NULL without GASM_WANT_KS. */
const double* gasm_scores_ks(const gasm_scores* s);
/* who computed lev_dist_vs_true: 0 nobody (not asked for), 1 the GPU (k_levenshtein), 2 host threads (a cost model prefers them
 * for a handful of short paths; targets with bytes outside ACGT; GASM_LEV_HOST) — same numbers either way */
int gasm_scores_lev_device(const gasm_scores* s);
void gasm_scores_free(gasm_scores* s);

/* contig_frac_len of lib/DeNovoAssembler.R:432-445: percentage of [1, seq_len] covered by the union of the inclusive
 * ranges [start_i, start_i + len_i] (GRanges reduce + setdiff). */
int gasm_coverage_percent(gasm_ctx* ctx, const int64_t* start, const int64_t* len, uint64_t n, int64_t seq_len, double* percent);

/* Levenshtein distance as the reference takes it from edlib (lib/DeNovoAssembler.cpp:41-55 global,
 * lib/BreakageScorer.cpp:41-55 infix); 0 for an empty operand, like the reference's failure branch. */
int gasm_levenshtein(const char* query, uint64_t nq, const char* target, uint64_t nt, int infix, int32_t* out);

/* ------------------------------------------------------------------------------------------------------------------
 * Segment batches: the reads-in / contigs+scores-out surface for many independent segments at once (the reference
 * loops `for(i in 1:total_iters)` over independent segments, scripts/02_Real_vs_rand_prob_own.R:33-53).  No
 * reference function takes reads directly: k-mer extraction is R code (lib/DeNovoAssembler.R:109-130); here it is
 * the first kernel.  All data stays in HBM between the three calls; only gasm_batch_fetch_* copies back.
 *
 *   reads        : all reads of all segments, concatenated ASCII, segment after segment
 *   read_off     : n_reads+1 offsets, or NULL when every read has fixed_len bases
 *   seg_read_off : n_segments+1 indices into the read list
 * gasm_batch_build(k)  : k-mers -> distinct k-mers + multiplicities -> (k-1)-mer graph -> contigs, per segment
 *                        (get_contigs without the shuffle; results identical to calling it per segment)
 * gasm_batch_score     : calc_breakscore (own variant, without Levenshtein/path_freq) of every segment's contigs
 *                        against that segment's reads; table = GASM_TABLE_ROWS normalised probabilities
 * Both queue work on the ctx stream and return; gasm_ctx_sync (or any fetch) waits.
 * ---------------------------------------------------------------------------------------------------------------- */
int gasm_batch_create(gasm_ctx* ctx, const char* reads, const uint64_t* read_off, uint64_t n_reads, uint32_t fixed_len,
                      const uint64_t* seg_read_off, uint32_t n_segments, gasm_batch** out);
/* the same from reads that are 2-bit packed already: A=0 C=1 G=2 T=3, first base in the two most significant bits of words[0],
 * 32 bases per word, reads back to back (read i = bases [read_off[i], read_off[i+1]), or i * fixed_len); a quarter of the
 * bytes over PCIe and no packing kernel */
int gasm_batch_create_packed(gasm_ctx* ctx, const uint64_t* words, const uint64_t* read_off, uint64_t n_reads, uint32_t fixed_len,
                             const uint64_t* seg_read_off, uint32_t n_segments, gasm_batch** out);
/* FASTQ / FASTA in: one file per segment (FASTQ with four-line records, FASTA with one- or multi-line sequences, plain or
 * gzip; lower case folded to upper case).  The reference has no reader of its own on this path (reads are simulated in R
 * and written as FASTA, lib/GenerateReads.R:405-433).  on_non_acgt: 0 = reads holding a byte outside ACGT are dropped and
 * counted in *dropped_reads, 1 = GASM_ERR_NON_ACGT.  The host packs 2-bit while it parses. */
int gasm_batch_from_files(gasm_ctx* ctx, const char* const* paths, uint32_t n_files, int on_non_acgt, gasm_batch** out,
                          uint64_t* dropped_reads);
/* Simulated reads, made on the device (lib/GenerateReads.R:235-313): per segment ceil(coverage * L / read_len) start
 * positions drawn with replacement, the weight of a position = table probability of the kmer-long window starting there
 * (table = the GASM_TABLE_ROWS normalised probabilities, kmer in {2,4,6,8}; table = NULL: every one of the L - kmer + 1
 * positions weighs the same), starts whose read would run past the end dropped, reads = substrings of the genome, forward
 * strand.  R's sample()/set.seed stream is not reproducible without R; the draws here are a documented counter-based
 * generator (kernels_sim.hip), identical for the same (seed, genomes) and restated by the oracle.
 * gasm_batch_fetch_read_starts: the 0-based start of every read in its genome (n_segments + 1 offsets, n_reads starts). */
int gasm_batch_simulate(gasm_ctx* ctx, const char* genomes, const uint64_t* genome_off, uint32_t n_segments, uint32_t read_len,
                        double coverage, uint64_t seed, int kmer, const double* table, gasm_batch** out);
int gasm_batch_fetch_read_starts(gasm_batch* b, const uint64_t** seg_read_off, const uint32_t** starts);
/* the reader alone (host only, no GPU needed): the reads of the files, packed as gasm_batch_create_packed takes them */
typedef struct gasm_packed gasm_packed;
int gasm_read_files(const char* const* paths, uint32_t n_files, int on_non_acgt, gasm_packed** out);
/* The same through the device path (csrc/ingest.hip): the host opens and inflates the file, the GPU finds the records (newline
 * scan), validates and 2-bit packs them; a text it does not recognise as plain four-line FASTQ or FASTA goes to the host reader,
 * whose grammar is the definition (gasm_packed_parsed_on_device(g, file) = 0 then).  Same results as gasm_read_files. */
int gasm_read_files_device(gasm_ctx* ctx, const char* const* paths, uint32_t n_files, int on_non_acgt, gasm_packed** out);
int gasm_packed_parsed_on_device(const gasm_packed* g, uint32_t file);
uint64_t gasm_packed_n_reads(const gasm_packed* g);
uint32_t gasm_packed_n_segments(const gasm_packed* g);
const uint64_t* gasm_packed_words(const gasm_packed* g);
const uint64_t* gasm_packed_read_off(const gasm_packed* g);        /* n_reads + 1 */
const uint64_t* gasm_packed_seg_read_off(const gasm_packed* g);    /* n_segments + 1 */
uint64_t gasm_packed_dropped(const gasm_packed* g);
void gasm_packed_free(gasm_packed* g);
void gasm_batch_free(gasm_batch* b);
/* genome_len_hint: expected distinct k-mers per segment (0 = derive from the k-mer count); only sizes buckets.
 * build and score only QUEUE their work (no host wait); the fetches below wait for it.  A batch that runs as one block
 * keeps up to four "step slots" — everything a step writes, on a stream of its own — and consecutive builds take them in
 * turn, so `build; score; build; score; ...` without a fetch in between runs step n + 1's streaming kernels beside step n's
 * graph and scoring kernels.  Every step still does all of its work, and every fetch returns the results of the LAST build
 * and score, bit for bit what one step at a time gives (GASM_PINGPONG=0 in the environment: exactly that; GASM_STEP_SLOTS:
 * 2..4 slots, default 3). */
int gasm_batch_build(gasm_batch* b, int k, uint64_t genome_len_hint);
int gasm_batch_score(gasm_batch* b, int kmer, const double* table);
uint64_t gasm_batch_total_kmers(const gasm_batch* b);   /* k-mers extracted by the last build */
uint64_t gasm_batch_total_reads(const gasm_batch* b);

/* results of the last build (host copies, valid until the next build/free) */
int gasm_batch_fetch_distinct(gasm_batch* b, const uint64_t** seg_off /*n_segments+1*/, const uint64_t** keys,
                              const uint32_t** mult, int* words);
/* The (k-1)-mer graph of the last build, one entry per distinct k-mer (= distinct edge) in the order of gasm_batch_fetch_distinct:
 * restates the degree table and the branching-node list of lib/DeNovoAssembler.cpp:125-169 and one step of its walk (:172-189).
 *   edge_flags bit 0: the edge's source node (its first k-1 bases) is a branching node (in-degree != 1 or out-degree != 1);
 *              bit 1: (on the first out-edge of a node, in sorted order) the node has two or more in-edges;
 *   edge_next: the edge the walk continues with (index into the batch's distinct list), 0xFFFFFFFF where a contig ends. */
int gasm_batch_fetch_graph(gasm_batch* b, const uint8_t** edge_flags, const uint32_t** edge_next);
int gasm_batch_fetch_contigs(gasm_batch* b, const uint64_t** seg_contig_off /*n_segments+1*/, const uint64_t** off,
                             const char** data);
/* results of the last score: one entry per contig, in the order of gasm_batch_fetch_contigs */
int gasm_batch_fetch_scores(gasm_batch* b, const double** bp_score, const double** norm_by_break_freqs,
                            const double** norm_by_len, const int32_t** kmer_breaks, const int32_t** sequence_len);

/* ------------------------------------------------------------------------------------------------------------------
 * Pooled builds over several GPUs: the reads of every segment are spread over the ranks, k-mers are bucketed by
 * (segment, first bits of the k-mer) and every bucket's records are brought together on one rank by an all-to-all before
 * the global edge-list merge (SURVEY.md §8(e) mode 2; the reference has no counterpart: it loops over segments on one
 * thread, scripts/02_Real_vs_rand_prob_own.R:33-53).  One gasm_pool per rank; the library does the device work of every
 * stage and packs / consumes device buffers, the caller moves them between ranks (RCCL through its own runtime —
 * genomeassembler_dev_amd/pooled.py uses torch.distributed — or plain copies between virtual ranks of one process) and
 * decides which rank owns which bucket and which segment.  Records travel as two streams: keys (8 bytes for k <= 31, 16
 * bytes hi:lo for k <= 63, sorted inside a run) and 32-bit counts; the bucket a run belongs to follows from its place in
 * the bucket lists both sides derive from the ownership function.  Results do not depend on the number of ranks.
 *
 *   gasm_pool_create       this rank's reads (fixed length) of ALL n_segments segments
 *   gasm_pool_local_runs   k-mers of those reads -> one sorted run of distinct (key, count) per bucket; bucket index =
 *                          segment << bbits | first bbits bits of the k-mer; run_len (host, n_segments << bbits entries)
 *   gasm_pool_pack_runs    the current runs of the listed buckets, back to back, into caller-owned device buffers
 *   gasm_pool_merge_runs   n_out output buckets, each the union (counts added) of up to n_src runs found at record offset
 *                          run_off[j * n_src + s], length run_len[j * n_src + s] (0 = none) of the input buffers; the
 *                          merged runs become the pool's current runs (bucket index = j); merged_len: host, n_out entries
 *   gasm_pool_graph        the current runs are the buckets of n_local segments (n_local << bbits, segment-major):
 *                          graph, traversal, contigs of those segments (as gasm_batch_build)
 *   gasm_pool_piece_words / gasm_pool_pack_reads   the 2-bit reads of segments [seg_lo, seg_hi) of this rank as
 *                          word-aligned pieces, one per segment, back to back
 *   gasm_pool_set_reads    the reads of the rank's own segments as received: piece i holds piece_reads[i] reads of local
 *                          segment piece_seg[i] (non-decreasing) from word piece_word_off[i] of d_words
 *   gasm_pool_score        as gasm_batch_score, for the rank's own segments
 *   gasm_pool_fetch_*      as gasm_batch_fetch_*, for the rank's own segments
 * GASM_ERR_CAPACITY from local_runs / merge_runs: a bucket holds too many distinct k-mers — use more bucket bits (all
 * ranks must use the same bbits).
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct gasm_pool gasm_pool;
int gasm_pool_create(gasm_ctx* ctx, const char* reads, uint64_t n_reads, uint32_t fixed_len, const uint64_t* seg_read_off,
                     uint32_t n_segments, gasm_pool** out);
void gasm_pool_free(gasm_pool* p);
int gasm_pool_key_words(const gasm_pool* p);   /* 64-bit words per key: 1 (k <= 31) or 2; valid after gasm_pool_local_runs */
int gasm_pool_local_runs(gasm_pool* p, int k, int bbits, const uint32_t** run_len);
int gasm_pool_pack_runs(gasm_pool* p, const uint32_t* bucket_ix, uint64_t n, void* d_keys_out, void* d_counts_out);
int gasm_pool_merge_runs(gasm_pool* p, uint32_t n_out, uint32_t n_src, const uint64_t* run_off, const uint32_t* run_len,
                         const void* d_keys_in, const void* d_counts_in, const uint32_t** merged_len);
int gasm_pool_graph(gasm_pool* p, uint32_t n_local_segments);
int gasm_pool_piece_words(gasm_pool* p, uint32_t seg_lo, uint32_t seg_hi, uint64_t* n_words /* seg_hi - seg_lo */);
int gasm_pool_pack_reads(gasm_pool* p, uint32_t seg_lo, uint32_t seg_hi, void* d_words_out);
int gasm_pool_set_reads(gasm_pool* p, const void* d_words, uint64_t n_words, uint32_t n_pieces, const uint32_t* piece_seg,
                        const uint64_t* piece_reads, const uint64_t* piece_word_off);
int gasm_pool_score(gasm_pool* p, int kmer, const double* table);
int gasm_pool_fetch_distinct(gasm_pool* p, const uint64_t** seg_off, const uint64_t** keys, const uint32_t** mult, int* words);
int gasm_pool_fetch_contigs(gasm_pool* p, const uint64_t** seg_contig_off, const uint64_t** off, const char** data);
int gasm_pool_fetch_scores(gasm_pool* p, const double** bp_score, const double** norm_by_break_freqs, const double** norm_by_len,
                           const int32_t** kmer_breaks, const int32_t** sequence_len);

/* ------------------------------------------------------------------------------------------------------------------
 * The pooled step with the exchanges inside the library (csrc/exchange.hip): what the north star calls "an RCCL all-to-all
 * over xGMI to bucket k-mers by hash before the global edge-list merge", as ONE call per step.  The reference analogue is
 * the sequential loop over segments, scripts/02_Real_vs_rand_prob_own.R:33-53.
 *
 *   gasm_comm_unique_id          128 bytes (ncclUniqueId) made by one rank and handed to all others by the caller's
 *                                bootstrap (a file, MPI, torch.distributed's store, ...): the only thing the host layer moves
 *   gasm_comm_create             ncclCommInitRank on the context's GPU: one rank per process, RCCL over xGMI
 *   gasm_comm_create_virtual     `world` ranks inside this process on one GPU; exchanges become device copies on the context's
 *                                stream — the same plans, kernels and directories, testable on a one-GPU box
 *   gasm_comm_stage              what the running gasm_pool_exchange_build is doing (10 local runs, 11/12/13 exchange 1: plan,
 *                                transfer, merge, 21/22/23 exchange 2, 31 reads, 32 scoring, 0 idle): for watchdogs
 *   gasm_pool_bucket_owner       owner rank of every bucket (index = segment << bbits | prefix)
 *   gasm_pool_segment_bounds     first[r] .. first[r + 1]: the segments rank r builds, scores and returns
 *   gasm_pool_exchange_build     local runs -> all-to-all #1 -> merge -> all-to-all #2 -> graph + contigs -> all-to-all #3
 *                                (reads) -> scoring (table != NULL).  pools: the caller's pool (RCCL) or one pool per
 *                                virtual rank, in rank order.  Run lengths, offsets and run directories stay on the device;
 *                                the host waits for two small reports per step (per-peer totals: ncclSend / ncclRecv take their
 *                                counts from the host).  Capacity failures are collective: every rank's overflow flag travels
 *                                with the length tables, all ranks take the same step of the retry ladder (exact partition,
 *                                larger tables, two more bucket bits) or return GASM_ERR_CAPACITY together.
 *                                stats (optional, 8 words): bytes the first local rank sent in exchange 1, 2, 3; of those to
 *                                other ranks; attempts; bucket bits used.  Results: gasm_pool_fetch_* of every pool.
 * ---------------------------------------------------------------------------------------------------------------- */
#define GASM_COMM_ID_BYTES 128
typedef struct gasm_comm gasm_comm;
int gasm_comm_unique_id(void* id /* GASM_COMM_ID_BYTES */);
int gasm_comm_create(gasm_ctx* ctx, const void* id, int rank, int world, gasm_comm** out);
int gasm_comm_create_virtual(gasm_ctx* ctx, int world, gasm_comm** out);
void gasm_comm_destroy(gasm_comm* c);
int gasm_comm_world(const gasm_comm* c);
int gasm_comm_rank(const gasm_comm* c);      /* -1: virtual */
int gasm_comm_stage(const gasm_comm* c);
int gasm_pool_bucket_owner(uint32_t n_segments, int bbits, uint32_t world, uint32_t* owner /* n_segments << bbits */);
int gasm_pool_segment_bounds(uint32_t n_segments, uint32_t world, uint32_t* first /* world + 1 */);
int gasm_pool_exchange_build(gasm_comm* c, gasm_pool* const* pools, uint32_t n_pools, int k, int bbits, int kmer,
                             const double* table /* 69 904 rows or NULL */, uint64_t* stats /* 8 words or NULL */);

/* ------------------------------------------------------------------------------------------------------------------
 * Breakage-score-guided traversal (BASELINE configs[4]'s "combined" mode).  NOT in the reference
 * (README.md:83 "out of scope"): specified by this project, parity = agreement with the project's own CPU
 * restatement (oracle/guided_oracle.py).  Contigs end at branching nodes; guided scaffolds chain them through those nodes:
 * seeds by descending breakage score per base (exact rational of the fixed-point sums), each extended to the right, then to
 * the left, by the best-scoring unused contig that overlaps by k-1 bases; every contig is used once.  Call after
 * gasm_batch_build + gasm_batch_score.  fetch: scaffolds per segment (longest first), their text and scores.
 * ---------------------------------------------------------------------------------------------------------------- */
int gasm_batch_guided(gasm_batch* b);
int gasm_batch_fetch_guided(gasm_batch* b, const uint64_t** seg_off /*n_segments+1*/, const uint64_t** off, const char** data,
                            const double** bp_score, const double** norm_by_len, const int32_t** kmer_breaks);
/* the exact fixed-point sums behind the batch scores: bp_score[c] == fx[c] * 2^-shift */
int gasm_batch_fetch_score_fixed(gasm_batch* b, const int64_t** fx, int* shift);

/* Per-kernel device time of the stages of build/score, accumulated with HIP events on the ctx stream since the last
 * reset (profiling on costs one event pair per launch).  names/ms/launches point into library storage. */
int gasm_profile_enable(gasm_ctx* ctx, int on);
/* restrict profiling to the comma-separated kernel names (NULL or "" = every kernel) */
int gasm_profile_filter(gasm_ctx* ctx, const char* names);
int gasm_profile_reset(gasm_ctx* ctx);
int gasm_profile_read(gasm_ctx* ctx, int* n, const char* const** names, const double** ms, const uint64_t** launches);

#ifdef __cplusplus
}
#endif
#endif /* GASM_H */
