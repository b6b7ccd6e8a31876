// pipeline.h — host-side state of the device pipeline (reads -> distinct k-mers -> graph -> contigs -> scores).
#pragma once
#include "gasm_internal.h"
#include "kernels.h"

// Reads of one or more segments, resident in HBM as one packed base stream.
struct DevReads {
    u32 n_segments = 0;
    u64 n_reads = 0, total_bases = 0;
    u32 fixed_len = 0;
    u32 min_len = 0, max_len = 0;          // over non-empty reads (0 if none)
    u64 n_empty = 0;
    std::vector<u64> h_read_off;            // ragged only
    std::vector<u64> h_seg_read_off;        // n_segments+1
    std::vector<u64> h_seg_empty;           // empty reads per segment
    bool positioned = false;                // fixed-length reads at the base positions in d_read_off (pooled builds)
    int sim_shift = 52;                     // fixed-point shift of the last weighted simulate()
    u64 upload_id = 0;                      // changes with every upload: "the same reads again?" (BuildState)
    DBuf d_words, d_read_off, d_seg_read_off;
    // tile directory cache (depends on reads per tile)
    u32 tiles_ipt = 0, tiles_orr = 0, n_tiles = 0;
    std::vector<u32> h_seg_tile_start, h_tile_info;
    DBuf d_seg_tile_start, d_tile_info;

    int upload(gasm_ctx* ctx, const char* reads, const u64* read_off, u64 n_reads, u32 fixed_len, const u64* seg_read_off,
               u32 n_segments);
    int upload_packed(gasm_ctx* ctx, const u64* words, const u64* read_off, u64 n_reads, u32 fixed_len, const u64* seg_read_off,
                      u32 n_segments);
    int adopt_packed(gasm_ctx* ctx, DBuf& words_dev, const u64* read_off, u64 n_reads, u32 fixed_len, const u64* seg_read_off, u32 n_segments);
    // reads simulated on the device from the genomes (lib/GenerateReads.R:235-313); d_kept_start receives every read's start
    int simulate(gasm_ctx* ctx, const char* genomes, const u64* genome_off, u32 n_segments, u32 read_len, double coverage, u64 seed, int kmer,
                 const double* table, DBuf& d_kept_start);
    int set_layout(const u64* read_off, u64 n_reads, u32 fixed_len, const u64* seg_read_off, u32 n_segments);
    int finish_upload(gasm_ctx* ctx);
    int set_tiles(gasm_ctx* ctx, u32 ipt, u32 orr);
    ReadSet view() const;
    u64 read_len(u64 r) const { return fixed_len ? fixed_len : h_read_off[r + 1] - h_read_off[r]; }
    void release();
};

// Paths to score (contigs produced by a build, or caller-supplied sequences).
struct DevPaths {
    u32 n_segments = 0, n_paths = 0;
    u64 total_bases = 0;
    std::vector<u64> h_p_off;               // n_paths+1
    std::vector<u32> h_seg_path_off;        // n_segments+1
    std::vector<u64> h_seg_base_off;        // n_segments+1: first base of each segment's paths
    DBuf d_words, d_p_off, d_seg_path_off, d_seg_base_off;
    // borrowed device directories (the contigs of a build): used instead of the owned buffers when set
    const u64* b_p_off = nullptr;
    const u32* b_seg_path_off = nullptr;
    const u64* b_seg_base_off = nullptr;
    const u64* seg_base_off_dev() const { return b_seg_base_off ? b_seg_base_off : d_seg_base_off.as<u64>(); }
    int pack_from_device_ascii(gasm_ctx* ctx, const u8* d_ascii);
    int upload_ascii(gasm_ctx* ctx, const char* data, const u64* off, u32 n_paths);
    int upload_dirs(gasm_ctx* ctx);
    PathSet view() const;
    void release();
};

struct BuildState {
    // ---- plan (host-side, from the reads): key width, tile shape, partition
    int k = 0, bbits = 0, fbits = 9, words = 1, bb_cap = 0;
    bool small_tbl = true;                  // 2048-slot de-duplication tables (else 4096)
    bool rank_global = false;               // list ranking by whole-GPU pointer doubling only (set after the LDS ranking gave up)
    bool single_pass = true;                // partition in one pass into regions of fixed capacity (k_bucket_partition); cleared when a region overflowed
    // the region layout in d_bstart belongs to ... (uploaded once per batch shape)
    bool part_valid = false;
    u64 part_reads_id = 0, part_alloc = 0;
    int part_k = 0, part_bbits = 0, part_slack = 0, part_forced = 0;
    u32 part_padm = 0, part_g = 0;
    bool multi_pass = false;                // de-duplication in passes over key sub-ranges (set after the bucket bits ran out: k_bucket_dedup_multi)
    bool ranked_in_lds = false;
    u32 tile_g = 1;                         // threads per read of the tile kernels
    u64 n_kmers = 0, hint = 0, reads_id = 0;
    std::vector<u64> h_seg_nk;              // k-mers per segment
    // upper bounds the arrays are allocated at, and estimates the grids are sized from (the kernels loop beyond them)
    u64 D_cap = 0, maxD_cap = 0, bases_cap = 0;
    u32 maxD_est = 1, paths_est = 0;
    bool have_actual = false;               // maxD_est / paths_est / the partition come from a finished build of the same reads
    // ---- optional stream choreography of sub-batches (capi.hip): wait for this event before the first kernel, record
    // that one once the streaming kernels (partition + de-duplication) are queued
    hipEvent_t ev_wait = nullptr, ev_streamed = nullptr;
    // the streaming kernels (partition, de-duplication, gather) on another context's stream (capi.hip: step slots), the
    // graph on the build's own; ev_slot / ev_dense order the two streams (created on first use, owned by this state)
    gasm_ctx* stream_ctx = nullptr;
    hipEvent_t ev_slot = nullptr, ev_dense = nullptr;
    hipEvent_t ev_before_dedup = nullptr;   // waited for between the partition and the de-duplication (the last step's scoring on its lane: capi.hip)
    // ---- report: written by the last kernels of a build into pinned memory, read by pipeline_build_finish
    u32* h_report = nullptr;
    size_t h_report_words = 0;
    u32 ticket = 0;
    bool pending = false;                   // a build is queued and its report has not been read
    // ---- results on the host (valid after pipeline_build_finish)
    u32 d_total = 0, n_contigs = 0;
    u64 contig_bases = 0;
    std::vector<u32> h_dstart;              // n_segments+1: first distinct k-mer of every segment
    std::vector<u32> h_seg_cstart;          // n_segments+1
    std::vector<u64> h_seg_bstart;          // n_segments+1
    DBuf d_keys2;                           // output of the multi-pass de-duplication (its passes re-read d_keys)
    DBuf d_keys, d_mult, d_hist, d_toff, d_tcnt, d_fdir, d_bstart, d_bucket_d, d_dstart, d_flags, d_rtab;
    DBuf d_dk_key, d_dk_cnt, d_eflag, d_nxt, d_link, d_clen, d_ecid, d_ecoff;
    DBuf d_seg_cbases, d_seg_cstart, d_seg_bstart, d_c_off, d_contig_ascii;
    // host copies filled by fetch
    std::vector<u64> h_seg_doff, h_dk_key, h_c_off, h_seg_coff;
    std::vector<u32> h_dk_cnt, h_nxt;
    std::vector<u8> h_eflag;
    std::vector<char> h_contigs;
    bool fetched_distinct = false, fetched_contigs = false;
    void release();
};

struct ScoreTable {
    // direct-address tables over ACGT strings of length 1..8 (87 380 rows)
    DBuf d_prob, d_row, d_fix;
    std::vector<double> h_prob;   // direct-address table as uploaded
    std::vector<double> h_row_prob;   // the caller's table, row by row (KS statistic: rows in ascending-probability order)
    double h_absmax = 0;          // max |prob| (set with the table)
    int fix_shift = -1;           // d_fix = round(prob * 2^fix_shift), -1 = not built
    u32 n_table = 0;
    int set_fixed(gasm_ctx* ctx, u64 max_terms);
    int set(gasm_ctx* ctx, const char* bp_kmer, const u64* bp_off, u64 n_table, const double* bp_prob);
    int set_standard(gasm_ctx* ctx, const double* table69904);
    void release();
};

struct ScoreState {
    DBuf d_tbl_off, d_seed, d_gpos, d_poscnt, d_total, d_out_f64, d_out_i32, d_freq, d_pd_off, d_pd, d_seg_empty, d_fxsum, d_first, d_first_off;
    std::vector<u64> h_toff;
    u32 n_paths = 0, n_table = 0;
    size_t stride = 1;                      // entries per output array on the device (paths + 1, or their upper bound + 1)
    const BuildState* graph = nullptr;      // batch scoring of a build's own contigs: the number of paths comes with its report
    bool want_freq = false, want_pd = false, launched = false;
    bool verify = false;                    // the graph scorer compared every read with its contig (GASM_SCORE_VERIFY)
    std::vector<double> h_bp, h_nf, h_nl, h_freq, h_pd;
    std::vector<int32_t> h_breaks, h_len;
    std::vector<u64> h_pd_off;
    bool valid = false;
    void release();
};

// queues a whole build on the ctx stream and returns; pipeline_build_finish (called by every fetch) waits for its report
// and repeats it with a larger configuration if it failed.  *rebuilt: the device arrays were produced anew (a score
// queued behind the first attempt must be queued again).
int pipeline_build(gasm_ctx* ctx, DevReads& rd, int k, u64 genome_len_hint, BuildState& bs);
int pipeline_build_finish(gasm_ctx* ctx, DevReads& rd, BuildState& bs, bool* rebuilt);
int pipeline_build_finish_n(gasm_ctx* ctx, DevReads* rd, u32 n_segments, BuildState& bs, bool* rebuilt);
// building blocks shared with the pooled build (pool.hip)
int plan_build(gasm_ctx* ctx, DevReads& rd, int k, u64 hint, BuildState& bs);
void distinct_caps(BuildState& bs, u32 n_segments);
int launch_distinct(gasm_ctx* ctx, DevReads& rd, BuildState& bs);
int launch_graph(gasm_ctx* ctx, u32 n_segments, BuildState& bs);
int pipeline_fetch_distinct(gasm_ctx* ctx, DevReads& rd, BuildState& bs);
int pipeline_fetch_contigs(gasm_ctx* ctx, DevReads& rd, BuildState& bs);
int pipeline_fetch_graph(gasm_ctx* ctx, DevReads& rd, BuildState& bs);    // h_eflag / h_nxt: per-edge flags and successors
// paths of the build as a DevPaths (packs the contig text on the device; works on a queued build); the host-side numbers
// of the same paths once the build's report has been read
int pipeline_contig_paths(gasm_ctx* ctx, const DevReads& rd, const BuildState& bs, DevPaths& dp);
void pipeline_contig_paths_host(const DevReads& rd, const BuildState& bs, DevPaths& dp);
// graph != nullptr: dp holds the contigs of that build (same order), so reads are matched through the edge list
int pipeline_score_launch(gasm_ctx* ctx, DevReads& rd, DevPaths& dp, int kmer, const ScoreTable& tb, bool want_freq,
                          bool want_pd, ScoreState& ss, const BuildState* graph);
int pipeline_score_fetch(gasm_ctx* ctx, ScoreState& ss);
// batch scoring can go through the build's graph (queued without waiting for the build) when every read holds a k-mer
bool pipeline_score_uses_graph(const DevReads& rd, const BuildState& graph);
// Levenshtein distance of every path of `dp` against `target` (ASCII) on the GPU (k_levenshtein).  *done = false when
// the target holds a byte outside ACGT (the packed form cannot represent it): the caller then uses the host routine.
// Two-sample KS statistic of every path's path_freq against the genome's per-position window probabilities
// (lib/DeNovoAssembler.R:414-424); needs the position counters of a general (non-graph) pipeline_score_launch.
int pipeline_ks(gasm_ctx* ctx, DevPaths& dp, ScoreState& ss, const ScoreTable& tb, const char* genome, u64 genome_len, int kmer, std::vector<double>& ks);
// contig_frac_len (lib/DeNovoAssembler.R:432-445)
int pipeline_coverage(gasm_ctx* ctx, const long long* start, const long long* len, u64 n, long long seq_len, double* percent);
int pipeline_levenshtein(gasm_ctx* ctx, DevPaths& dp, const char* target, u64 target_len, bool infix, std::vector<int32_t>& lev, bool* done);

// wait for `ticket` to appear at `word` (pinned memory written last by a kernel of the ctx stream); spin, then poll with a deadline
int gasm_wait_word32(gasm_ctx* ctx, const volatile u32* word, u32 ticket);
int gasm_wait_word64(gasm_ctx* ctx, const volatile u64* word, u64 ticket);

// sequence files parsed and packed on the device (ingest.hip); on_device[f] = 0 where the host reader took an irregular file
int ingest_files_device(gasm_ctx* ctx, const char* const* paths, u32 n_files, bool error_on_non_acgt, DBuf& d_words, std::vector<u64>& read_off,
                        std::vector<u64>& seg, u64* dropped, std::vector<u8>& on_device);
