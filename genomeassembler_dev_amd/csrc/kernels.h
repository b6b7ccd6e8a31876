// kernels.h — kernel declarations and the small POD views passed to them by value.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_utils.h"
#include "keyops.h"

// Reads of all segments of a batch, packed 2-bit in one base stream.
struct ReadSet {
    const u64* words;          // packed bases (+2 padding words)
    const u64* read_off;       // n_reads+1 base offsets; nullptr for fixed-length reads back to back; n_reads base positions
                               // of fixed-length reads that do not lie back to back (read_span)
    const u64* seg_read_off;   // n_segments+1 read indices
    const u32* seg_tile_start; // n_segments+1 tile indices (depends on the tile width of the launch)
    const uint4* tile_info;    // per tile {segment, reads in the tile, first read lo, first read hi (16 bits) | offset round << 16}
    u32 fixed_len;
    u32 n_segments;
};

// Where read r lies in the packed stream.  Three layouts: ragged (read_off[n+1], fixed_len = 0), fixed length back to back
// (read_off = nullptr), and fixed length at given base positions (read_off[n] with fixed_len > 0: the reads of a pooled
// build arrive as word-aligned pieces from every rank, pipeline.hip "pool").
__device__ __forceinline__ void read_span(const ReadSet& rs, u64 r, u64* p0, u32* len) {
    if (rs.fixed_len) { *p0 = rs.read_off ? rs.read_off[r] : r * rs.fixed_len; *len = rs.fixed_len; }
    else { const u64 a = rs.read_off[r]; *p0 = a; *len = (u32)(rs.read_off[r + 1] - a); }
}

// Sorted distinct k-mers (= distinct edges) of all segments, dense, with the per-(segment,bucket) directory.
struct GraphView {
    const void* dk_key;  // D_total keys (u64 or K128), sorted inside each segment
    const u32* dstart;   // n_segments * 2^bbits + 1
    const u16* fdir;     // fine directory: per bucket 2^fbits + 1 offsets (relative to the bucket) of the key bins the
                         // de-duplication kernel sorted by — the next fbits key bits below the bucket prefix
    int k;
    int bbits;
    int fbits;
};

// Lower bound of k-mer t among the distinct k-mers of segment `seg`: bucket by the first bbits bits, bin by the next
// fbits bits (about 1.5 keys per bin), then a search inside the bin.  *bucket_hi = end of the bucket: the result is a
// valid lower bound inside [bucket start, *bucket_hi].
template <class K>
__device__ __forceinline__ u32 graph_lower_bound(const GraphView& gv, u32 seg, const K& t, u32* bucket_hi) {
    const K* dk = reinterpret_cast<const K*>(gv.dk_key);
    const u32 nb = 1u << gv.bbits;
    const int low = 2 * gv.k - gv.bbits;
    const u32 gb = seg * nb + (gv.bbits ? kfield(t, low) : 0u);
    // (two neighbours of a directory = ONE load each, unaligned: the lookups of the graph kernels are bound by the number of
    // line requests they send to L2 — ~16 per clock and XCD —, and a pair of 2- or 4-byte loads are two requests)
    uint2 dd;
    __builtin_memcpy(&dd, gv.dstart + gb, 8);
    const u32 base = dd.x;
    *bucket_hi = dd.y;
    const int bshift = low > gv.fbits ? low - gv.fbits : 0;
    const u32 nbin = 1u << gv.fbits;
    const u16* f = gv.fdir + (u64)gb * (nbin + 1) + (kfield(t, bshift) & (nbin - 1));
    u32 ff;
    __builtin_memcpy(&ff, f, 4);
    u32 lo = base + (ff & 0xFFFFu);
    u32 hi = base + (ff >> 16);
    // (a directory that is not one — a bucket of a failed build attempt whose kernels were queued ahead of its report —
    // must still end the search: bounds inside the bucket, in order)
    if (hi > *bucket_hi) hi = *bucket_hi;
    if (lo > hi) lo = hi;
    while (hi - lo > 4) {                          // only skewed bins are this long
        const u32 m = (lo + hi) >> 1;
        if (kless(dk[m], t)) lo = m + 1;
        else hi = m;
    }
    while (lo < hi && kless(dk[lo], t)) ++lo;      // bins usually hold one or two keys
    return lo;
}

// Paths (contigs or caller-supplied sequences) of all segments, packed, for scoring.
struct PathSet {
    const u64* words;        // packed bases of all paths, concatenated without gaps (+2 padding words)
    const u64* p_off;        // n_paths+1 base offsets
    const u32* seg_path_off; // n_segments+1 path indices
    u32 n_segments;
};

#define GASM_KT 16          // k-mers per thread and round of the tile kernels for 64-bit keys (8 for 128-bit keys)
#define GASM_SCORE_PATH_CAP 6144   // paths of one segment whose score accumulators k_score_reads_graph keeps in LDS
#define GASM_TBL 4096       // slots of the large LDS de-duplication table (the small one has 2048)
#define GASM_TBL_LIMIT 2816 // distinct keys one bucket may hold (11/16 of the table) before the host re-partitions

// ---- kernels_build.hip
__global__ void k_pack_ascii(const u8* ascii, u64 nbases_host, const u64* nbases_dev, u64* words, u32* err);
#define GASM_TILE_WG 512     // threads of a tile workgroup (k_tile_hist, k_bucket_scatter)
template <class K> __global__ void k_tile_hist(ReadSet rs, const uint4* tinfo, int k, int bbits, u32 g, u32 n_tiles, ushort4* tcnt);
__global__ void k_tile_scan(ReadSet rs, int bbits, u32 padm, const ushort4* tcnt, u32* toff, u32* hist, u32* flags);
template <class TO> __global__ void k_scan_excl(const u32* in, TO* out, u32 n);
__global__ void k_copy_u64(const u64* a, u64* b, u32 n);
template <class K>
__global__ void k_bucket_scatter(ReadSet rs, const uint4* tinfo, int k, int bbits, u32 g, u32 padm, u32 n_tiles, const u64* bstart, const u32* toff,
                                 const ushort4* tcnt, K* keys, u64 scratch);
#define GASM_RT_MAX 16      // rounds (read groups x offset rounds) one scatter tile may hold
template <class K>
__global__ void k_bucket_partition(ReadSet rs, const uint4* tinfo, int k, int bbits, u32 g, u32 padm, u32 n_tiles, const u64* bstart, u32* cursor,
                                   K* keys, u64 scratch, u32* flags);
template <class K, int TBL>
__global__ void k_bucket_dedup(K* keys, u32* mult, const u64* bstart, const u32* blen, u32* bucket_d, u32* overflow, u16* fdir, int low_bits, int dbg,
                               unsigned long long* stamps, u32* dstart);
#define GASM_BUCKET_MAX 65535   // distinct keys of one bucket (16-bit fine directory)
template <class K>
__global__ void k_bucket_dedup_multi(const K* keys, K* keys_out, u32* mult, const u64* bstart, u32* bucket_d, u32* overflow, u16* fdir, int low_bits);
template <class K>
__global__ void k_bucket_gather(const K* keys, const u32* mult, const u64* bstart, const u32* dstart, K* dk_key, u32* dk_cnt, u32* claim, u8* eflag,
                                u32* flags);
template <class K> __global__ void k_edge_target(GraphView gv, u32 n_segments, u32 chunks, u32* tgt, u32* claim);
__global__ void k_edge_multi(GraphView gv, u32 n_segments, u32 chunks, const u32* tgt, const u32* claim, u8* eflag);
template <class K>
__global__ void k_node_flags(GraphView gv, u32 n_segments, u32 chunks, const u32* claim, u8* eflag, u64* link, u32* clen);
__global__ void k_edge_next(GraphView gv, u32 n_segments, u32 chunks, const u32* tgt, const u8* eflag, u32* nxt, u64* link);
__global__ void k_link_jump(GraphView gv, u32 n_segments, u32 chunks, u64* link, const u32* prev_active, u32* active, int jumps, const u32* nxt,
                            u32* clen);
__global__ void k_rank_rulers(GraphView gv, u32 n_segments, u32 chunks, u64* link, u32* rtab, u32 rshift, u32* flags);
__global__ void k_rank_lds(GraphView gv, const u32* rtab, u64* link, int max_rounds, u32 rshift, u32 lds_entries, u32* flags);
__global__ void k_chain_len(const u32* nxt, const u64* link, u32* clen, const u32* n_edges_p);
__global__ void k_contig_scan(GraphView gv, const u8* eflag, const u32* clen, u32* e_cid, u64* e_coff, u32* seg_ncontig,
                              u64* seg_cbases, u32* done, u32* seg_cstart, u64* seg_bstart, const u32* flags, u32* report, u32 ticket);
__global__ void k_contig_place(GraphView gv, const u8* eflag, const u32* seg_cstart, const u64* seg_bstart, u32* e_cid, u64* e_coff,
                               u64* c_off, u32 n_segments, u32 chunks);
template <class K>
__global__ void k_contig_emit(GraphView gv, const u64* link, const u32* nxt, const u64* e_coff, u8* out, u32 n_segments, u32 chunks);

// ---- kernels_pool.hip
template <class K>
__global__ void k_pack_runs(const K* keys, const u32* mult, const u64* bstart, const u32* bucket_d, const u32* gb_list, const u64* dst_off, K* out_keys,
                            u32* out_cnt);
template <class K, int TBL>
__global__ void k_bucket_merge(const K* in_keys, const u32* in_cnt, const u64* run_off, const u32* run_len, u32 n_src, K* out_keys, u32* out_cnt,
                               const u64* bstart, u32* bucket_d, u32* overflow, u16* fdir, int low_bits, const u64* src_base);
__global__ void k_repack_reads(const u64* src, const u64* dir, u64* words_out);
__global__ void k_piece_positions(const u64* piece_first, const u64* piece_word_off, u32 n_pieces, u64 n_reads, u32 fixed_len, u64* pos);
__global__ void k_slice_last(const u32* a, const u64* off, u32 n, u32* out);

// ---- kernels_asm.hip
__global__ void k_chain_expand(const u64* cwords, const u64* c_off, const u64* sig_off, const u32* elem_contig, const u32* elem_skip, const u64* elem_pos,
                               const u64* out_off, u32 n_scaffolds, u64* out, u64 n_words);
struct ChainSigs {             // the scaffolds' signatures beside their text (k_str_bitonic skips what two chains share)
    const u64* sig_off;        // n + 1
    const u32* elem_contig;
    const u32* elem_skip;
    const u64* elem_pos;       // where the element's own bases start in its scaffold
};
__global__ void k_str_bitonic(const u64* words, const u64* off, u32* idx, u32 n_pow2, u32 kk, u32 j, ChainSigs cs);
__global__ void k_str_bitonic_block(const u64* words, const u64* off, u32* idx, u32 n_pow2, u32 kk, u32 j0, ChainSigs cs);
__global__ void k_str_adjacent_eq(const u64* words, const u64* off, const u32* idx, u32 n, u8* same_as_prev, ChainSigs cs);
__global__ void k_unpack_ascii(const u64* words, u64 nbases, u8* out);
__global__ void k_guided_chain(PathSet ps, const unsigned long long* fx, int k, u32* g_next, u32* g_prev, u32* dbg_order);
__global__ void k_asm_match(const u64* cwords, const u64* c_off, u32 n, int k, u8* match, u8* row_any, u8* level_any);
__global__ void k_asm_merge(const u32* perm, u32 rows, u32 n, int k, const u32* clen, const u8* match, const u8* row_any, const u8* level_any,
                            const u64* cwords, const u64* c_off, u32* out_next, u8* out_ov, u32* out_heads, u32* out_nchains, u8* need_host);

// ---- kernels_sim.hip
__global__ void k_sim_weights(const u64* gwords, const u64* gbase, const u64* woff, u32 n_segments, int kmer, const long long* fixw, u64* w);
template <class T> __global__ void k_seg_scan_incl(T* a, const u64* off);
__global__ void k_sim_draw(const u64* cum, const u64* woff, const u64* doff, const u64* glen, u64 seed, u32 read_len, u32* start, u32* keep);
__global__ void k_sim_compact(const u32* start, const u32* keep, const u32* rank, const u64* doff, const u64* seg_read_off, u32* kept_start);
__global__ void k_sim_extract(const u64* gwords, const u64* gbase, const u64* seg_read_off, const u32* kept_start, u32 read_len, unsigned long long* out);

// ---- kernels_score.hip
struct SeedTable {
    u64* seed;            // slot -> seed value
    u32* gpos;            // slot -> read index (GASM_NONE32 = empty)
    const u64* tbl_off;   // n_segments+1 slot offsets; every segment's table size is a power of two
};
__global__ void k_read_insert(ReadSet rs, SeedTable st, int w);
__global__ void k_path_scan(ReadSet rs, PathSet ps, SeedTable st, const u64* seg_base_off, int w, const u64* first_off, u32* first, u32 seg0,
                            u32 path_lo, u32 path_hi);
__global__ void k_first_to_poscnt(const u32* first, u64 n, u32* poscnt);
template <class K>
__global__ void k_score_reads_graph(ReadSet rs, GraphView gv, const u64* link, const u32* e_cid, PathSet ps, const long long* dfix,
                                    int kmer, u32 reads_per_wg, u32 chunks, u32 lds_paths, u32* cnt, unsigned long long* sum, int verify, u32* verify_flag);
__global__ void k_levenshtein(PathSet ps, u32 n_paths, const u64* twords, u32 nt, int infix, u8* carry_ws, u64 carry_stride, int32_t* out);
__global__ void k_levenshtein2(PathSet ps, u32 n_paths, const u64* twords, u32 nt, int infix, uint4* carry_ws, u64 carry_stride, int32_t* out);
__global__ void k_score_zero(u32* cnt, unsigned long long* sum, const u32* n_paths_p);
__global__ void k_score_finish(PathSet ps, const u32* cnt, const unsigned long long* sum, const long long* dfix, const u64* seg_empty,
                               int kmer, double inv_scale, double* bp_score, double* norm_freq, double* norm_len, int32_t* kmer_breaks,
                               int32_t* seq_len, const u32* n_paths_p);
__global__ void k_path_reduce(PathSet ps, const u32* poscnt, const u32* extra, const double* dprob, int kmer,
                              double* bp_score, double* norm_freq, double* norm_len, int32_t* kmer_breaks,
                              int32_t* seq_len, u32 n_paths);
__global__ void k_path_freq(PathSet ps, const u32* poscnt, const u32* total, const int32_t* drow, int kmer, u32 n_table,
                            u32* freq_cnt, u32 n_paths);
__global__ void k_ks_genome_hist(const u64* gwords, u64 glen, int kmer, const int32_t* drow, u32* hist);
__global__ void k_path_ks(PathSet ps, const u32* poscnt, const int32_t* drow, int kmer, u32 n_table, const double* pv, const u32* cumy, u32* scratch,
                          double* out, u32 n_paths, const u32* list);
__global__ void k_path_ks2(PathSet ps, const u32* poscnt, const int32_t* drow, int kmer, u32 n_table, const double* pv, const u32* cumy, const u32* run_end,
                           u32* scratch, double* out, u32* flags, u32 n_paths, u32 bins);
__global__ void k_cover_mark(const long long* start, const long long* len, u64 n, long long seq_len, int* diff);
__global__ void k_cover_count(const int* diff, long long seq_len, unsigned long long* covered);
__global__ void k_prob_dist(PathSet ps, const double* dprob, int kmer, const u64* pd_off, double* out, u32 n_paths);

// ---- kernels_pool.hip: exchange plans (exchange.hip)
__global__ void k_x_flag_word(const u32* flags, u32* row_tail);
__global__ void k_x1_plan(const u32* lens_all, u64 stride, u32 nbt, const u32* order, const u32* dst_first, const u32* mine, u32 n_mine, u32 W, u32 r, u32 limit,
                          u64* send_off, u64* send_tot, u64* run_off, u32* run_len, u64* recv_tot, u64* bstart, u32* flags_or);
__global__ void k_x2_fill(u32* G, u32 nbt, const u32* mine, u32 n_mine, const u32* bucket_d, const u32* flags);
__global__ void k_x2_plan(const u32* G, const u16* own1, u32 W, u32 r, u32 gb_lo, u32 n_out, int bbits, const u32* mine, u32 n_mine, const u32* seg_first,
                          u64* send_off, u64* send_tot, u64* run_off, u32* run_len, u64* recv_tot, u64* bstart, u64* info);
__global__ void k_x_report(const u64* send_tot, const u64* recv_tot, u32 W, const u64* info, u32 n_info, const u32* flags_or, u64* report, u64 ticket);
__global__ void k_x_add_u32(u32* acc, const u32* v, u64 n);
