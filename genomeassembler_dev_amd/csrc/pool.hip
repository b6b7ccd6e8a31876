// pool.hip — the pooled (multi-GPU) build behind the gasm_pool_* entry points of include/gasm.h: SURVEY §8(e) mode 2, the
// "RCCL all-to-all over xGMI to bucket k-mers by hash before the global edge-list merge" of the north star.
//
// One gasm_pool per rank.  The rank holds its share of the reads of ALL segments; the library does the device work of
// every stage and hands the host layer (genomeassembler_dev_amd/pooled.py: torch.distributed over RCCL, or an in-memory
// swap between virtual ranks) packed device buffers to exchange:
//   gasm_pool_local_runs   reads -> k-mers -> per-(segment, bucket) sorted distinct (key, count) runs   [kernels_build.hip]
//   gasm_pool_pack_runs    the runs of the buckets bound for one destination, back to back               [k_pack_runs]
//        ... all-to-all #1: every bucket's runs meet at owner(segment, bucket) ...
//   gasm_pool_merge_runs   runs of one bucket from all sources -> one run, counts added                  [k_bucket_merge]
//        ... all-to-all #2: the merged runs (the global distinct edge list) go to their segment's owner ...
//   gasm_pool_merge_runs   placement of a segment's buckets (one source each) + their fine directories
//   gasm_pool_graph        (k-1)-mer graph, list ranking, contigs of the rank's segments                 [kernels_build.hip]
//   gasm_pool_pack_reads / gasm_pool_set_reads   all-to-all #3: a segment's reads (2-bit) to the segment's owner
//   gasm_pool_score        breakage scores of the rank's contigs against the segment's reads             [kernels_score.hip]
// Which rank owns what is the host layer's decision (a pure function of (segment, bucket, world size), the same on every
// rank); the library sees only bucket lists and run directories, so outputs cannot depend on the number of ranks.
#include <algorithm>

#include "pool.h"

#define POOL_GUARD_BEGIN try {
#define POOL_GUARD_END                                                         \
    } catch (const std::bad_alloc&) {                                          \
        gasm_set_error("out of host memory");                                  \
        return GASM_ERR_CAPACITY;                                              \
    } catch (const std::exception& e) {                                        \
        gasm_set_error("internal error: %s", e.what());                        \
        return GASM_ERR_INVALID;                                               \
    }

static int up(gasm_ctx* ctx, DBuf& b, const void* src, size_t bytes) {
    GCHK(b.ensure(bytes ? bytes : 8));
    if (bytes) HIPCHK(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return GASM_OK;
}

static int read_lens(gasm_pool* p, u32 n) {
    p->h_len.assign(n, 0);
    if (n) HIPCHK(hipMemcpyAsync(p->h_len.data(), p->bs.d_bucket_d.p, (size_t)n * 4, hipMemcpyDeviceToHost, p->ctx->stream));
    u32 fl[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(fl, p->bs.d_flags.p, 8, hipMemcpyDeviceToHost, p->ctx->stream));
    HIPCHK(hipStreamSynchronize(p->ctx->stream));
    return (int)(fl[0] & 3u);   // bit 0 = a bucket overflowed its table, bit 1 = a bucket outgrew its region (single-pass partition)
}

int pool_graph_launch(gasm_pool* p, u32 S, u64 D, u64 maxD) {
    gasm_ctx* ctx = p->ctx;
    BuildState& bs = p->bs;
    HIPCHK(hipSetDevice(ctx->device));
    p->n_local = S;
    p->own.n_segments = S;
    if (!p->reads_set) {
        p->own.n_reads = 0; p->own.fixed_len = p->rd.fixed_len; p->own.min_len = p->own.max_len = p->rd.fixed_len;
        p->own.h_seg_read_off.assign((size_t)S + 1, 0);
        p->own.h_seg_empty.assign(S, 0);
    }
    bs.fetched_distinct = bs.fetched_contigs = false;
    bs.h_dstart.assign((size_t)S + 1, 0); bs.h_seg_cstart.assign((size_t)S + 1, 0); bs.h_seg_bstart.assign((size_t)S + 1, 0);
    bs.d_total = 0; bs.n_contigs = 0; bs.contig_bases = 0;
    bs.D_cap = D; bs.maxD_cap = maxD; bs.maxD_est = (u32)std::max<u64>(1, std::min<u64>(maxD, 0xFFFFFFF0ull));
    bs.n_kmers = D;                 // (> 0 iff there is a graph: what the scorer asks)
    bs.have_actual = false; bs.rank_global = false;
    bs.pending = false;
    p->graphed = true; p->paths_ready = false; p->ss.launched = false; p->ss.valid = false; p->scored = false;
    if (S == 0) return GASM_OK;
    GCHK(bs.d_dstart.ensure(((size_t)p->n_runs + 2) * 4));
    GLAUNCH(ctx, "k_scan_excl", k_scan_excl<u32>, dim3(1), dim3(1024), 0, bs.d_bucket_d.as<u32>(), bs.d_dstart.as<u32>(), p->n_runs);
    GCHK(launch_graph(ctx, S, bs));
    return GASM_OK;
}

int pool_score_launch(gasm_pool* p, int kmer, const double* table, bool wait_for_build) {
    if (!p->graphed || !p->reads_set) { gasm_set_error("scoring needs the rank's graph and the reads of its segments first"); return GASM_ERR_STATE; }
    gasm_ctx* ctx = p->ctx;
    if (!p->table_set || memcmp(p->table_copy.data(), table, GASM_TABLE_ROWS * sizeof(double)) != 0) {
        GCHK(p->tb.set_standard(ctx, table));
        p->table_copy.assign(table, table + GASM_TABLE_ROWS);
        p->table_set = true;
    }
    // through the graph (every read holds a k-mer) the scoring needs no size from the host and is queued behind the build as it is
    const bool through_graph = p->n_local && pipeline_score_uses_graph(p->own, p->bs);
    if (wait_for_build || !through_graph) GCHK(pipeline_build_finish_n(ctx, nullptr, p->n_local, p->bs, nullptr));
    if (!p->paths_ready) {
        GCHK(pipeline_contig_paths(ctx, p->own, p->bs, p->dp));
        p->paths_ready = true;
    }
    if (!p->bs.pending) pipeline_contig_paths_host(p->own, p->bs, p->dp);
    GCHK(pipeline_score_launch(ctx, p->own, p->dp, kmer, p->tb, false, false, p->ss, &p->bs));
    p->scored = true; p->score_kmer = kmer;
    return GASM_OK;
}

int pool_finish(gasm_pool* p) {
    bool rebuilt = false;
    GCHK(pipeline_build_finish_n(p->ctx, nullptr, p->n_local, p->bs, &rebuilt));
    if (rebuilt && p->scored) {
        p->paths_ready = false;
        GCHK(pool_score_launch(p, p->score_kmer, p->table_copy.data(), true));
    }
    return GASM_OK;
}

extern "C" {

int gasm_pool_create(gasm_ctx* ctx, const char* reads, uint64_t n_reads, uint32_t fixed_len, const uint64_t* seg_read_off,
                     uint32_t n_segments, gasm_pool** out) {
    POOL_GUARD_BEGIN
    if (!ctx || !out) { gasm_set_error("gasm_pool_create: null argument"); return GASM_ERR_INVALID; }
    *out = nullptr;
    if (fixed_len == 0) { gasm_set_error("pooled builds take fixed-length reads"); return GASM_ERR_INVALID; }
    gasm_pool* p = new gasm_pool();
    p->ctx = ctx;
    const int st = p->rd.upload(ctx, reads, nullptr, n_reads, fixed_len, seg_read_off, n_segments);
    if (st != GASM_OK) { gasm_pool_free(p); return st; }
    *out = p;
    return GASM_OK;
    POOL_GUARD_END
}

void gasm_pool_free(gasm_pool* p) {
    if (!p) return;
    if (p->ctx) { (void)hipSetDevice(p->ctx->device); (void)hipStreamSynchronize(p->ctx->stream); }
    p->rd.release(); p->bs.release(); p->own.release(); p->dp.release(); p->tb.release(); p->ss.release();
    p->d_list.release(); p->d_off.release(); p->d_roff.release(); p->d_rlen.release();
    delete p;
}

int gasm_pool_key_words(const gasm_pool* p) { return p ? p->bs.words : 0; }

int gasm_pool_local_runs(gasm_pool* p, int k, int bbits, const uint32_t** run_len) {
    POOL_GUARD_BEGIN
    if (!p || !run_len) { gasm_set_error("gasm_pool_local_runs: null argument"); return GASM_ERR_INVALID; }
    gasm_ctx* ctx = p->ctx;
    BuildState& bs = p->bs;
    GCHK(plan_build(ctx, p->rd, k, 0, bs));
    if (bbits < 0 || bbits > bs.bb_cap) { gasm_set_error("bbits = %d out of range (0..%d for k = %d)", bbits, bs.bb_cap, k); return GASM_ERR_INVALID; }
    const u32 S = p->rd.n_segments;
    bs.bbits = bbits;
    bs.have_actual = false;
    p->graphed = false; p->paths_ready = false; p->ss.launched = false; p->ss.valid = false;
    p->n_runs = S << bbits;
    if (bs.n_kmers == 0) {
        // this rank holds no k-mer: empty runs everywhere (the arrays the later stages read still have to exist)
        GCHK(bs.d_bucket_d.ensure((size_t)p->n_runs * 4 + 8));
        GCHK(bs.d_bstart.ensure(((size_t)p->n_runs + 1) * 8));
        GCHK(bs.d_keys.ensure(64)); GCHK(bs.d_mult.ensure(64));
        HIPCHK(hipMemsetAsync(bs.d_bucket_d.p, 0, (size_t)p->n_runs * 4 + 8, ctx->stream));
        bs.part_valid = false;
        HIPCHK(hipMemsetAsync(bs.d_bstart.p, 0, ((size_t)p->n_runs + 1) * 8, ctx->stream));
        HIPCHK(hipMemsetAsync(bs.d_flags.p, 0, 256, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        p->h_len.assign(p->n_runs, 0);
        *run_len = p->h_len.data();
        return GASM_OK;
    }
    bs.small_tbl = true;        // the caller chose bbits for buckets of <= ~900 distinct k-mers; a bucket that does not fit retries below
    for (;;) {
        distinct_caps(bs, S);
        GCHK(launch_distinct(ctx, p->rd, bs));
        const int ov = read_lens(p, p->n_runs);
        if (ov < 0) return ov;
        if (!ov) break;
        if ((ov & 2) && bs.single_pass) { bs.single_pass = false; continue; }
        if (bs.small_tbl && bs.words == 1) { bs.small_tbl = false; continue; }
        gasm_set_error("a k-mer bucket of this rank holds more than %d distinct k-mers with %d bucket bits", GASM_TBL_LIMIT, bbits);
        return GASM_ERR_CAPACITY;
    }
    *run_len = p->h_len.data();
    return GASM_OK;
    POOL_GUARD_END
}

int gasm_pool_pack_runs(gasm_pool* p, const uint32_t* bucket_ix, uint64_t n, void* d_keys_out, void* d_counts_out) {
    POOL_GUARD_BEGIN
    if (!p || (n && !bucket_ix)) { gasm_set_error("gasm_pool_pack_runs: null argument"); return GASM_ERR_INVALID; }
    if (n == 0) return GASM_OK;
    gasm_ctx* ctx = p->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    std::vector<u64> off(n);
    u64 run = 0;
    for (u64 i = 0; i < n; ++i) {
        if (bucket_ix[i] >= p->n_runs) { gasm_set_error("bucket index %u out of range (%u runs)", bucket_ix[i], p->n_runs); return GASM_ERR_INVALID; }
        off[i] = run;
        run += p->h_len[bucket_ix[i]];
    }
    if (run == 0) return GASM_OK;                 // every listed run is empty: nothing to write (the buffers may be null)
    if (!d_keys_out || !d_counts_out) { gasm_set_error("gasm_pool_pack_runs: null output buffer"); return GASM_ERR_INVALID; }
    GCHK(up(ctx, p->d_list, bucket_ix, n * 4));
    GCHK(up(ctx, p->d_off, off.data(), n * 8));
    const BuildState& bs = p->bs;
    if (bs.words == 1) {
        GLAUNCH(ctx, "k_pack_runs", k_pack_runs<u64>, dim3((u32)n), dim3(GASM_WG), 0, bs.d_keys.as<u64>(), bs.d_mult.as<u32>(), bs.d_bstart.as<u64>(),
                bs.d_bucket_d.as<u32>(), p->d_list.as<u32>(), p->d_off.as<u64>(), static_cast<u64*>(d_keys_out), static_cast<u32*>(d_counts_out));
    } else {
        GLAUNCH(ctx, "k_pack_runs", k_pack_runs<K128>, dim3((u32)n), dim3(GASM_WG), 0, bs.d_keys.as<K128>(), bs.d_mult.as<u32>(), bs.d_bstart.as<u64>(),
                bs.d_bucket_d.as<u32>(), p->d_list.as<u32>(), p->d_off.as<u64>(), static_cast<K128*>(d_keys_out), static_cast<u32*>(d_counts_out));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));     // the buffers belong to the caller's streams from here on
    return GASM_OK;
    POOL_GUARD_END
}

int gasm_pool_merge_runs(gasm_pool* p, uint32_t n_out, uint32_t n_src, const uint64_t* run_off, const uint32_t* run_len, const void* d_keys_in,
                         const void* d_counts_in, const uint32_t** merged_len) {
    POOL_GUARD_BEGIN
    if (!p || !merged_len || (n_out && n_src && (!run_off || !run_len))) { gasm_set_error("gasm_pool_merge_runs: null argument"); return GASM_ERR_INVALID; }
    if (p->bs.k == 0) { gasm_set_error("gasm_pool_merge_runs before gasm_pool_local_runs"); return GASM_ERR_STATE; }
    gasm_ctx* ctx = p->ctx;
    BuildState& bs = p->bs;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t KB = 8 * (size_t)bs.words;
    const u64 limit = bs.words == 1 ? GASM_TBL_LIMIT : GASM_TBL_LIMIT / 2;
    // capacity of every output run: the union of its inputs, at most what the table holds
    std::vector<u64> bstart((size_t)n_out + 1, 0);
    for (u32 j = 0; j < n_out; ++j) {
        u64 sum = 0;
        for (u32 s = 0; s < n_src; ++s) sum += run_len[(size_t)j * n_src + s];
        bstart[j + 1] = bstart[j] + std::min(sum, limit);
    }
    // (the previous runs have been packed and sent: their arrays are overwritten)
    GCHK(bs.d_keys.ensure(std::max<u64>(bstart[n_out], 1) * KB));
    GCHK(bs.d_mult.ensure(std::max<u64>(bstart[n_out], 1) * 4));
    bs.part_valid = false;                      // (d_bstart no longer holds the partition's region layout)
    GCHK(up(ctx, bs.d_bstart, bstart.data(), bstart.size() * 8));
    GCHK(bs.d_bucket_d.ensure((size_t)n_out * 4 + 8));
    bs.small_tbl = bs.words == 2;               // the merge uses 4096-slot tables for 64-bit keys, 2048-slot ones for 128-bit keys
    bs.fbits = bs.words == 1 ? 10 : 9;
    GCHK(bs.d_fdir.ensure(((size_t)n_out + 1) * ((1u << bs.fbits) + 1) * 2));
    GCHK(up(ctx, p->d_roff, run_off, (size_t)n_out * n_src * 8));
    GCHK(up(ctx, p->d_rlen, run_len, (size_t)n_out * n_src * 4));
    HIPCHK(hipMemsetAsync(bs.d_flags.p, 0, 256, ctx->stream));
    if (n_out) {
        if (bs.words == 1) {
            GLAUNCH(ctx, "k_bucket_merge", (k_bucket_merge<u64, 4096>), dim3(n_out), dim3(GASM_WG), 0, static_cast<const u64*>(d_keys_in),
                    static_cast<const u32*>(d_counts_in), p->d_roff.as<u64>(), p->d_rlen.as<u32>(), n_src, bs.d_keys.as<u64>(), bs.d_mult.as<u32>(),
                    bs.d_bstart.as<u64>(), bs.d_bucket_d.as<u32>(), bs.d_flags.as<u32>(), bs.d_fdir.as<u16>(), 2 * bs.k - bs.bbits, (const u64*)nullptr);
        } else {
            GLAUNCH(ctx, "k_bucket_merge", (k_bucket_merge<K128, 2048>), dim3(n_out), dim3(GASM_WG), 0, static_cast<const K128*>(d_keys_in),
                    static_cast<const u32*>(d_counts_in), p->d_roff.as<u64>(), p->d_rlen.as<u32>(), n_src, bs.d_keys.as<K128>(), bs.d_mult.as<u32>(),
                    bs.d_bstart.as<u64>(), bs.d_bucket_d.as<u32>(), bs.d_flags.as<u32>(), bs.d_fdir.as<u16>(), 2 * bs.k - bs.bbits, (const u64*)nullptr);
        }
    }
    p->n_runs = n_out;
    const int ov = read_lens(p, n_out);
    if (ov < 0) return ov;
    if (ov) {
        gasm_set_error("a merged k-mer bucket holds more than %llu distinct k-mers with %d bucket bits: use more bucket bits", (unsigned long long)limit, bs.bbits);
        return GASM_ERR_CAPACITY;
    }
    p->graphed = false;
    *merged_len = p->h_len.data();
    return GASM_OK;
    POOL_GUARD_END
}

int gasm_pool_graph(gasm_pool* p, uint32_t n_local_segments) {
    POOL_GUARD_BEGIN
    if (!p) { gasm_set_error("pool is null"); return GASM_ERR_INVALID; }
    const u32 S = n_local_segments, nb = 1u << p->bs.bbits;
    if ((u64)S * nb != p->n_runs) { gasm_set_error("the current runs cover %u buckets, %u segments need %llu", p->n_runs, S, (unsigned long long)S * nb); return GASM_ERR_STATE; }
    // sizes are exact here: the merged run lengths are on the host
    u64 D = 0, maxD = 0;
    for (u32 s = 0; s < S; ++s) {
        u64 d = 0;
        for (u32 b = 0; b < nb; ++b) d += p->h_len[(size_t)s * nb + b];
        D += d; maxD = std::max(maxD, d);
    }
    return pool_graph_launch(p, S, D, maxD);
    POOL_GUARD_END
}

// ---- reads to their segment's owner ------------------------------------------------------------------------------
int gasm_pool_piece_words(gasm_pool* p, uint32_t seg_lo, uint32_t seg_hi, uint64_t* n_words) {
    if (!p || !n_words || seg_lo > seg_hi || seg_hi > p->rd.n_segments) { gasm_set_error("gasm_pool_piece_words: bad argument"); return GASM_ERR_INVALID; }
    for (u32 s = seg_lo; s < seg_hi; ++s) {
        const u64 bases = (p->rd.h_seg_read_off[s + 1] - p->rd.h_seg_read_off[s]) * (u64)p->rd.fixed_len;
        n_words[s - seg_lo] = (bases + 31) / 32;
    }
    return GASM_OK;
}

int gasm_pool_pack_reads(gasm_pool* p, uint32_t seg_lo, uint32_t seg_hi, void* d_words_out) {
    POOL_GUARD_BEGIN
    if (!p || seg_lo > seg_hi || seg_hi > p->rd.n_segments) { gasm_set_error("gasm_pool_pack_reads: bad argument"); return GASM_ERR_INVALID; }
    gasm_ctx* ctx = p->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    const u32 np = seg_hi - seg_lo;
    if (np == 0) return GASM_OK;
    std::vector<u64> dir((size_t)np * 3);
    u64 woff = 0, max_words = 0;
    for (u32 s = seg_lo; s < seg_hi; ++s) {
        const u64 b0 = p->rd.h_seg_read_off[s] * (u64)p->rd.fixed_len, b1 = p->rd.h_seg_read_off[s + 1] * (u64)p->rd.fixed_len;
        const u64 nw = (b1 - b0 + 31) / 32;
        dir[3 * (size_t)(s - seg_lo)] = b0; dir[3 * (size_t)(s - seg_lo) + 1] = b1; dir[3 * (size_t)(s - seg_lo) + 2] = woff;
        woff += nw;
        max_words = std::max(max_words, nw);
    }
    if (woff == 0) return GASM_OK;
    if (!d_words_out) { gasm_set_error("gasm_pool_pack_reads: null buffer"); return GASM_ERR_INVALID; }
    if (np > 65535) { gasm_set_error("at most 65535 pieces per call"); return GASM_ERR_CAPACITY; }
    GCHK(up(ctx, p->d_off, dir.data(), dir.size() * 8));
    GLAUNCH(ctx, "k_repack_reads", k_repack_reads, dim3(std::max(1u, std::min<u32>(ceil_div_u64(max_words, GASM_WG), 64u)), np), dim3(GASM_WG), 0,
            p->rd.d_words.as<u64>(), p->d_off.as<u64>(), static_cast<u64*>(d_words_out));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return GASM_OK;
    POOL_GUARD_END
}

int gasm_pool_set_reads(gasm_pool* p, const void* d_words, uint64_t n_words, uint32_t n_pieces, const uint32_t* piece_seg, const uint64_t* piece_reads,
                        const uint64_t* piece_word_off) {
    POOL_GUARD_BEGIN
    if (!p || (n_pieces && (!piece_seg || !piece_reads || !piece_word_off)) || (n_words && !d_words)) { gasm_set_error("gasm_pool_set_reads: null argument"); return GASM_ERR_INVALID; }
    if (!p->graphed) { gasm_set_error("gasm_pool_set_reads before gasm_pool_graph"); return GASM_ERR_STATE; }
    gasm_ctx* ctx = p->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    DevReads& o = p->own;
    const u32 S = p->n_local, flen = p->rd.fixed_len;
    o.n_segments = S; o.fixed_len = flen; o.positioned = true; o.min_len = o.max_len = flen; o.n_empty = 0;
    o.h_seg_empty.assign(S, 0);
    o.h_seg_read_off.assign((size_t)S + 1, 0);
    for (u32 i = 0; i < n_pieces; ++i) {
        if (piece_seg[i] >= S || (i && piece_seg[i] < piece_seg[i - 1])) { gasm_set_error("pieces must be listed by local segment"); return GASM_ERR_INVALID; }
        if (piece_word_off[i] + (piece_reads[i] * (u64)flen + 31) / 32 > n_words) { gasm_set_error("piece %u runs past the word buffer", i); return GASM_ERR_INVALID; }
        o.h_seg_read_off[piece_seg[i] + 1] += piece_reads[i];
    }
    for (u32 s = 0; s < S; ++s) o.h_seg_read_off[s + 1] += o.h_seg_read_off[s];
    o.n_reads = o.h_seg_read_off[S];
    o.total_bases = o.n_reads * (u64)flen;
    // the reads' base positions, computed on the device from the piece directory (only pieces with reads are listed)
    std::vector<u64> pfirst, pword;
    u64 r = 0;
    for (u32 i = 0; i < n_pieces; ++i) {
        if (piece_reads[i]) { pfirst.push_back(r); pword.push_back(piece_word_off[i]); }
        r += piece_reads[i];
    }
    GCHK(o.d_words.ensure((n_words + 4) * 8));
    if (n_words) HIPCHK(hipMemcpyAsync(o.d_words.p, d_words, n_words * 8, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(static_cast<char*>(o.d_words.p) + n_words * 8, 0, 32, ctx->stream));
    GCHK(o.d_read_off.ensure(std::max<u64>(o.n_reads, 1) * 8));
    if (o.n_reads) {
        GCHK(up(ctx, p->d_roff, pfirst.data(), pfirst.size() * 8));
        GCHK(up(ctx, p->d_off, pword.data(), pword.size() * 8));
        GLAUNCH(ctx, "k_piece_positions", k_piece_positions, dim3(std::min<u32>(ceil_div_u64(o.n_reads, GASM_WG), (u32)ctx->n_cu * 8u)), dim3(GASM_WG), 0,
                p->d_roff.as<u64>(), p->d_off.as<u64>(), (u32)pfirst.size(), o.n_reads, flen, o.d_read_off.as<u64>());
    }
    GCHK(up(ctx, o.d_seg_read_off, o.h_seg_read_off.data(), ((size_t)S + 1) * 8));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    p->reads_set = true;
    p->ss.launched = false; p->ss.valid = false;
    return GASM_OK;
    POOL_GUARD_END
}

int gasm_pool_score(gasm_pool* p, int kmer, const double* table) {
    POOL_GUARD_BEGIN
    if (!p || !table) { gasm_set_error("gasm_pool_score: null argument"); return GASM_ERR_INVALID; }
    return pool_score_launch(p, kmer, table, true);
    POOL_GUARD_END
}

int gasm_pool_fetch_distinct(gasm_pool* p, const uint64_t** seg_off, const uint64_t** keys, const uint32_t** mult, int* words) {
    POOL_GUARD_BEGIN
    if (!p || !seg_off || !keys || !mult || !words) { gasm_set_error("null argument"); return GASM_ERR_INVALID; }
    if (!p->graphed) { gasm_set_error("fetch before gasm_pool_graph"); return GASM_ERR_STATE; }
    GCHK(pool_finish(p));
    GCHK(pipeline_fetch_distinct(p->ctx, p->own, p->bs));
    *seg_off = p->bs.h_seg_doff.data(); *keys = p->bs.h_dk_key.data(); *mult = p->bs.h_dk_cnt.data(); *words = p->bs.words;
    return GASM_OK;
    POOL_GUARD_END
}

int gasm_pool_fetch_contigs(gasm_pool* p, const uint64_t** seg_contig_off, const uint64_t** off, const char** data) {
    POOL_GUARD_BEGIN
    if (!p || !seg_contig_off || !off || !data) { gasm_set_error("null argument"); return GASM_ERR_INVALID; }
    if (!p->graphed) { gasm_set_error("fetch before gasm_pool_graph"); return GASM_ERR_STATE; }
    GCHK(pool_finish(p));
    GCHK(pipeline_fetch_contigs(p->ctx, p->own, p->bs));
    *seg_contig_off = p->bs.h_seg_coff.data(); *off = p->bs.h_c_off.data(); *data = p->bs.h_contigs.data();
    return GASM_OK;
    POOL_GUARD_END
}

int gasm_pool_fetch_scores(gasm_pool* p, const double** bp_score, const double** norm_by_break_freqs, const double** norm_by_len,
                           const int32_t** kmer_breaks, const int32_t** sequence_len) {
    POOL_GUARD_BEGIN
    if (!p || !bp_score || !norm_by_break_freqs || !norm_by_len || !kmer_breaks || !sequence_len) { gasm_set_error("null argument"); return GASM_ERR_INVALID; }
    GCHK(pool_finish(p));
    GCHK(pipeline_score_fetch(p->ctx, p->ss));
    *bp_score = p->ss.h_bp.data(); *norm_by_break_freqs = p->ss.h_nf.data(); *norm_by_len = p->ss.h_nl.data();
    *kmer_breaks = p->ss.h_breaks.data(); *sequence_len = p->ss.h_len.data();
    return GASM_OK;
    POOL_GUARD_END
}

}  // extern "C"
