// pipeline.hip — host orchestration of the device pipeline.  All launches go to the ctx stream; the host only waits
// where a size is needed to allocate the next stage (distinct count, contig bytes) and at fetch time.
#include <atomic>
#include <chrono>
#include <thread>
#include "pipeline.h"

#include <algorithm>
#include <cmath>

// ---------------------------------------------------------------------------------------------------------------
// small local kernels
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_set_u64(u64* p, u64 v) { *p = v; }

// empty reads match every path at position 0 (std::string::find("") == 0, lib/DeNovoAssembler.cpp:360)
__global__ void k_add_empty_reads(PathSet ps, const u64* __restrict__ seg_empty, u32* __restrict__ poscnt, u32* __restrict__ total) {
    const u32 seg = blockIdx.y;
    const u32 p = ps.seg_path_off[seg] + blockIdx.x * GASM_WG + threadIdx.x;
    if (p >= ps.seg_path_off[seg + 1]) return;
    const u32 e = (u32)seg_empty[seg];
    if (!e) return;
    if (ps.p_off[p + 1] > ps.p_off[p]) poscnt[ps.p_off[p]] += e;
    else total[p] += e;   // an empty path has no position counter: carried as an extra addend
}

// Tuning / diagnostic knobs, read from the environment once (none changes results except GASM_DBG_DEDUP, an ablation):
//   GASM_DEDUP_TBL=2048|4096   force the de-duplication table size        GASM_SCATTER_WGS=n   scatter workgroups per CU (8)
//   GASM_HIST_WGS=n            histogram workgroups per CU (64)           GASM_DBG_BBITS_ADD=n extra bucket bits (tuning)
//   GASM_DBG_PADM=m            cap the run padding at m + 1 keys          GASM_RANK_GLOBAL=1   whole-GPU list ranking only
//   GASM_RULER_SHIFT=1..4      rulers of the LDS list ranking = every 2^n-th edge
//   GASM_DBG_RANK_ROUNDS=n     cap the LDS ranking rounds (ablation)      GASM_DBG_DEDUP=1|2   loads only / no ordering (ablation)
//   GASM_DBG_STAMPS=file       per-phase clock stamps of k_bucket_dedup to stderr and `file`
//   GASM_SYNC_BUILD=1          every build waits for its own report (no queued-ahead builds)
struct Knobs {
    int dedup_tbl = 0, dbg_dedup = 0, padm = -1, scatter_wgs = 8, hist_wgs = 64, rank_rounds = 18, bbits_add = 0, ruler_shift = 0;
    bool rank_global = false, sync_build = false;
    int dedup_warm = 0;
    const char* stamps = nullptr;
    Knobs() {
        if (const char* v = getenv("GASM_DEDUP_TBL")) dedup_tbl = atoi(v);
        if (const char* v = getenv("GASM_DBG_DEDUP")) dbg_dedup = atoi(v);
        if (const char* v = getenv("GASM_DBG_PADM")) padm = atoi(v);
        if (const char* v = getenv("GASM_SCATTER_WGS")) scatter_wgs = std::max(1, atoi(v));
        if (const char* v = getenv("GASM_HIST_WGS")) hist_wgs = std::max(1, atoi(v));
        if (const char* v = getenv("GASM_DBG_BBITS_ADD")) bbits_add = std::max(0, atoi(v));
        if (const char* v = getenv("GASM_RULER_SHIFT")) ruler_shift = std::min(4, std::max(0, atoi(v)));
        if (const char* v = getenv("GASM_DBG_RANK_ROUNDS")) rank_rounds = atoi(v);
        rank_global = getenv("GASM_RANK_GLOBAL") != nullptr;
        sync_build = getenv("GASM_SYNC_BUILD") != nullptr;       // wait for every build's report at once (diagnostic)
        if (const char* v = getenv("GASM_DEDUP_WARM")) dedup_warm = std::max(0, std::min(3, atoi(v)));     // first iterations of the 64-bit de-duplication taken key by key
        stamps = getenv("GASM_DBG_STAMPS");
    }
};
static const Knobs& knobs() { static const Knobs k; return k; }
// knobs read at every build (tests switch them inside one process):
//   GASM_SINGLE_PASS=0     two-pass partition (count, scan, scatter) from the start       GASM_PART_SLACK=percent   room per bucket region (100)
//   GASM_DBG_PART_CAP=n    force the capacity of every bucket region to n keys (exercises the overflow path)

// Wait for a report a kernel writes into pinned host memory: the kernel's last store is `ticket` at `word`.  Spinning on
// that word costs a few microseconds; waking up from hipStreamSynchronize costs 15-20 us (more on a busy host) — twice
// per build, with the GPU idle meanwhile.  After ~2 ms of spinning (a large batch) the thread polls the stream with short
// sleeps, up to a deadline (GASM_WAIT_TIMEOUT_S, default 300 s): a kernel that never finishes becomes an error, not a hang.
template <class T>
static int wait_word(gasm_ctx* ctx, const volatile T* word, T ticket) {
    const auto t0 = std::chrono::steady_clock::now();
    for (u32 spin = 0;; ++spin) {
        if (*word == ticket) { std::atomic_thread_fence(std::memory_order_acquire); return GASM_OK; }
        if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
    static const double limit_s = [] { const char* v = getenv("GASM_WAIT_TIMEOUT_S"); const double x = v && *v ? atof(v) : 300.0; return x > 0 ? x : 300.0; }();
    for (;;) {
        if (*word == ticket) { std::atomic_thread_fence(std::memory_order_acquire); return GASM_OK; }
        const hipError_t q = hipStreamQuery(ctx->stream);
        if (q == hipSuccess) break;                       // the stream has drained: the report is there or never comes
        if (q != hipErrorNotReady) { gasm_set_error("the stream failed while a report was awaited: %s", hipGetErrorString(q)); return GASM_ERR_HIP; }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s) {
            gasm_set_error("no report from the device within %.0f s (GASM_WAIT_TIMEOUT_S): a kernel hangs or the GPU is lost", limit_s);
            return GASM_ERR_HIP;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    if (*word != ticket) { gasm_set_error("a kernel finished without writing its report"); return GASM_ERR_HIP; }
    std::atomic_thread_fence(std::memory_order_acquire);
    return GASM_OK;
}
int gasm_wait_word32(gasm_ctx* ctx, const volatile u32* word, u32 ticket) { return wait_word<u32>(ctx, word, ticket); }
int gasm_wait_word64(gasm_ctx* ctx, const volatile u64* word, u64 ticket) { return wait_word<u64>(ctx, word, ticket); }
static int wait_report(gasm_ctx* ctx, const volatile u32* word, u32 ticket) { return wait_word<u32>(ctx, word, ticket); }
static u32 next_ticket() { static std::atomic<u32> t{1}; u32 v = t.fetch_add(1); return v ? v : t.fetch_add(1); }

// grid of the segment-major kernels (seg_chunk, device_utils.h): 8 x chunks x ceil(S / 8) workgroups
static dim3 seg_grid(u32 chunks, u32 S) { return dim3(8u * chunks * ((S + 7u) / 8u)); }

static int h2d(gasm_ctx* ctx, DBuf& b, const void* src, size_t bytes) {
    GCHK(b.ensure(bytes ? bytes : 8));
    if (bytes) HIPCHK(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return GASM_OK;
}

// Device -> host copy of a small array, then wait.  Goes through the ctx's pinned area (a pageable destination makes
// hipMemcpyAsync stage through the runtime and costs tens of microseconds more per call).
static int d2h_sync(gasm_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) { HIPCHK(hipStreamSynchronize(ctx->stream)); return GASM_OK; }
    if (bytes <= ctx->h_pin_words * sizeof(u64)) {
        HIPCHK(hipMemcpyAsync(ctx->h_pin, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        memcpy(dst, ctx->h_pin, bytes);
    } else {
        HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    return GASM_OK;
}

static int pack_ascii(gasm_ctx* ctx, const u8* d_ascii, u64 nbases, DBuf& words, u32* d_err) {
    const u64 nw = (nbases + 31) / 32;
    GCHK(words.ensure((nw + 4) * 8));     // four zero padding words: a 128-bit rolling window reads up to three words ahead
    // (the kernel also writes the padding words: positions past the last base pack as zero)
    GLAUNCH(ctx, "k_pack_ascii", k_pack_ascii, dim3(std::min<u32>(ceil_div_u64(nw + 4, GASM_WG), (u32)ctx->n_cu * 32u)), dim3(GASM_WG), 0, d_ascii, nbases,
            (const u64*)nullptr, words.as<u64>(), d_err);
    return GASM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// DevReads
// ---------------------------------------------------------------------------------------------------------------
// host-side layout of a batch's reads: per-segment directories, shortest / longest read, empty reads
int DevReads::set_layout(const u64* read_off, u64 n, u32 flen, const u64* seg_off, u32 S) {
    if (S == 0 || !seg_off) { gasm_set_error("need at least one segment and seg_read_off"); return GASM_ERR_INVALID; }
    if (S > 65535) { gasm_set_error("at most 65535 segments per batch (got %u)", S); return GASM_ERR_CAPACITY; }
    if (!read_off && flen == 0 && n) { gasm_set_error("give read_off or fixed_len"); return GASM_ERR_INVALID; }
    if (seg_off[0] != 0 || seg_off[S] != n) { gasm_set_error("seg_read_off must run from 0 to n_reads"); return GASM_ERR_INVALID; }
    for (u32 s = 0; s < S; ++s) if (seg_off[s] > seg_off[s + 1]) { gasm_set_error("seg_read_off not monotone"); return GASM_ERR_INVALID; }
    n_segments = S; n_reads = n; fixed_len = read_off ? 0 : flen;
    positioned = false;
    h_seg_read_off.assign(seg_off, seg_off + S + 1);
    h_seg_empty.assign(S, 0);
    n_empty = 0; min_len = 0; max_len = 0;
    if (read_off) {
        h_read_off.resize(n + 1);
        for (u64 r = 0; r <= n; ++r) {
            if (r && read_off[r] < read_off[r - 1]) { gasm_set_error("read_off not monotone"); return GASM_ERR_INVALID; }
            h_read_off[r] = read_off[r] - read_off[0];
        }
        total_bases = h_read_off[n];
        u32 s = 0;
        for (u64 r = 0; r < n; ++r) {
            while (r >= h_seg_read_off[s + 1]) ++s;
            const u64 L = h_read_off[r + 1] - h_read_off[r];
            if (L > 0xFFFFFFFFull) { gasm_set_error("read longer than 2^32"); return GASM_ERR_CAPACITY; }
            if (L == 0) { ++n_empty; ++h_seg_empty[s]; continue; }
            if (min_len == 0 || L < min_len) min_len = (u32)L;
            if (L > max_len) max_len = (u32)L;
        }
    } else {
        h_read_off.clear();
        total_bases = n * (u64)flen;
        if (n) { min_len = max_len = flen; }
    }
    return GASM_OK;
}

int DevReads::finish_upload(gasm_ctx* ctx) {
    if (!fixed_len) GCHK(h2d(ctx, d_read_off, h_read_off.data(), (n_reads + 1) * 8));
    GCHK(h2d(ctx, d_seg_read_off, h_seg_read_off.data(), ((size_t)n_segments + 1) * 8));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    tiles_ipt = 0;
    static std::atomic<u64> uploads{0};
    upload_id = ++uploads;
    return GASM_OK;
}

int DevReads::upload(gasm_ctx* ctx, const char* reads, const u64* read_off, u64 n, u32 flen, const u64* seg_off, u32 S) {
    if (!ctx) { gasm_set_error("ctx is null"); return GASM_ERR_INVALID; }
    GCHK(set_layout(read_off, n, flen, seg_off, S));
    HIPCHK(hipSetDevice(ctx->device));
    const char* base = read_off ? reads + read_off[0] : reads;
    if (n && !reads && total_bases) { gasm_set_error("reads is null"); return GASM_ERR_INVALID; }
    DBuf ascii, err;
    struct Rel { DBuf &a, &b; ~Rel() { a.release(); b.release(); } } rel{ascii, err};      // (also on the early returns below)
    GCHK(err.ensure(8));
    HIPCHK(hipMemsetAsync(err.p, 0, 8, ctx->stream));
    GCHK(h2d(ctx, ascii, base, total_bases));
    int st = pack_ascii(ctx, ascii.as<u8>(), total_bases, d_words, err.as<u32>());
    u32 herr = 0;
    if (st == GASM_OK && hipMemcpyAsync(&herr, err.p, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) st = GASM_ERR_HIP;
    if (st == GASM_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) { gasm_set_error("sync failed after packing"); st = GASM_ERR_HIP; }
    ascii.release(); err.release();
    if (st != GASM_OK) return st;
    if (herr) { gasm_set_error("reads contain a base outside upper-case ACGT"); return GASM_ERR_NON_ACGT; }
    return finish_upload(ctx);
}

// Reads that are 2-bit packed already (first base most significant, 32 per word, back to back): a quarter of the bytes over
// PCIe, no packing kernel.  Bits past the last base are ignored.
int DevReads::upload_packed(gasm_ctx* ctx, const u64* words, const u64* read_off, u64 n, u32 flen, const u64* seg_off, u32 S) {
    if (!ctx) { gasm_set_error("ctx is null"); return GASM_ERR_INVALID; }
    GCHK(set_layout(read_off, n, flen, seg_off, S));
    if (read_off && read_off[0] != 0) { gasm_set_error("packed reads: read_off must start at 0"); return GASM_ERR_INVALID; }
    if (total_bases && !words) { gasm_set_error("words is null"); return GASM_ERR_INVALID; }
    HIPCHK(hipSetDevice(ctx->device));
    const u64 nw = (total_bases + 31) / 32;
    GCHK(d_words.ensure((nw + 4) * 8));
    if (nw) HIPCHK(hipMemcpyAsync(d_words.p, words, nw * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(static_cast<char*>(d_words.p) + nw * 8, 0, 32, ctx->stream));
    if (nw && (total_bases & 31)) {
        // the tail of the last word must be zero (windows read past the last base)
        const u64 last = words[nw - 1] & (~0ull << (64 - 2 * (total_bases & 31)));
        HIPCHK(hipMemcpyAsync(static_cast<char*>(d_words.p) + (nw - 1) * 8, &last, 8, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    return finish_upload(ctx);
}

// Reads packed on the device already (ingest.hip): the stream is taken over as it is (zero beyond its last base, padding words
// included); only the directories come from the host.
int DevReads::adopt_packed(gasm_ctx* ctx, DBuf& words_dev, const u64* read_off, u64 n, u32 flen, const u64* seg_off, u32 S) {
    if (!ctx) { gasm_set_error("ctx is null"); return GASM_ERR_INVALID; }
    GCHK(set_layout(read_off, n, flen, seg_off, S));
    HIPCHK(hipSetDevice(ctx->device));
    const u64 nw = (total_bases + 31) / 32;
    if (words_dev.cap < (nw + 4) * 8) { gasm_set_error("adopt_packed: the stream is shorter than its directories say"); return GASM_ERR_INVALID; }
    d_words.release();
    std::swap(d_words, words_dev);
    return finish_upload(ctx);
}

// ---------------------------------------------------------------------------------------------------------------
// Simulated reads, made on the device (kernels_sim.hip; lib/GenerateReads.R:235-313)
// ---------------------------------------------------------------------------------------------------------------
int DevReads::simulate(gasm_ctx* ctx, const char* genomes, const u64* genome_off, u32 S, u32 read_len, double coverage, u64 seed, int kmer,
                       const double* table, DBuf& d_kept_start) {
    if (!ctx || !genome_off || S == 0) { gasm_set_error("simulate: bad argument"); return GASM_ERR_INVALID; }
    if (read_len == 0 || !(coverage >= 0.0) || kmer < 1 || kmer > 8) { gasm_set_error("simulate: read_len > 0, coverage >= 0, 1 <= kmer <= 8"); return GASM_ERR_INVALID; }
    HIPCHK(hipSetDevice(ctx->device));
    std::vector<u64> gbase((size_t)S + 1), glen(S), woff((size_t)S + 1, 0), doff((size_t)S + 1, 0);
    u64 max_np = 0, max_nd = 0;
    for (u32 s = 0; s <= S; ++s) {
        if (s && genome_off[s] < genome_off[s - 1]) { gasm_set_error("genome_off not monotone"); return GASM_ERR_INVALID; }
        gbase[s] = genome_off[s] - genome_off[0];
    }
    for (u32 s = 0; s < S; ++s) {
        const u64 L = gbase[s + 1] - gbase[s];
        if (L > 0xFFFFFFF0ull) { gasm_set_error("genome longer than 2^32"); return GASM_ERR_CAPACITY; }
        glen[s] = L;
        const u64 np = L >= (u64)kmer ? L - kmer + 1 : 0;                                // lib/GenerateReads.R:243
        const u64 nd = np ? (u64)std::ceil(coverage * (double)L / (double)read_len) : 0;   // :302
        woff[s + 1] = woff[s] + np;
        doff[s + 1] = doff[s] + nd;
        max_np = std::max(max_np, np); max_nd = std::max(max_nd, nd);
    }
    DBuf ascii, err, gwords, d_gbase, d_glen, d_woff, d_doff, d_w, d_fix, d_start, d_keep, d_rank, d_sro;
    struct Rel { std::vector<DBuf*> v; ~Rel() { for (DBuf* b : v) b->release(); } } rel{{&ascii, &err, &gwords, &d_gbase, &d_glen, &d_woff, &d_doff, &d_w, &d_fix,
                                                                                         &d_start, &d_keep, &d_rank, &d_sro}};
    GCHK(err.ensure(8));
    HIPCHK(hipMemsetAsync(err.p, 0, 8, ctx->stream));
    GCHK(h2d(ctx, ascii, genomes + genome_off[0], gbase[S]));
    GCHK(pack_ascii(ctx, ascii.as<u8>(), gbase[S], gwords, err.as<u32>()));
    GCHK(h2d(ctx, d_gbase, gbase.data(), gbase.size() * 8));
    GCHK(h2d(ctx, d_glen, glen.data(), glen.size() * 8));
    GCHK(h2d(ctx, d_woff, woff.data(), woff.size() * 8));
    GCHK(h2d(ctx, d_doff, doff.data(), doff.size() * 8));
    const long long* fixw = nullptr;
    if (table) {
        // the kmer-long rows of the standard table (rows of lengths 2, 4, 6, 8 in that order), direct-addressed, in fixed point
        if (kmer != 2 && kmer != 4 && kmer != 6 && kmer != 8) { gasm_set_error("the weighted simulator needs kmer in {2,4,6,8} (the table's row lengths)"); return GASM_ERR_INVALID; }
        // fixed-point weights round(p * 2^shift), running sums in 64 bits: shift = 52 unless a segment's sum could then pass
        // 2^62 (kmer = 2 on a genome of more than ~65 kb: the sum used to wrap and k_sim_draw bisected a non-monotone CDF) —
        // then the largest shift that cannot.  The oracle takes the same shift (sim_weight_shift, oracle/orc.py)
        std::vector<long long> fx(87380, 0);
        double max_p = 0;
        {
            u32 src = 0;
            for (u32 Lk = 2; Lk <= 8; Lk += 2) {
                const u32 n = 1u << (2 * Lk);
                for (u32 v = 0; v < n; ++v, ++src) if ((int)Lk == kmer) max_p = std::max(max_p, table[src]);
            }
        }
        int shift = 52;
        while (shift > 0 && std::ldexp(max_p, shift) * (double)std::max<u64>(max_np, 1) >= 4611686018427387904.0) --shift;
        sim_shift = shift;
        u32 src = 0;
        for (u32 Lk = 2; Lk <= 8; Lk += 2) {
            const u32 n = 1u << (2 * Lk), b = ((1u << (2 * Lk)) - 4u) / 3u;
            for (u32 v = 0; v < n; ++v, ++src) if ((int)Lk == kmer) fx[b + v] = std::llrint(std::ldexp(table[src], shift));
        }
        GCHK(h2d(ctx, d_fix, fx.data(), fx.size() * 8));
        fixw = d_fix.as<long long>();
    }
    GCHK(d_w.ensure(std::max<u64>(woff[S], 1) * 8));
    GCHK(d_start.ensure(std::max<u64>(doff[S], 1) * 4));
    GCHK(d_keep.ensure(std::max<u64>(doff[S], 1) * 4));
    GCHK(d_rank.ensure(std::max<u64>(doff[S], 1) * 4));
    const u32 gx_p = std::max(1u, std::min<u32>(ceil_div_u64(std::max<u64>(max_np, 1), GASM_WG), 64u));
    const u32 gx_d = std::max(1u, std::min<u32>(ceil_div_u64(std::max<u64>(max_nd, 1), GASM_WG), 64u));
    GLAUNCH(ctx, "k_sim_weights", k_sim_weights, dim3(gx_p, S), dim3(GASM_WG), 0, gwords.as<u64>(), d_gbase.as<u64>(), d_woff.as<u64>(), S, kmer, fixw, d_w.as<u64>());
    GLAUNCH(ctx, "k_seg_scan_incl", k_seg_scan_incl<u64>, dim3(S), dim3(1024), 0, d_w.as<u64>(), d_woff.as<u64>());
    GLAUNCH(ctx, "k_sim_draw", k_sim_draw, dim3(gx_d, S), dim3(GASM_WG), 0, d_w.as<u64>(), d_woff.as<u64>(), d_doff.as<u64>(), d_glen.as<u64>(), seed, read_len,
            d_start.as<u32>(), d_keep.as<u32>());
    if (doff[S]) HIPCHK(hipMemcpyAsync(d_rank.p, d_keep.p, doff[S] * 4, hipMemcpyDeviceToDevice, ctx->stream));
    GLAUNCH(ctx, "k_seg_scan_incl", k_seg_scan_incl<u32>, dim3(S), dim3(1024), 0, d_rank.as<u32>(), d_doff.as<u64>());
    // kept reads per segment = the last rank of the segment (one small kernel + one copy, not a copy per segment)
    std::vector<u32> last(S, 0);
    u32 herr = 0;
    GCHK(d_sro.ensure((size_t)S * 4 + 8));
    GLAUNCH(ctx, "k_slice_last", k_slice_last, dim3(ceil_div_u64(S, GASM_WG)), dim3(GASM_WG), 0, d_rank.as<u32>(), d_doff.as<u64>(), S, d_sro.as<u32>());
    HIPCHK(hipMemcpyAsync(&herr, err.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(last.data(), d_sro.p, (size_t)S * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (herr) { gasm_set_error("a genome holds a base outside upper-case ACGT"); return GASM_ERR_NON_ACGT; }
    std::vector<u64> sro((size_t)S + 1, 0);
    for (u32 s = 0; s < S; ++s) sro[s + 1] = sro[s] + last[s];
    const u64 n = sro[S];
    GCHK(set_layout(nullptr, n, read_len, sro.data(), S));
    GCHK(h2d(ctx, d_sro, sro.data(), sro.size() * 8));
    GCHK(d_kept_start.ensure(std::max<u64>(n, 1) * 4));
    GLAUNCH(ctx, "k_sim_compact", k_sim_compact, dim3(gx_d, S), dim3(GASM_WG), 0, d_start.as<u32>(), d_keep.as<u32>(), d_rank.as<u32>(), d_doff.as<u64>(), d_sro.as<u64>(),
            d_kept_start.as<u32>());
    const u64 nw = (total_bases + 31) / 32;
    GCHK(d_words.ensure((nw + 4) * 8));
    HIPCHK(hipMemsetAsync(d_words.p, 0, (nw + 4) * 8, ctx->stream));
    if (n) GLAUNCH(ctx, "k_sim_extract", k_sim_extract, dim3(std::max(1u, std::min<u32>(ceil_div_u64(max_nd * ((read_len + 31) / 32), GASM_WG), 256u)), S), dim3(GASM_WG), 0,
                   gwords.as<u64>(), d_gbase.as<u64>(), d_sro.as<u64>(), d_kept_start.as<u32>(), read_len, d_words.as<unsigned long long>());
    return finish_upload(ctx);
}

// Tile table: a tile = up to ipt consecutive reads of one segment at one of orr offset rounds (kernels_build.hip, "Tiles").
int DevReads::set_tiles(gasm_ctx* ctx, u32 ipt, u32 orr) {
    if (tiles_ipt == ipt && tiles_orr == orr) return GASM_OK;
    std::vector<u32>& t = h_seg_tile_start;
    t.assign(n_segments + 1, 0);
    for (u32 s = 0; s < n_segments; ++s) {
        const u64 rs = h_seg_read_off[s + 1] - h_seg_read_off[s];
        const u64 nt = (rs + ipt - 1) / ipt * orr;
        if ((u64)t[s] + nt > 0xFFFFFFF0ull) { gasm_set_error("too many tiles"); return GASM_ERR_CAPACITY; }
        t[s + 1] = t[s] + (u32)nt;
    }
    GCHK(h2d(ctx, d_seg_tile_start, t.data(), t.size() * 4));
    n_tiles = t[n_segments];
    std::vector<u32>& info = h_tile_info;
    info.assign((size_t)n_tiles * 4 + 4, 0);
    for (u32 s = 0; s < n_segments; ++s)
        for (u32 tile = t[s]; tile < t[s + 1]; ++tile) {
            const u32 grp = (tile - t[s]) / orr, o = (tile - t[s]) % orr;
            const u64 first = h_seg_read_off[s] + (u64)grp * ipt;
            const u64 left = h_seg_read_off[s + 1] - first;
            if ((first >> 48) != 0) { gasm_set_error("too many reads"); return GASM_ERR_CAPACITY; }
            u32* e = &info[(size_t)tile * 4];
            e[0] = s; e[1] = (u32)std::min<u64>(left, ipt); e[2] = (u32)first; e[3] = (u32)(first >> 32) | (o << 16);
        }
    GCHK(h2d(ctx, d_tile_info, info.data(), info.size() * 4));
    tiles_ipt = ipt;
    tiles_orr = orr;
    return GASM_OK;
}

ReadSet DevReads::view() const {
    ReadSet v;
    v.words = d_words.as<u64>();
    v.read_off = (fixed_len && !positioned) ? nullptr : d_read_off.as<u64>();
    v.seg_read_off = d_seg_read_off.as<u64>();
    v.seg_tile_start = d_seg_tile_start.as<u32>();
    v.tile_info = d_tile_info.as<uint4>();
    v.fixed_len = fixed_len;
    v.n_segments = n_segments;
    return v;
}

void DevReads::release() { d_words.release(); d_read_off.release(); d_seg_read_off.release(); d_seg_tile_start.release(); d_tile_info.release(); }

// ---------------------------------------------------------------------------------------------------------------
// DevPaths
// ---------------------------------------------------------------------------------------------------------------
int DevPaths::upload_dirs(gasm_ctx* ctx) {
    GCHK(h2d(ctx, d_p_off, h_p_off.data(), h_p_off.size() * 8));
    GCHK(h2d(ctx, d_seg_path_off, h_seg_path_off.data(), h_seg_path_off.size() * 4));
    std::vector<u64>& sb = h_seg_base_off;
    sb.resize(n_segments + 1);
    for (u32 s = 0; s <= n_segments; ++s) sb[s] = h_p_off[h_seg_path_off[s]];
    GCHK(h2d(ctx, d_seg_base_off, sb.data(), sb.size() * 8));
    return GASM_OK;
}

int DevPaths::pack_from_device_ascii(gasm_ctx* ctx, const u8* d_ascii) {
    DBuf err;
    GCHK(err.ensure(8));
    HIPCHK(hipMemsetAsync(err.p, 0, 8, ctx->stream));
    int st = pack_ascii(ctx, d_ascii, total_bases, d_words, err.as<u32>());
    u32 herr = 0;
    if (st == GASM_OK && hipMemcpyAsync(&herr, err.p, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) st = GASM_ERR_HIP;
    if (st == GASM_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = GASM_ERR_HIP;
    err.release();
    if (st != GASM_OK) return st;
    if (herr) { gasm_set_error("paths contain a base outside upper-case ACGT"); return GASM_ERR_NON_ACGT; }
    return GASM_OK;
}

int DevPaths::upload_ascii(gasm_ctx* ctx, const char* data, const u64* off, u32 np) {
    n_segments = 1; n_paths = np;
    b_p_off = nullptr; b_seg_path_off = nullptr; b_seg_base_off = nullptr;
    h_p_off.resize((size_t)np + 1);
    for (u32 i = 0; i <= np; ++i) {
        if (i && off[i] < off[i - 1]) { gasm_set_error("path offsets not monotone"); return GASM_ERR_INVALID; }
        h_p_off[i] = off[i] - off[0];
    }
    total_bases = h_p_off[np];
    if (total_bases >= 0xFFFFFFF0ull) { gasm_set_error("paths exceed 2^32 bases"); return GASM_ERR_CAPACITY; }
    h_seg_path_off = {0u, np};
    DBuf ascii;
    GCHK(h2d(ctx, ascii, data + off[0], total_bases));
    int st = pack_from_device_ascii(ctx, ascii.as<u8>());
    ascii.release();
    GCHK(st);
    return upload_dirs(ctx);
}

PathSet DevPaths::view() const {
    PathSet v;
    v.words = d_words.as<u64>();
    v.p_off = b_p_off ? b_p_off : d_p_off.as<u64>();
    v.seg_path_off = b_seg_path_off ? b_seg_path_off : d_seg_path_off.as<u32>();
    v.n_segments = n_segments;
    return v;
}

void DevPaths::release() { d_words.release(); d_p_off.release(); d_seg_path_off.release(); d_seg_base_off.release(); }

void BuildState::release() {
    for (DBuf* b : {&d_keys, &d_keys2, &d_mult, &d_hist, &d_toff, &d_tcnt, &d_fdir, &d_bstart, &d_bucket_d, &d_dstart, &d_flags, &d_dk_key, &d_dk_cnt,
                    &d_eflag, &d_nxt, &d_link, &d_clen, &d_ecid, &d_ecoff, &d_rtab, &d_seg_cbases, &d_seg_cstart,
                    &d_seg_bstart, &d_c_off, &d_contig_ascii})
        b->release();
    part_valid = false;         // (d_bstart no longer holds the partition's region layout)
    if (h_report) { (void)hipHostFree(h_report); h_report = nullptr; h_report_words = 0; }
    if (ev_slot) { (void)hipEventDestroy(ev_slot); ev_slot = nullptr; }
    if (ev_dense) { (void)hipEventDestroy(ev_dense); ev_dense = nullptr; }
}

void ScoreState::release() {
    for (DBuf* b : {&d_tbl_off, &d_seed, &d_gpos, &d_poscnt, &d_total, &d_out_f64, &d_out_i32, &d_freq, &d_pd_off, &d_pd, &d_seg_empty, &d_fxsum, &d_first, &d_first_off}) b->release();
}

// ---------------------------------------------------------------------------------------------------------------
// build.  The host queues a whole build — k-mers -> buckets -> distinct k-mers -> graph -> contigs — without waiting for
// any size: arrays are allocated at upper bounds known from the reads (288 GB of HBM are there for this), grids come from
// estimates with the kernels looping over what is really there, and the last kernel writes every size the host will
// need, plus the failure flags, into the batch's pinned report.  pipeline_build_finish reads it when results are asked
// for; a build that tripped a flag (a bucket's table overflowed, list ranking's LDS estimate was too small) is repeated
// with the next larger configuration there.  So a step of a resident pipeline (build + score, fetch much later) never
// stalls the stream.
// ---------------------------------------------------------------------------------------------------------------
static int ensure_lds_attrs(gasm_ctx* ctx) {
    if (ctx->lds_attrs_set) return GASM_OK;         // per context = per device: the attribute is a property of the device's code object
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bucket_scatter<u64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bucket_scatter<K128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bucket_partition<u64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bucket_partition<K128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_hist<u64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_hist<K128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rank_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_score_reads_graph<u64>), hipFuncAttributeMaxDynamicSharedMemorySize, GASM_SCORE_PATH_CAP * 12));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_score_reads_graph<K128>), hipFuncAttributeMaxDynamicSharedMemorySize, GASM_SCORE_PATH_CAP * 12));
    ctx->lds_attrs_set = true;
    return GASM_OK;
}

// upper bound of the distinct k-mers of the segments, given the partition: a bucket holds at most `limit` (else it overflows)
void distinct_caps(BuildState& bs, u32 S) {
    const u64 limit = bs.multi_pass ? GASM_BUCKET_MAX : bs.small_tbl ? GASM_TBL_LIMIT / 2 : GASM_TBL_LIMIT;
    const u64 per_seg = ((u64)1 << bs.bbits) * limit;
    const u64 all = bs.k < 16 ? ((u64)1 << (2 * bs.k)) : ~(u64)0;     // 4^k different k-mers exist
    bs.D_cap = 0; bs.maxD_cap = 0;
    for (u32 s = 0; s < S; ++s) {
        const u64 c = std::min(std::min(bs.h_seg_nk[s], per_seg), all);
        bs.D_cap += c;
        bs.maxD_cap = std::max(bs.maxD_cap, c);
    }
}

static int launch_graph_dense(gasm_ctx* ctx, u32 S, BuildState& bs);

static GraphView graph_view(const BuildState& bs) {
    GraphView gv;
    gv.dk_key = bs.d_dk_key.p;
    gv.dstart = bs.d_dstart.as<u32>();
    gv.fdir = bs.d_fdir.as<u16>();
    gv.k = bs.k;
    gv.bbits = bs.bbits;
    gv.fbits = bs.fbits;
    return gv;
}

// ---- reads -> per-(segment, bucket) distinct k-mers with multiplicities (in place in d_keys / d_mult) + dstart
int launch_distinct(gasm_ctx* ctx, DevReads& rd, BuildState& bs) {
    GasmRange range("gasm:distinct (partition + de-duplication)");
    const int k = bs.k, W = bs.words, bbits = bs.bbits;
    const size_t KB = 8 * (size_t)W;
    const u32 S = rd.n_segments, nb = 1u << bbits, nbt = S * nb;
    const u64 N = bs.n_kmers;
    // every (tile, bucket) run is padded to a multiple of padm + 1 keys (a 128-byte line when the bucket count allows;
    // the padding of one tile must fit the flush passes k_bucket_scatter has beyond KT)
    const u32 line_keys = 128 / KB, pad_room = W == 1 ? 2 * GASM_TILE_WG : GASM_TILE_WG;
    u32 padm = line_keys - 1;
    if (knobs().padm >= 0) padm = std::min<u32>(padm, (u32)knobs().padm);
    while (padm && (u64)nb * padm > pad_room) padm >>= 1;
    const ReadSet rs = rd.view();
    const u32 g = bs.tile_g;
    const u32 grid_tiles = std::min<u32>(rd.n_tiles, (u32)ctx->n_cu * (u32)knobs().hist_wgs);
    constexpr u32 SCRATCH_KEYS = 1024 * (GASM_TILE_WG / 64) * 64 * 2;   // 16 bytes per lane and wave of 1024 workgroup slots (k_bucket_scatter)
    GCHK(bs.d_bstart.ensure(((size_t)nbt + 1) * 8));
    GCHK(bs.d_bucket_d.ensure((size_t)nbt * 4));
    GCHK(bs.d_dstart.ensure(((size_t)nbt + 2) * 4));
    u64 n_alloc;
    const u32* d_blen = nullptr;              // single pass: the buckets' padded lengths (the partition's cursors)
    const bool single = bs.single_pass && !bs.multi_pass && nb <= GASM_TILE_WG;
    if (single) {
        // ---- one pass (k_bucket_partition): a region of fixed capacity per (segment, bucket) — the segment's k-mers per
        // bucket with `part_slack` percent to spare, room for Poisson noise and for the padding of its tiles' runs, in whole
        // lines — and a cursor per region; cursors and the build's flag words are one allocation, zeroed by one fill
        const int slack = env_int("GASM_PART_SLACK", 100), forced = env_int("GASM_DBG_PART_CAP", 0);
        const bool same_layout = bs.part_valid && bs.part_reads_id == rd.upload_id && bs.part_k == k && bs.part_bbits == bbits && bs.part_padm == padm &&
                                 bs.part_g == g && bs.part_slack == slack && bs.part_forced == forced;
        if (!same_layout) {
            std::vector<u64> h((size_t)nbt + 1);
            u64 at = 0;
            for (u32 s = 0; s < S; ++s) {
                const u64 mean = ceil_div_u64(bs.h_seg_nk[s], nb), tiles = rd.h_seg_tile_start[s + 1] - rd.h_seg_tile_start[s];
                u64 cap = mean + (u64)((double)mean * std::max(slack, 0) / 100.0) + (u64)(8.0 * std::sqrt((double)mean)) + 32 + tiles * padm;
                if (forced > 0) cap = (u64)forced;
                cap = (cap + line_keys - 1) / line_keys * line_keys;
                for (u32 b = 0; b < nb; ++b) { h[(size_t)s * nb + b] = at; at += cap; }
            }
            h[nbt] = at;
            bs.part_alloc = at;
            GCHK(h2d(ctx, bs.d_bstart, h.data(), h.size() * 8));
            HIPCHK(hipStreamSynchronize(ctx->stream));          // (`h` goes out of scope; only when the batch shape changes)
            bs.part_valid = true; bs.part_reads_id = rd.upload_id; bs.part_k = k; bs.part_bbits = bbits; bs.part_padm = padm; bs.part_g = g;
            bs.part_slack = slack; bs.part_forced = forced;
        }
        n_alloc = bs.part_alloc;
        GCHK(bs.d_keys.ensure((n_alloc + SCRATCH_KEYS) * KB));
        GCHK(bs.d_mult.ensure(n_alloc * 4));
        GCHK(bs.d_flags.ensure(256 + (size_t)nbt * 4));
        HIPCHK(hipMemsetAsync(bs.d_flags.p, 0, 256 + (size_t)nbt * 4, ctx->stream));
        u32* const d_cursor = bs.d_flags.as<u32>() + 64;
        d_blen = d_cursor;
        const size_t lds = (size_t)(W == 1 ? 18 : 9) * GASM_TILE_WG * KB + (GASM_TILE_WG / 8) * KB + (size_t)((nb + 1) & ~1u) * 8 + (size_t)(4 * nb + 4) * 8 + 48 + (size_t)nb * 4;
        const u32 grid_part = std::min<u32>(rd.n_tiles, (u32)ctx->n_cu * (u32)knobs().scatter_wgs);
        if (W == 1) {
            GLAUNCH(ctx, "k_bucket_partition", k_bucket_partition<u64>, dim3(grid_part), dim3(GASM_TILE_WG), lds, rs, rs.tile_info, k, bbits, g, padm, rd.n_tiles,
                    bs.d_bstart.as<u64>(), d_cursor, bs.d_keys.as<u64>(), n_alloc, bs.d_flags.as<u32>());
        } else {
            GLAUNCH(ctx, "k_bucket_partition", k_bucket_partition<K128>, dim3(grid_part), dim3(GASM_TILE_WG), lds, rs, rs.tile_info, k, bbits, g, padm, rd.n_tiles,
                    bs.d_bstart.as<u64>(), d_cursor, bs.d_keys.as<K128>(), n_alloc, bs.d_flags.as<u32>());
        }
    } else {
        // ---- two passes: count, scan, scatter (exact layout; the retry path of the single pass, > 512 buckets, multi-pass builds)
        bs.part_valid = false;                // (k_scan_excl overwrites d_bstart)
        GCHK(bs.d_hist.ensure((size_t)nbt * 4));
        GCHK(bs.d_toff.ensure((size_t)rd.n_tiles * nb * 4));
        n_alloc = N + (u64)padm * rd.n_tiles * nb;
        GCHK(bs.d_keys.ensure((n_alloc + SCRATCH_KEYS) * KB));
        GCHK(bs.d_mult.ensure(n_alloc * 4));
        GCHK(bs.d_tcnt.ensure((size_t)rd.n_tiles * nb * 8 + 64));      // four 16-bit sub-counts per (tile, bucket)
        // (d_flags is zeroed by k_tile_scan)
        if (W == 1) {
            GLAUNCH(ctx, "k_tile_hist", k_tile_hist<u64>, dim3(grid_tiles), dim3(GASM_TILE_WG), (size_t)nb * 16, rs, rs.tile_info, k, bbits, g, rd.n_tiles,
                    bs.d_tcnt.as<ushort4>());
        } else {
            GLAUNCH(ctx, "k_tile_hist", k_tile_hist<K128>, dim3(grid_tiles), dim3(GASM_TILE_WG), (size_t)nb * 16, rs, rs.tile_info, k, bbits, g, rd.n_tiles,
                    bs.d_tcnt.as<ushort4>());
        }
        GLAUNCH(ctx, "k_tile_scan", k_tile_scan, dim3(S, std::max(1u, nb / 32u)), dim3(1024), 0, rs, bbits, padm, bs.d_tcnt.as<ushort4>(),
                bs.d_toff.as<u32>(), bs.d_hist.as<u32>(), bs.d_flags.as<u32>());
        GLAUNCH(ctx, "k_scan_excl", k_scan_excl<u64>, dim3(1), dim3(1024), 0, bs.d_hist.as<u32>(), bs.d_bstart.as<u64>(), nbt);
        const size_t lds = (size_t)(W == 1 ? 18 : 9) * GASM_TILE_WG * KB + (GASM_TILE_WG / 8) * KB + (size_t)nb * 24 + 96;   // KeyTraits<K>::NFL passes + trash slots + cursors
        // two workgroups per CU fit (LDS); a few tiles per workgroup so that the prefetch of the next tile pays
        const u32 grid_scatter = std::min<u32>(rd.n_tiles, (u32)ctx->n_cu * (u32)knobs().scatter_wgs);
        if (W == 1) {
            GLAUNCH(ctx, "k_bucket_scatter", k_bucket_scatter<u64>, dim3(grid_scatter), dim3(GASM_TILE_WG), lds, rs, rs.tile_info, k, bbits, g, padm, rd.n_tiles,
                    bs.d_bstart.as<u64>(), bs.d_toff.as<u32>(), bs.d_tcnt.as<ushort4>(), bs.d_keys.as<u64>(), n_alloc);
        } else {
            GLAUNCH(ctx, "k_bucket_scatter", k_bucket_scatter<K128>, dim3(grid_scatter), dim3(GASM_TILE_WG), lds, rs, rs.tile_info, k, bbits, g, padm, rd.n_tiles,
                    bs.d_bstart.as<u64>(), bs.d_toff.as<u32>(), bs.d_tcnt.as<ushort4>(), bs.d_keys.as<K128>(), n_alloc);
        }
    }
    // (what still reads the directories and dense arrays the de-duplication and everything behind it rewrite)
    if (bs.ev_before_dedup) HIPCHK(hipStreamWaitEvent(ctx->stream, bs.ev_before_dedup, 0));
    unsigned long long* d_stamps = nullptr;
    static DBuf stamp_buf;
    if (knobs().stamps) {   // diagnostic: per-phase cycle totals of k_bucket_dedup to stderr
        int o1 = 0, o2 = 0, o3 = 0, o4 = 0;
        const size_t lds_stage = (size_t)(W == 1 ? 18 : 9) * GASM_TILE_WG * KB + (GASM_TILE_WG / 8) * KB;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&o1, k_bucket_dedup<u64, 2048>, GASM_WG, 0);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&o2, k_bucket_dedup<u64, 4096>, GASM_WG, 0);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&o3, k_bucket_scatter<u64>, GASM_TILE_WG, lds_stage + (size_t)nb * 24 + 96);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&o4, k_bucket_partition<u64>, GASM_TILE_WG, lds_stage + (size_t)nb * 40 + 96);
        fprintf(stderr, "[occupancy API] dedup<2048> %d  dedup<4096> %d  scatter %d  partition %d blocks/CU\n", o1, o2, o3, o4);
        GCHK(stamp_buf.ensure(64 + (size_t)nbt * 24));
        HIPCHK(hipMemsetAsync(stamp_buf.p, 0, 64 + (size_t)nbt * 24, ctx->stream));
        d_stamps = stamp_buf.as<unsigned long long>();
    }
    const int dbg_d = knobs().dbg_dedup | (knobs().dedup_warm << 2);
    // the buckets' distinct counts -> dstart by the de-duplication's last workgroup (a word of the build's zeroed flags counts
    // the finished ones); beyond 16 384 buckets one workgroup of 256 is too slow a scanner: k_scan_excl in a launch of its own
    const bool scan_in_dedup = !bs.multi_pass && nbt <= 16384 && env_int("GASM_SCAN_IN_DEDUP", 1) != 0;
    u32* const d_scan_out = scan_in_dedup ? bs.d_dstart.as<u32>() : nullptr;
    bs.fbits = (bs.small_tbl && !bs.multi_pass) ? 9 : 10;   // bins of the de-duplication kernel's counting sort = TBL / 4 (the multi-pass kernel: 4096 slots)
    GCHK(bs.d_fdir.ensure((size_t)nbt * ((1u << bs.fbits) + 1) * 2));
    // (k_bucket_dedup writes every entry of its bucket's fine directory)
    if (bs.multi_pass) {
        // passes over key sub-ranges, for buckets no table can hold; they re-read the bucket, so the result goes to a second
        // array and is copied back
        GCHK(bs.d_keys2.ensure(n_alloc * KB));
        if (W == 2) GLAUNCH(ctx, "k_bucket_dedup_multi", k_bucket_dedup_multi<K128>, dim3(nbt), dim3(GASM_WG), 0, bs.d_keys.as<K128>(), bs.d_keys2.as<K128>(),
                            bs.d_mult.as<u32>(), bs.d_bstart.as<u64>(), bs.d_bucket_d.as<u32>(), bs.d_flags.as<u32>(), bs.d_fdir.as<u16>(), 2 * k - bbits);
        else GLAUNCH(ctx, "k_bucket_dedup_multi", k_bucket_dedup_multi<u64>, dim3(nbt), dim3(GASM_WG), 0, bs.d_keys.as<u64>(), bs.d_keys2.as<u64>(),
                     bs.d_mult.as<u32>(), bs.d_bstart.as<u64>(), bs.d_bucket_d.as<u32>(), bs.d_flags.as<u32>(), bs.d_fdir.as<u16>(), 2 * k - bbits);
        HIPCHK(hipMemcpyAsync(bs.d_keys.p, bs.d_keys2.p, n_alloc * KB, hipMemcpyDeviceToDevice, ctx->stream));
    } else if (W == 2) {
        GLAUNCH(ctx, "k_bucket_dedup", (k_bucket_dedup<K128, 2048>), dim3(nbt), dim3(GASM_WG), 0, bs.d_keys.as<K128>(), bs.d_mult.as<u32>(),
                bs.d_bstart.as<u64>(), d_blen, bs.d_bucket_d.as<u32>(), bs.d_flags.as<u32>(), bs.d_fdir.as<u16>(), 2 * k - bbits, dbg_d, d_stamps, d_scan_out);
    } else if (bs.small_tbl) {
        GLAUNCH(ctx, "k_bucket_dedup", (k_bucket_dedup<u64, 2048>), dim3(nbt), dim3(GASM_WG), 0, bs.d_keys.as<u64>(), bs.d_mult.as<u32>(),
                bs.d_bstart.as<u64>(), d_blen, bs.d_bucket_d.as<u32>(), bs.d_flags.as<u32>(), bs.d_fdir.as<u16>(), 2 * k - bbits, dbg_d, d_stamps, d_scan_out);
    } else {
        GLAUNCH(ctx, "k_bucket_dedup", (k_bucket_dedup<u64, 4096>), dim3(nbt), dim3(GASM_WG), 0, bs.d_keys.as<u64>(), bs.d_mult.as<u32>(),
                bs.d_bstart.as<u64>(), d_blen, bs.d_bucket_d.as<u32>(), bs.d_flags.as<u32>(), bs.d_fdir.as<u16>(), 2 * k - bbits, dbg_d, d_stamps, d_scan_out);
    }
    if (d_stamps) {
        std::vector<unsigned long long> hv(8 + (size_t)nbt * 3);
        HIPCHK(hipMemcpyAsync(hv.data(), d_stamps, hv.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        const unsigned long long* h = hv.data();
        if (FILE* f = fopen(knobs().stamps, "wb")) { fwrite(hv.data(), 8, hv.size(), f); fclose(f); }
        fprintf(stderr, "[dedup stamps, 100 MHz ticks per workgroup] init %.1f  first-iter %.1f  stream %.1f  barrier %.1f  order %.1f  writeback %.1f\n",
                (double)h[0] / nbt, (double)h[1] / nbt, (double)h[2] / nbt, (double)h[3] / nbt, (double)h[4] / nbt, (double)h[5] / nbt);
    }
    if (!scan_in_dedup) GLAUNCH(ctx, "k_scan_excl", k_scan_excl<u32>, dim3(1), dim3(1024), 0, bs.d_bucket_d.as<u32>(), bs.d_dstart.as<u32>(), nbt);
    return GASM_OK;
}

// ---- the per-edge arrays at their upper bound (the number of distinct k-mers is not known on the host)
static int alloc_graph(BuildState& bs, u32 S) {
    const u64 D = std::max<u64>(bs.D_cap, 1);
    const size_t KB = 8 * (size_t)bs.words;
    if (D >= 0xFFFFFFF0ull) { gasm_set_error("more than 2^32 distinct k-mers possible in one batch: split it"); return GASM_ERR_CAPACITY; }
    GCHK(bs.d_dk_key.ensure(D * KB));
    GCHK(bs.d_dk_cnt.ensure(D * 4));
    GCHK(bs.d_eflag.ensure(D));
    GCHK(bs.d_nxt.ensure(D * 4));
    GCHK(bs.d_link.ensure(D * 8));
    GCHK(bs.d_clen.ensure(D * 4));
    GCHK(bs.d_ecid.ensure(D * 4));
    GCHK(bs.d_ecoff.ensure(D * 8));
    GCHK(bs.d_rtab.ensure((D / 2 + S + 2) * 4));
    GCHK(bs.d_seg_cbases.ensure((size_t)S * 8 + (size_t)S * 4));   // u64 bases[S] then u32 counts[S]
    GCHK(bs.d_seg_cstart.ensure(((size_t)S + 1) * 4));
    GCHK(bs.d_seg_bstart.ensure(((size_t)S + 1) * 8));
    // contigs: at most one per edge, k-1 bases of the first node + one base per edge
    bs.bases_cap = D * (u64)bs.k;
    if (bs.bases_cap >= 0xFFFFFFF0ull) { gasm_set_error("contigs could exceed 2^32 bases: split the batch"); return GASM_ERR_CAPACITY; }
    GCHK(bs.d_c_off.ensure((D + 1) * 8));
    GCHK(bs.d_contig_ascii.ensure(bs.bases_cap + 64));
    return GASM_OK;
}

// ---- dense arrays -> (k-1)-mer graph -> chains -> contigs + the report.  Inputs: d_keys/d_mult/d_bstart (the buckets'
// distinct runs) and d_dstart/d_fdir.
int launch_graph(gasm_ctx* ctx, u32 S, BuildState& bs) {
    const int W = bs.words, bbits = bs.bbits;
    const u32 nb = 1u << bbits, nbt = S * nb;
    GCHK(alloc_graph(bs, S));
    // (k_bucket_gather zeroes the ranking's failure flag and the "still active" words of the k_link_jump launches; the
    // overflow flag [0] stays)
    u32* const d_claim = bs.d_nxt.as<u32>();      // claim words of the degree kernels live in nxt until k_edge_next overwrites them
    if (W == 1) {
        GLAUNCH(ctx, "k_bucket_gather", k_bucket_gather<u64>, dim3(nbt), dim3(GASM_WG), 0, bs.d_keys.as<u64>(), bs.d_mult.as<u32>(),
                bs.d_bstart.as<u64>(), bs.d_dstart.as<u32>(), bs.d_dk_key.as<u64>(), bs.d_dk_cnt.as<u32>(), d_claim, bs.d_eflag.as<u8>(), bs.d_flags.as<u32>());
    } else {
        GLAUNCH(ctx, "k_bucket_gather", k_bucket_gather<K128>, dim3(nbt), dim3(GASM_WG), 0, bs.d_keys.as<K128>(), bs.d_mult.as<u32>(),
                bs.d_bstart.as<u64>(), bs.d_dstart.as<u32>(), bs.d_dk_key.as<K128>(), bs.d_dk_cnt.as<u32>(), d_claim, bs.d_eflag.as<u8>(), bs.d_flags.as<u32>());
    }
    return launch_graph_dense(ctx, S, bs);
}

static int launch_graph_dense(gasm_ctx* ctx, u32 S, BuildState& bs) {
    GasmRange range("gasm:graph (degrees, list ranking, contigs)");
    const int W = bs.words;
    u32* const d_claim = bs.d_nxt.as<u32>();
    u32* const d_seg_ncontig = reinterpret_cast<u32*>(bs.d_seg_cbases.as<u64>() + S);
    u32* const d_fl = bs.d_flags.as<u32>();
    const GraphView gv = graph_view(bs);
    const u32 nb = 1u << bs.bbits;
    const u32 est = std::max<u32>(1, (u32)std::min<u64>(bs.maxD_est, bs.maxD_cap));   // grids: the kernels loop when a segment is larger
    const u32 dchunks = (u32)ceil_div_u64(est, GASM_WG);     // workgroups per segment of the per-edge kernels
    const dim3 grid_seg = seg_grid(dchunks, S);
    const u32 gchunks = (u32)ceil_div_u64(est, GASM_WG * GASM_EDGE_ILP);     // the kernels that take GASM_EDGE_ILP edges per thread and round
    const dim3 grid_grp = seg_grid(gchunks, S);
    u32* const d_tgt = bs.d_ecid.as<u32>();        // first out-edge of every edge's target node; e_cid is written later (k_contig_scan)
    if (W == 1) GLAUNCH(ctx, "k_edge_target", k_edge_target<u64>, grid_seg, dim3(GASM_WG), 0, gv, S, dchunks, d_tgt, d_claim);
    else GLAUNCH(ctx, "k_edge_target", k_edge_target<K128>, grid_seg, dim3(GASM_WG), 0, gv, S, dchunks, d_tgt, d_claim);
    GLAUNCH(ctx, "k_edge_multi", k_edge_multi, grid_grp, dim3(GASM_WG), 0, gv, S, gchunks, d_tgt, d_claim, bs.d_eflag.as<u8>());
    // (links: k_edge_next is their only writer — every edge's — before the ranking reads them; chain lengths: k_link_jump / k_chain_len)
    if (W == 1) GLAUNCH(ctx, "k_node_flags", k_node_flags<u64>, grid_seg, dim3(GASM_WG), 0, gv, S, dchunks, d_claim, bs.d_eflag.as<u8>(), bs.d_link.as<u64>(), bs.d_clen.as<u32>());
    else GLAUNCH(ctx, "k_node_flags", k_node_flags<K128>, grid_seg, dim3(GASM_WG), 0, gv, S, dchunks, d_claim, bs.d_eflag.as<u8>(), bs.d_link.as<u64>(), bs.d_clen.as<u32>());
    GLAUNCH(ctx, "k_edge_next", k_edge_next, grid_grp, dim3(GASM_WG), 0, gv, S, gchunks, d_tgt, bs.d_eflag.as<u8>(), bs.d_nxt.as<u32>(), bs.d_link.as<u64>());
    u32* const act = d_fl + 16;     // "still active" words of the k_link_jump launches
    const u32 jchunks = (u32)ceil_div_u64(est, GASM_WG * 4);      // GASM_JUMP_ILP links per thread
    bs.ranked_in_lds = est <= 65534 && !bs.rank_global && !knobs().rank_global;
    if (bs.ranked_in_lds) {
        // every second edge (the rulers) is ranked inside LDS, the others then need a step or two (kernels_build.hip)
        // rulers: every 2nd edge when every CU has a segment of its own to rank, every 4th when segments are few
        const u32 rshift = knobs().ruler_shift ? (u32)knobs().ruler_shift : (S >= (u32)ctx->n_cu / 4 ? 1u : 2u);
        const u32 rchunks = (u32)ceil_div_u64((est + (1u << rshift) - 1) >> rshift, GASM_WG);
        // LDS list of a segment: everything a workgroup can have while a CU only ever holds one of them, else the
        // estimate with a quarter to spare (a segment that does not fit raises flags[1]: pipeline_build_finish)
        u32 lds_entries = 32767;
        if (S > (u32)ctx->n_cu) lds_entries = std::min<u32>(32767, (((est + est / 4) >> rshift) + 1024) & ~1023u);
        GLAUNCH(ctx, "k_rank_rulers", k_rank_rulers, seg_grid(rchunks, S), dim3(GASM_WG), 0, gv, S, rchunks, bs.d_link.as<u64>(), bs.d_rtab.as<u32>(), rshift, d_fl);
        GLAUNCH(ctx, "k_rank_lds", k_rank_lds, dim3(S), dim3(1024), (size_t)lds_entries * 4 + 4, gv, bs.d_rtab.as<u32>(), bs.d_link.as<u64>(), knobs().rank_rounds, rshift,
                lds_entries, d_fl);
        // the odd edges: the ruler behind an edge is usually one or two steps away (a longer gap is geometrically rare)
        // (a thread stops as soon as its link is final; spans grow by a factor of jumps + 1 per launch at the very
        // least, so two launches cover any segment of this size, and the second normally returns at once)
        const int jumps = 255, launches = 2;
        for (int r = 0; r < launches; ++r)
            GLAUNCH(ctx, "k_link_jump", k_link_jump, seg_grid(jchunks, S), dim3(GASM_WG), 0, gv, S, jchunks, bs.d_link.as<u64>(), r ? act + r - 1 : nullptr, act + r, jumps,
                    bs.d_nxt.as<u32>(), bs.d_clen.as<u32>());         // (+ the chains' lengths: k_chain_len's work)
    } else {
        // whole-GPU launches of `jumps` doubling steps each; a launch returns at once when its predecessor found every chain done
        // (spans grow by at least jumps + 1 = 5 per launch: log2(5) > 2.3 rounds' worth)
        int rounds = 1;
        while ((1ull << rounds) < bs.maxD_cap) ++rounds;
        rounds += 1;
        const int jumps = 4, launches = (rounds * 10 + 22) / 23 + 1;
        if (launches > 40) { gasm_set_error("segment too large for the list-ranking flags"); return GASM_ERR_CAPACITY; }
        for (int r = 0; r < launches; ++r)
            GLAUNCH(ctx, "k_link_jump", k_link_jump, seg_grid(jchunks, S), dim3(GASM_WG), 0, gv, S, jchunks, bs.d_link.as<u64>(), r ? act + r - 1 : nullptr, act + r, jumps,
                    (const u32*)nullptr, (u32*)nullptr);
    }
    const u32 grid_all = (u32)std::min<u64>(ceil_div_u64((u64)est * S, GASM_WG), (u64)ctx->n_cu * 64);
    if (!bs.ranked_in_lds)      // (the LDS path's k_link_jump launches publish the chains' lengths themselves)
        GLAUNCH(ctx, "k_chain_len", k_chain_len, dim3(std::max(1u, grid_all)), dim3(GASM_WG), 0, bs.d_nxt.as<u32>(), bs.d_link.as<u64>(),
                bs.d_clen.as<u32>(), gv.dstart + (size_t)S * nb);
    // segment directories of the contigs + the report (ticket last): written by the last workgroup of k_contig_scan to finish
    const size_t words = 4 * (size_t)S + 8;
    if (bs.h_report_words < words) {
        if (bs.h_report) (void)hipHostFree(bs.h_report);
        bs.h_report = nullptr; bs.h_report_words = 0;
        HIPCHK(hipHostMalloc((void**)&bs.h_report, words * 4, hipHostMallocCoherent));
        bs.h_report_words = words;
    }
    bs.ticket = next_ticket();
    reinterpret_cast<volatile u32*>(bs.h_report)[4 * (size_t)S + 6] = 0;     // (the report of the previous build has been read or is void)
    GLAUNCH(ctx, "k_contig_scan", k_contig_scan, dim3(S), dim3(1024), 0, gv, bs.d_eflag.as<u8>(), bs.d_clen.as<u32>(),
            bs.d_ecid.as<u32>(), bs.d_ecoff.as<u64>(), d_seg_ncontig, bs.d_seg_cbases.as<u64>(), d_fl + 9, bs.d_seg_cstart.as<u32>(),
            bs.d_seg_bstart.as<u64>(), d_fl, bs.h_report, bs.ticket);
    GLAUNCH(ctx, "k_contig_place", k_contig_place, grid_seg, dim3(GASM_WG), 0, gv, bs.d_eflag.as<u8>(),
            bs.d_seg_cstart.as<u32>(), bs.d_seg_bstart.as<u64>(), bs.d_ecid.as<u32>(), bs.d_ecoff.as<u64>(), bs.d_c_off.as<u64>(), S, dchunks);
    if (W == 1) {
        GLAUNCH(ctx, "k_contig_emit", k_contig_emit<u64>, grid_grp, dim3(GASM_WG), 0, gv, bs.d_link.as<u64>(), bs.d_nxt.as<u32>(),
                bs.d_ecoff.as<u64>(), bs.d_contig_ascii.as<u8>(), S, gchunks);
    } else {
        GLAUNCH(ctx, "k_contig_emit", k_contig_emit<K128>, grid_grp, dim3(GASM_WG), 0, gv, bs.d_link.as<u64>(), bs.d_nxt.as<u32>(),
                bs.d_ecoff.as<u64>(), bs.d_contig_ascii.as<u8>(), S, gchunks);
    }
    bs.pending = true;
    return GASM_OK;
}

// host-side sizes of a batch's reads for key width / tile shape / bucket bits
int plan_build(gasm_ctx* ctx, DevReads& rd, int k, u64 hint, BuildState& bs) {
    if (k < 2 || k > GASM_MAX_K) { gasm_set_error("k = %d not supported (2..%d)", k, GASM_MAX_K); return GASM_ERR_INVALID; }
    const int W = k <= 31 ? 1 : 2;            // 64-bit keys up to k = 31, 128-bit keys (K128) up to k = 63
    const u32 KT = W == 1 ? 16u : 8u;         // k-mers per thread and round of the tile kernels (KeyTraits<K>::KT)
    HIPCHK(hipSetDevice(ctx->device));
    GCHK(ensure_lds_attrs(ctx));
    const u32 S = rd.n_segments;
    const bool same = bs.k == k && bs.words == W && bs.h_seg_nk.size() == S && bs.hint == hint && bs.reads_id == rd.upload_id;
    bs.words = W; bs.k = k; bs.hint = hint; bs.reads_id = rd.upload_id;
    bs.fetched_distinct = bs.fetched_contigs = false;
    bs.pending = false;
    // ---- sizes known on the host
    u64 N = 0, maxNs = 0;
    bs.h_seg_nk.assign(S, 0);
    if (rd.fixed_len) {
        const u64 nk = rd.fixed_len >= (u32)k ? rd.fixed_len - k + 1 : 0;
        for (u32 s = 0; s < S; ++s) bs.h_seg_nk[s] = (rd.h_seg_read_off[s + 1] - rd.h_seg_read_off[s]) * nk;
    } else {
        u32 s = 0;
        for (u64 r = 0; r < rd.n_reads; ++r) {
            while (r >= rd.h_seg_read_off[s + 1]) ++s;
            const u64 L = rd.h_read_off[r + 1] - rd.h_read_off[r];
            if (L >= (u64)k) bs.h_seg_nk[s] += L - k + 1;
        }
    }
    for (u64 v : bs.h_seg_nk) { N += v; maxNs = std::max(maxNs, v); }
    bs.n_kmers = N;
    const u32 nk_max = rd.max_len >= (u32)k ? rd.max_len - k + 1 : 0;
    u32 g = next_pow2_u32((nk_max + KT - 1) / KT);     // threads per read (a power of two: divides the workgroup)
    g = std::max(1u, std::min((u32)GASM_TILE_WG, g));
    bs.tile_g = g;
    const u32 ipt = GASM_TILE_WG / g;
    // offset rounds: more than one only for reads with more than g*KT k-mers; every (read group, round) is a tile
    const u32 orr = std::max(1u, (nk_max + g * KT - 1) / (g * KT));
    if (orr > 0xFFFFu) { gasm_set_error("reads longer than %u bases are not supported", GASM_TILE_WG * KT * 0xFFFFu); return GASM_ERR_CAPACITY; }
    GCHK(rd.set_tiles(ctx, ipt, orr));
    // bucket bits: aim at <= ~900 distinct k-mers per bucket (2048-slot LDS table in two-slot sets, limit 1408)
    bs.bb_cap = std::min(10, 2 * (k - 1));
    const u64 dest = hint ? hint : std::max<u64>(1, maxNs / 8);
    if (!same || !bs.have_actual) {
        int bbits = 0;
        while (bbits < bs.bb_cap && (dest >> bbits) > 900) ++bbits;
        // few segments: up to two more bits so that there are enough (segment, bucket) workgroups to fill the chip
        for (int extra = 0; extra < 2 && bbits < bs.bb_cap && ((u64)S << bbits) < 1024; ++extra) ++bbits;
        bbits = std::min(bs.bb_cap, bbits + knobs().bbits_add);
        bs.bbits = bbits;
        // small (2048-slot) LDS tables when the expected number of distinct k-mers per bucket is small: more workgroups per CU
        bs.small_tbl = (dest >> bbits) <= 900;   // else the 4096-slot table (fewer workgroups per CU)
        if (knobs().dedup_tbl) bs.small_tbl = knobs().dedup_tbl == 2048;
        if (W == 2) bs.small_tbl = true;         // 128-bit keys: 2048-slot tables only
        bs.maxD_est = (u32)std::min<u64>(dest, 0xFFFFFFF0ull);
        bs.have_actual = false;
        bs.rank_global = false;
        bs.multi_pass = false;
        bs.single_pass = env_int("GASM_SINGLE_PASS", 1) != 0;
    }   // else: the same reads again — the partition that worked and the sizes the last build reported
    GCHK(bs.d_flags.ensure(256));      // [0] bucket overflow, [1] list ranking gave up, [16..] the list-ranking launches' "still active" words
    bs.d_total = 0; bs.n_contigs = 0; bs.contig_bases = 0;
    return GASM_OK;
}

static void zero_results(BuildState& bs, u32 S) {
    bs.h_dstart.assign((size_t)S + 1, 0);
    bs.h_seg_cstart.assign((size_t)S + 1, 0);
    bs.h_seg_bstart.assign((size_t)S + 1, 0);
    bs.d_total = 0; bs.n_contigs = 0; bs.contig_bases = 0;
}

int pipeline_build(gasm_ctx* ctx, DevReads& rd, int k, u64 hint, BuildState& bs) {
    GCHK(plan_build(ctx, rd, k, hint, bs));
    const u32 S = rd.n_segments;
    zero_results(bs, S);
    if (bs.n_kmers == 0) {
        // nothing to do on the device; the directories the scorer borrows exist and are zero
        bs.bbits = 0; bs.D_cap = 0; bs.maxD_cap = 0;
        GCHK(alloc_graph(bs, S));
        GCHK(bs.d_dstart.ensure(((size_t)S + 2) * 4));
        HIPCHK(hipMemsetAsync(bs.d_dstart.p, 0, ((size_t)S + 2) * 4, ctx->stream));
        HIPCHK(hipMemsetAsync(bs.d_seg_cstart.p, 0, ((size_t)S + 1) * 4, ctx->stream));
        HIPCHK(hipMemsetAsync(bs.d_seg_bstart.p, 0, ((size_t)S + 1) * 8, ctx->stream));
        HIPCHK(hipMemsetAsync(bs.d_c_off.p, 0, 8, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        return GASM_OK;
    }
    distinct_caps(bs, S);
    gasm_ctx* const sx = bs.stream_ctx && bs.stream_ctx != ctx ? bs.stream_ctx : ctx;
    if (sx != ctx) {
        if (!bs.ev_slot) HIPCHK(hipEventCreateWithFlags(&bs.ev_slot, hipEventDisableTiming));
        if (!bs.ev_dense) HIPCHK(hipEventCreateWithFlags(&bs.ev_dense, hipEventDisableTiming));
        HIPCHK(hipEventRecord(bs.ev_slot, ctx->stream));          // whatever this state's last build left queued (graph, scoring, fetches)
        HIPCHK(hipStreamWaitEvent(sx->stream, bs.ev_slot, 0));
    }
    if (bs.ev_wait) HIPCHK(hipStreamWaitEvent(sx->stream, bs.ev_wait, 0));
    GCHK(launch_distinct(sx, rd, bs));
    if (bs.ev_streamed) HIPCHK(hipEventRecord(bs.ev_streamed, sx->stream));
    if (sx != ctx) {
        HIPCHK(hipEventRecord(bs.ev_dense, sx->stream));
        HIPCHK(hipStreamWaitEvent(ctx->stream, bs.ev_dense, 0));
    }
    GCHK(launch_graph(ctx, S, bs));
    if (knobs().sync_build) GCHK(pipeline_build_finish(ctx, rd, bs, nullptr));
    return GASM_OK;
}

// Read the report of a queued build; repeat the build with the next larger configuration while it reports a failure.
int pipeline_build_finish(gasm_ctx* ctx, DevReads& rd, BuildState& bs, bool* rebuilt) {
    return pipeline_build_finish_n(ctx, &rd, rd.n_segments, bs, rebuilt);
}

// rd == nullptr: the dense arrays did not come from reads of this rank (pooled build): only the graph can be repeated
int pipeline_build_finish_n(gasm_ctx* ctx, DevReads* rd, u32 S, BuildState& bs, bool* rebuilt) {
    if (rebuilt) *rebuilt = false;
    if (!bs.pending) return GASM_OK;
    HIPCHK(hipSetDevice(ctx->device));
    for (;;) {
        const u32* const rep = bs.h_report;
        GCHK(wait_report(ctx, rep + 4 * (size_t)S + 6, bs.ticket));
        bs.pending = false;
        // flags[0]: bit 0 = a bucket overflowed its table, bit 1 = a bucket outgrew its region (single-pass partition)
        const bool part_overflow = (rep[4 * (size_t)S + 4] & 2u) != 0;
        const bool overflow = (rep[4 * (size_t)S + 4] & 1u) != 0 || part_overflow, rank_failed = rep[4 * (size_t)S + 5] != 0;
        if (!overflow && !rank_failed) break;
        if (rebuilt) *rebuilt = true;
        if (overflow) {
            if (!rd) { gasm_set_error("a merged k-mer bucket overflowed its table: the pooled build needs more bucket bits"); return GASM_ERR_CAPACITY; }
            if (part_overflow && bs.single_pass) bs.single_pass = false;  // same partition, exact layout (count, scan, scatter)
            else if (bs.small_tbl && bs.words == 1) bs.small_tbl = false;      // same partition, larger tables
            else if (bs.bbits < bs.bb_cap) bs.bbits = std::min(bs.bb_cap, bs.bbits + 2);
            else if (!bs.multi_pass) bs.multi_pass = true;                // all bucket bits used: key sub-ranges, pass by pass
            else {
                gasm_set_error("a k-mer bucket holds more than %d distinct k-mers even with %d bucket bits", GASM_BUCKET_MAX, bs.bbits);
                return GASM_ERR_CAPACITY;
            }
            distinct_caps(bs, S);
            GCHK(launch_distinct(ctx, *rd, bs));
        } else {
            bs.rank_global = true;        // whole-GPU pointer doubling instead of the LDS ranking
        }
        GCHK(launch_graph(ctx, S, bs));
    }
    const u32* const rep = bs.h_report;
    bs.h_dstart.assign(rep, rep + S + 1);
    u32 maxD = 0;
    for (u32 s = 0; s <= S; ++s) {
        bs.h_seg_cstart[s] = rep[S + 1 + s];
        bs.h_seg_bstart[s] = (u64)rep[2 * S + 2 + 2 * s] | ((u64)rep[2 * S + 3 + 2 * s] << 32);
        if (s < S) maxD = std::max(maxD, rep[s + 1] - rep[s]);
    }
    bs.d_total = bs.h_dstart[S];
    bs.n_contigs = bs.h_seg_cstart[S];
    bs.contig_bases = bs.h_seg_bstart[S];
    // what the next build of the same reads starts from: sizes instead of estimates
    bs.maxD_est = std::max(1u, maxD);
    bs.paths_est = 1;
    for (u32 s = 0; s < S; ++s) bs.paths_est = std::max(bs.paths_est, bs.h_seg_cstart[s + 1] - bs.h_seg_cstart[s]);
    bs.have_actual = true;
    return GASM_OK;
}

int pipeline_fetch_distinct(gasm_ctx* ctx, DevReads& rd, BuildState& bs) {
    GCHK(pipeline_build_finish(ctx, rd, bs, nullptr));
    if (bs.fetched_distinct) return GASM_OK;
    const u32 S = rd.n_segments;
    bs.h_seg_doff.resize((size_t)S + 1);
    for (u32 s = 0; s <= S; ++s) bs.h_seg_doff[s] = bs.h_dstart.empty() ? 0 : bs.h_dstart[s];
    bs.h_dk_key.resize((size_t)bs.d_total * bs.words);     // 128-bit keys come back as (hi, lo) pairs
    bs.h_dk_cnt.resize(bs.d_total);
    if (bs.d_total) {
        HIPCHK(hipMemcpyAsync(bs.h_dk_key.data(), bs.d_dk_key.p, (size_t)bs.d_total * 8 * bs.words, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(bs.h_dk_cnt.data(), bs.d_dk_cnt.p, (size_t)bs.d_total * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    bs.fetched_distinct = true;
    return GASM_OK;
}

// per-edge flags and successors of the finished build (rows A4-A6 of SURVEY §8 made visible: lib/DeNovoAssembler.cpp:125-189)
int pipeline_fetch_graph(gasm_ctx* ctx, DevReads& rd, BuildState& bs) {
    GCHK(pipeline_build_finish(ctx, rd, bs, nullptr));
    bs.h_eflag.resize(bs.d_total);
    bs.h_nxt.resize(bs.d_total);
    if (bs.d_total) {
        HIPCHK(hipMemcpyAsync(bs.h_eflag.data(), bs.d_eflag.p, (size_t)bs.d_total, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(bs.h_nxt.data(), bs.d_nxt.p, (size_t)bs.d_total * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return GASM_OK;
}

int pipeline_fetch_contigs(gasm_ctx* ctx, DevReads& rd, BuildState& bs) {
    GCHK(pipeline_build_finish(ctx, rd, bs, nullptr));
    if (bs.fetched_contigs) return GASM_OK;
    const u32 S = rd.n_segments;
    bs.h_seg_coff.resize((size_t)S + 1);
    for (u32 s = 0; s <= S; ++s) bs.h_seg_coff[s] = bs.h_seg_cstart.empty() ? 0 : bs.h_seg_cstart[s];
    bs.h_c_off.assign((size_t)bs.n_contigs + 1, 0);
    bs.h_contigs.resize(bs.contig_bases);
    if (bs.n_contigs) {
        HIPCHK(hipMemcpyAsync(bs.h_c_off.data(), bs.d_c_off.p, ((size_t)bs.n_contigs + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(bs.h_contigs.data(), bs.d_contig_ascii.p, bs.contig_bases, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    bs.fetched_contigs = true;
    return GASM_OK;
}

// The contigs of a build as paths: offsets and directories are the build's own device arrays (borrowed, no copy, no
// host round trip); only the 2-bit packing of the contig text is new.  Works on a build the host has not waited for:
// the number of bases is read on the device.
int pipeline_contig_paths(gasm_ctx* ctx, const DevReads& rd, const BuildState& bs, DevPaths& dp) {
    const u32 S = rd.n_segments;
    dp.n_segments = S;
    dp.n_paths = 0; dp.total_bases = 0;            // host-side numbers: pipeline_contig_paths_host, after the report
    dp.h_p_off.clear(); dp.h_seg_path_off.clear(); dp.h_seg_base_off.clear();
    dp.b_p_off = bs.d_c_off.as<u64>();
    dp.b_seg_path_off = bs.d_seg_cstart.as<u32>();
    dp.b_seg_base_off = bs.d_seg_bstart.as<u64>();
    const u64 cap_words = (bs.bases_cap + 31) / 32 + 4;
    GCHK(dp.d_words.ensure(cap_words * 8));
    const u32 grid = (u32)std::min<u64>(ceil_div_u64(std::max<u64>(1, (u64)bs.maxD_est * S / 32 + 4), GASM_WG), (u64)ctx->n_cu * 16);
    GLAUNCH(ctx, "k_pack_ascii", k_pack_ascii, dim3(std::max(1u, grid)), dim3(GASM_WG), 0, bs.d_contig_ascii.as<u8>(), (u64)0,
            bs.d_seg_bstart.as<u64>() + S, dp.d_words.as<u64>(), (u32*)nullptr);
    return GASM_OK;
}

void pipeline_contig_paths_host(const DevReads& rd, const BuildState& bs, DevPaths& dp) {
    dp.n_paths = bs.n_contigs;
    dp.total_bases = bs.contig_bases;
    dp.h_seg_path_off.assign(bs.h_seg_cstart.begin(), bs.h_seg_cstart.end());
    dp.h_seg_base_off.assign(bs.h_seg_bstart.begin(), bs.h_seg_bstart.end());
    if (dp.h_seg_path_off.size() != (size_t)rd.n_segments + 1) { dp.h_seg_path_off.assign((size_t)rd.n_segments + 1, 0); dp.h_seg_base_off.assign((size_t)rd.n_segments + 1, 0); }
}

// ---------------------------------------------------------------------------------------------------------------
// score tables
// ---------------------------------------------------------------------------------------------------------------
#define DIRECT_ROWS 87380
static inline u32 direct_base_h(u32 L) { return ((1u << (2 * L)) - 4u) / 3u; }

int ScoreTable::set(gasm_ctx* ctx, const char* bp_kmer, const u64* bp_off, u64 nt, const double* bp_prob) {
    if (nt > 0x7FFFFFFFull) { gasm_set_error("table too large"); return GASM_ERR_CAPACITY; }
    std::vector<double> prob(DIRECT_ROWS, 0.0);
    std::vector<int32_t> row(DIRECT_ROWS, -1);
    for (u64 i = 0; i < nt; ++i) {
        const u64 L = bp_off[i + 1] - bp_off[i];
        if (L < 1 || L > 8) { gasm_set_error("bp_kmer[%llu] has length %llu; supported: 1..8", (unsigned long long)i, (unsigned long long)L); return GASM_ERR_INVALID; }
        u32 v = 0;
        for (u64 j = 0; j < L; ++j) {
            const char c = bp_kmer[bp_off[i] + j];
            u32 code;
            switch (c) { case 'A': code = 0; break; case 'C': code = 1; break; case 'G': code = 2; break; case 'T': code = 3; break;
                default: gasm_set_error("bp_kmer[%llu] holds a base outside ACGT", (unsigned long long)i); return GASM_ERR_NON_ACGT; }
            v = (v << 2) | code;
        }
        const u32 idx = direct_base_h((u32)L) + v;
        if (row[idx] >= 0) { gasm_set_error("bp_kmer[%llu] repeats an earlier key", (unsigned long long)i); return GASM_ERR_INVALID; }
        row[idx] = (int32_t)i;
        prob[idx] = bp_prob[i];
    }
    n_table = (u32)nt;
    h_row_prob.assign(bp_prob, bp_prob + nt);
    GCHK(h2d(ctx, d_prob, prob.data(), prob.size() * 8));
    GCHK(h2d(ctx, d_row, row.data(), row.size() * 4));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return GASM_OK;
}

int ScoreTable::set_standard(gasm_ctx* ctx, const double* t) {
    std::vector<double> prob(DIRECT_ROWS, 0.0);
    std::vector<int32_t> row(DIRECT_ROWS, -1);
    u32 src = 0;
    for (u32 L = 2; L <= 8; L += 2) {
        const u32 n = 1u << (2 * L), b = direct_base_h(L);
        for (u32 v = 0; v < n; ++v) { prob[b + v] = t[src]; row[b + v] = (int32_t)src; ++src; }
    }
    n_table = GASM_TABLE_ROWS;
    GCHK(h2d(ctx, d_prob, prob.data(), prob.size() * 8));
    GCHK(h2d(ctx, d_row, row.data(), row.size() * 4));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    h_prob = prob;
    h_absmax = 0;
    for (double v : h_prob) h_absmax = std::max(h_absmax, std::fabs(v));
    fix_shift = -1;
    return GASM_OK;
}

// 64-bit fixed-point copy of the table for the batch scorer: round(prob * 2^shift), shift chosen so that the sum over
// `max_terms` reads cannot overflow 62 bits.
int ScoreTable::set_fixed(gasm_ctx* ctx, u64 max_terms) {
    const double mx = h_absmax;
    int shift = 62;
    if (mx > 0) {
        const double need = std::log2(mx * (double)std::max<u64>(1, max_terms));
        shift = (int)std::floor(62.0 - need) - 1;
    }
    shift = std::max(0, std::min(1000, shift));
    if (shift > 1000) shift = 1000;
    if (shift == fix_shift) return GASM_OK;
    std::vector<long long> fx(h_prob.size());
    for (size_t i = 0; i < fx.size(); ++i) fx[i] = std::llrint(std::ldexp(h_prob[i], shift));
    GCHK(h2d(ctx, d_fix, fx.data(), fx.size() * 8));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    fix_shift = shift;
    return GASM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Levenshtein distances
// ---------------------------------------------------------------------------------------------------------------
int pipeline_levenshtein(gasm_ctx* ctx, DevPaths& dp, const char* target, u64 target_len, bool infix, std::vector<int32_t>& lev, bool* done) {
    GasmRange range("gasm:levenshtein");
    *done = false;
    const u32 P = dp.n_paths;
    lev.assign(P, 0);
    if (P == 0 || target_len == 0) { *done = true; return GASM_OK; }      // (empty target: the reference returns 0)
    if (target_len >= 0xFFFFFF00ull) return GASM_OK;                       // host routine
    HIPCHK(hipSetDevice(ctx->device));
    DBuf ascii, err, twords, carry, d_out;
    struct Rel { std::vector<DBuf*> v; ~Rel() { for (DBuf* b : v) b->release(); } } rel{{&ascii, &err, &twords, &carry, &d_out}};
    GCHK(err.ensure(8));
    HIPCHK(hipMemsetAsync(err.p, 0, 8, ctx->stream));
    GCHK(h2d(ctx, ascii, target, target_len));
    int st = pack_ascii(ctx, ascii.as<u8>(), target_len, twords, err.as<u32>());
    u32 herr = 0;
    if (st == GASM_OK && hipMemcpyAsync(&herr, err.p, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) st = GASM_ERR_HIP;
    if (st == GASM_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = GASM_ERR_HIP;
    if (st != GASM_OK || herr) { ascii.release(); err.release(); twords.release(); return st; }   // not ACGT: host routine
    // a wave per path, in the order of the paths (scaffolds come longest first): the dispatcher hands the next workgroup to
    // the CU that has room, which balances better than a fixed share of paths per resident wave (22 053 scaffolds of one
    // segment: 329 -> 318 ms); what a wave parks for its next band is 16 bytes per 64 target columns, capped at 2 GB in all
    // (global distances may run with the roles of path and target swapped: the columns are then the path's)
    u64 max_cols = target_len;
    if (!infix) for (u32 p = 0; p < P; ++p) max_cols = std::max<u64>(max_cols, dp.h_p_off[p + 1] - dp.h_p_off[p]);
    const u64 per_wave = ((max_cols - 1) / 64 + 2) * 16;
    const u32 cap = (u32)std::max<u64>((u64)ctx->n_cu * 4u, std::min<u64>(1u << 20, (2ull << 30) / per_wave));      // (at least a wave per SIMD)
    const u32 waves = std::min<u32>(P, getenv("GASM_LEV_WAVES") ? (u32)ctx->n_cu * (u32)atoi(getenv("GASM_LEV_WAVES")) : cap);
    const u32 wgs = (waves + GASM_WG / 64 - 1) / (GASM_WG / 64);
    const bool v2 = env_int("GASM_LEV_V", 2) != 1;
    // what a band leaves for the next one, per wave: 16 bytes per 64 target columns (v2) / a byte per column
    const u64 stride = v2 ? ((max_cols - 1) / 64 + 2) : ((target_len + 64 + 63) & ~(u64)63);
    st = carry.ensure((size_t)wgs * (GASM_WG / 64) * stride * (v2 ? 16 : 1));
    if (st == GASM_OK) st = d_out.ensure((size_t)P * 4);
    if (st == GASM_OK) {
        if (v2) hipLaunchKernelGGL(k_levenshtein2, dim3(wgs), dim3(GASM_WG), 0, ctx->stream, dp.view(), P, twords.as<u64>(), (u32)target_len, infix ? 1 : 0,
                                   carry.as<uint4>(), stride, d_out.as<int32_t>());
        else hipLaunchKernelGGL(k_levenshtein, dim3(wgs), dim3(GASM_WG), 0, ctx->stream, dp.view(), P, twords.as<u64>(), (u32)target_len, infix ? 1 : 0,
                                carry.as<u8>(), stride, d_out.as<int32_t>());
        if (hipGetLastError() != hipSuccess) st = GASM_ERR_HIP;
    }
    if (st == GASM_OK && hipMemcpyAsync(lev.data(), d_out.p, (size_t)P * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) st = GASM_ERR_HIP;
    if (st == GASM_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) { gasm_set_error("k_levenshtein failed"); st = GASM_ERR_HIP; }
    ascii.release(); err.release(); twords.release(); carry.release(); d_out.release();
    if (st == GASM_OK) *done = true;
    return st;
}

void ScoreTable::release() { d_prob.release(); d_row.release(); d_fix.release(); }

// ---------------------------------------------------------------------------------------------------------------
// F4: KS statistic and coverage (kernels_score.hip)
// ---------------------------------------------------------------------------------------------------------------
int pipeline_ks(gasm_ctx* ctx, DevPaths& dp, ScoreState& ss, const ScoreTable& tb, const char* genome, u64 genome_len, int kmer, std::vector<double>& ks) {
    const u32 P = dp.n_paths, NT = tb.n_table;
    ks.assign(P, std::nan(""));
    if (P == 0) return GASM_OK;
    if (tb.h_row_prob.size() != NT) { gasm_set_error("the KS statistic needs the table rows (ScoreTable::set)"); return GASM_ERR_STATE; }
    HIPCHK(hipSetDevice(ctx->device));
    DBuf ascii, err, gwords, hist, d_pv, d_cumy, scratch, d_out;
    struct Rel { std::vector<DBuf*> v; ~Rel() { for (DBuf* b : v) b->release(); } } rel{{&ascii, &err, &gwords, &hist, &d_pv, &d_cumy, &scratch, &d_out}};
    // ---- the genome's side: rows of its kmer-long windows, counted, in ascending-probability order, prefix-summed
    GCHK(err.ensure(8));
    HIPCHK(hipMemsetAsync(err.p, 0, 8, ctx->stream));
    GCHK(h2d(ctx, ascii, genome, genome_len));
    GCHK(pack_ascii(ctx, ascii.as<u8>(), genome_len, gwords, err.as<u32>()));
    GCHK(hist.ensure((size_t)std::max<u32>(NT, 1) * 4));
    HIPCHK(hipMemsetAsync(hist.p, 0, (size_t)std::max<u32>(NT, 1) * 4, ctx->stream));
    GLAUNCH(ctx, "k_ks_genome_hist", k_ks_genome_hist, dim3(std::max(1u, std::min<u32>(ceil_div_u64(genome_len + 1, GASM_WG), (u32)ctx->n_cu * 8u))), dim3(GASM_WG), 0,
            gwords.as<u64>(), genome_len, kmer, tb.d_row.as<int32_t>(), hist.as<u32>());
    std::vector<u32> h_hist(NT);
    u32 herr = 0;
    HIPCHK(hipMemcpyAsync(&herr, err.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    if (NT) HIPCHK(hipMemcpyAsync(h_hist.data(), hist.p, (size_t)NT * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (herr) { gasm_set_error("true_solution holds a base outside upper-case ACGT: the KS statistic needs its window probabilities"); return GASM_ERR_NON_ACGT; }
    std::vector<u32> order(NT);
    for (u32 i = 0; i < NT; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](u32 a, u32 b) { return tb.h_row_prob[a] < tb.h_row_prob[b]; });
    std::vector<double> pv(NT);
    std::vector<u32> cumy(NT);
    u64 run = 0;
    for (u32 r = 0; r < NT; ++r) { pv[r] = tb.h_row_prob[order[r]]; run += h_hist[order[r]]; cumy[r] = (u32)run; }
    GCHK(h2d(ctx, d_pv, pv.data(), (size_t)NT * 8));
    GCHK(h2d(ctx, d_cumy, cumy.data(), (size_t)NT * 4));
    // ---- the paths' side: k_path_ks2 (histogram over the counts' values); a path with a count beyond its bins goes to k_path_ks
    const u32 bins = (u32)std::max(2, std::min(1024, env_int("GASM_DBG_KS_BINS", 1024)));
    const bool v2 = env_int("GASM_KS_V", 2) != 1 && NT > 0;
    const u32 grid = std::min<u32>(P, (u32)ctx->n_cu * (v2 ? 4u : 2u));
    GCHK(scratch.ensure((size_t)grid * 2 * std::max<u32>(NT, 1) * 4));
    GCHK(d_out.ensure((size_t)P * 8));
    std::vector<u32> todo;
    if (v2) {
        DBuf d_run_end, d_flags;
        struct Rel2 { std::vector<DBuf*> v; ~Rel2() { for (DBuf* b : v) b->release(); } } rel2{{&d_run_end, &d_flags}};
        std::vector<u32> run_end(NT);
        for (u32 r = NT; r-- > 0;) run_end[r] = (r + 1 < NT && pv[r + 1] == pv[r]) ? run_end[r + 1] : r;
        GCHK(h2d(ctx, d_run_end, run_end.data(), (size_t)NT * 4));
        GCHK(d_flags.ensure((size_t)P * 4));
        HIPCHK(hipMemsetAsync(d_flags.p, 0, (size_t)P * 4, ctx->stream));
        HIPCHK(hipMemsetAsync(scratch.p, 0, (size_t)grid * 2 * NT * 4, ctx->stream));         // (the kernel keeps the counts zero between paths)
        GLAUNCH(ctx, "k_path_ks2", k_path_ks2, dim3(grid), dim3(GASM_WG), 0, dp.view(), ss.d_poscnt.as<u32>(), tb.d_row.as<int32_t>(), kmer, NT, d_pv.as<double>(),
                d_cumy.as<u32>(), d_run_end.as<u32>(), scratch.as<u32>(), d_out.as<double>(), d_flags.as<u32>(), P, bins);
        std::vector<u32> h_flags(P);
        HIPCHK(hipMemcpyAsync(h_flags.data(), d_flags.p, (size_t)P * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(ks.data(), d_out.p, (size_t)P * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        for (u32 p = 0; p < P; ++p) if (h_flags[p]) todo.push_back(p);
        if (todo.empty()) return GASM_OK;
    }
    {
        DBuf d_list;
        struct Rel3 { DBuf* b; ~Rel3() { b->release(); } } rel3{&d_list};
        const u32 n = v2 ? (u32)todo.size() : P;
        if (v2) GCHK(h2d(ctx, d_list, todo.data(), (size_t)n * 4));
        const u32 g1 = std::min<u32>(n, (u32)ctx->n_cu * 2u);
        GLAUNCH(ctx, "k_path_ks", k_path_ks, dim3(g1), dim3(GASM_WG), 0, dp.view(), ss.d_poscnt.as<u32>(), tb.d_row.as<int32_t>(), kmer, NT, d_pv.as<double>(),
                d_cumy.as<u32>(), scratch.as<u32>(), d_out.as<double>(), n, v2 ? d_list.as<u32>() : (const u32*)nullptr);
        HIPCHK(hipMemcpyAsync(ks.data(), d_out.p, (size_t)P * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    return GASM_OK;
}

int pipeline_coverage(gasm_ctx* ctx, const long long* start, const long long* len, u64 n, long long seq_len, double* percent) {
    *percent = 0.0;
    if (seq_len <= 0) return GASM_OK;
    HIPCHK(hipSetDevice(ctx->device));
    DBuf d_s, d_l, diff, cov;
    struct Rel { std::vector<DBuf*> v; ~Rel() { for (DBuf* b : v) b->release(); } } rel{{&d_s, &d_l, &diff, &cov}};
    GCHK(h2d(ctx, d_s, start, n * 8));
    GCHK(h2d(ctx, d_l, len, n * 8));
    GCHK(diff.ensure(((size_t)seq_len + 3) * 4));
    GCHK(cov.ensure(8));
    HIPCHK(hipMemsetAsync(diff.p, 0, ((size_t)seq_len + 3) * 4, ctx->stream));
    if (n) GLAUNCH(ctx, "k_cover_mark", k_cover_mark, dim3(ceil_div_u64(n, GASM_WG)), dim3(GASM_WG), 0, d_s.as<long long>(), d_l.as<long long>(), n, seq_len, diff.as<int>());
    GLAUNCH(ctx, "k_cover_count", k_cover_count, dim3(1), dim3(1024), 0, diff.as<int>(), seq_len, cov.as<unsigned long long>());
    unsigned long long c = 0;
    HIPCHK(hipMemcpyAsync(&c, cov.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    // lib/DeNovoAssembler.R:445: (1 - uncovered / seq_len) * 100
    *percent = (1.0 - (double)(seq_len - (long long)c) / (double)seq_len) * 100.0;
    return GASM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// score: pipeline_score_launch queues everything on the stream; pipeline_score_fetch copies the results back.
// ---------------------------------------------------------------------------------------------------------------
// Batch scoring of a build's own contigs (gasm_batch_score): queued behind the build without waiting for it.  The number
// of paths is read on the device (graph.d_seg_cstart[S]); outputs are allocated at the build's upper bound.
static int score_launch_graph(gasm_ctx* ctx, DevReads& rd, DevPaths& dp, int kmer, const ScoreTable& tb, ScoreState& ss, const BuildState& graph) {
    GasmRange range("gasm:score (graph-indexed)");
    const u32 S = rd.n_segments;
    const size_t PC = (size_t)graph.D_cap + 1;
    ss.stride = PC;
    ss.graph = &graph;
    ss.n_paths = 0;
    const size_t fx_off = (PC * 4 + 15) & ~(size_t)15;
    GCHK(ss.d_total.ensure(fx_off + PC * 8));
    GCHK(ss.d_out_f64.ensure(PC * 8 * 3));
    GCHK(ss.d_out_i32.ensure(PC * 4 * 2));
    unsigned long long* const d_fx = reinterpret_cast<unsigned long long*>(static_cast<char*>(ss.d_total.p) + fx_off);
    const u32* const n_paths_p = graph.d_seg_cstart.as<u32>() + S;
    const u64 p_est = graph.have_actual ? std::max<u64>(graph.n_contigs, 1) : (u64)S * 256;
    const u32 grid_p = (u32)std::max<u64>(1, std::min<u64>(ceil_div_u64(p_est, GASM_WG), (u64)ctx->n_cu * 8));
    GLAUNCH(ctx, "k_score_zero", k_score_zero, dim3(grid_p), dim3(GASM_WG), 0, ss.d_total.as<u32>(), d_fx, n_paths_p);
    const PathSet ps = dp.view();
    // the paths are this build's contigs and every read holds a k-mer: the sorted edge list is the index, and the
    // per-path sums are accumulated per read in fixed point (no position counters)
    GraphView gv;
    gv.dk_key = graph.d_dk_key.p;
    gv.dstart = graph.d_dstart.as<u32>();
    gv.fdir = graph.d_fdir.as<u16>();
    gv.k = graph.k;
    gv.bbits = graph.bbits;
    gv.fbits = graph.fbits;
    u64 max_reads = 0;
    for (u32 s = 0; s < S; ++s) max_reads = std::max(max_reads, rd.h_seg_read_off[s + 1] - rd.h_seg_read_off[s]);
    GCHK(const_cast<ScoreTable&>(tb).set_fixed(ctx, max_reads));
    // GASM_SCORE_VERIFY=1: compare every read with the contig text where the graph says it lies (kernels_score.hip, graph_match);
    // a mismatch raises flags[2] of the build and pipeline_score_fetch refuses the scores
    const int verify = env_int("GASM_SCORE_VERIFY", 0);
    ss.verify = verify != 0;
    const u32 reads_per_wg = 256;     // one read per thread: the match is a chain of dependent loads
    const u32 rchunks = (u32)ceil_div_u64(max_reads, reads_per_wg);
    // per-path accumulators kept in LDS: sized from the last build of these reads (a segment with more paths than that
    // goes to global atomics: slower, same numbers)
    const u32 lds_paths = std::min<u32>(std::max<u32>(graph.have_actual ? graph.paths_est : 1024u, 1u), GASM_SCORE_PATH_CAP);
    if (graph.words == 1) {
        GLAUNCH(ctx, "k_score_reads_graph", k_score_reads_graph<u64>, seg_grid(rchunks, S), dim3(GASM_WG), (size_t)lds_paths * 12, rd.view(),
                gv, graph.d_link.as<u64>(), graph.d_ecid.as<u32>(), ps, tb.d_fix.as<long long>(), kmer, reads_per_wg, rchunks, lds_paths, ss.d_total.as<u32>(),
                d_fx, verify, graph.d_flags.as<u32>() + 2);
    } else {
        GLAUNCH(ctx, "k_score_reads_graph", k_score_reads_graph<K128>, seg_grid(rchunks, S), dim3(GASM_WG), (size_t)lds_paths * 12, rd.view(),
                gv, graph.d_link.as<u64>(), graph.d_ecid.as<u32>(), ps, tb.d_fix.as<long long>(), kmer, reads_per_wg, rchunks, lds_paths, ss.d_total.as<u32>(),
                d_fx, verify, graph.d_flags.as<u32>() + 2);
    }
    double* o_bp = ss.d_out_f64.as<double>();
    double* o_nf = o_bp + PC;
    double* o_nl = o_nf + PC;
    int32_t* o_br = ss.d_out_i32.as<int32_t>();
    int32_t* o_ln = o_br + PC;
    const u64* d_se = nullptr;
    if (rd.n_empty) { GCHK(h2d(ctx, ss.d_seg_empty, rd.h_seg_empty.data(), (size_t)S * 8)); d_se = ss.d_seg_empty.as<u64>(); }
    GLAUNCH(ctx, "k_score_finish", k_score_finish, dim3(grid_p), dim3(GASM_WG), 0, ps, ss.d_total.as<u32>(),
            d_fx, tb.d_fix.as<long long>(), d_se, kmer, std::ldexp(1.0, -tb.fix_shift), o_bp, o_nf, o_nl, o_br, o_ln, n_paths_p);
    ss.h_pd_off.clear();
    ss.launched = true;
    return GASM_OK;
}

bool pipeline_score_uses_graph(const DevReads& rd, const BuildState& graph) {
    return graph.n_kmers > 0 && rd.n_reads > rd.n_empty && rd.min_len >= (u32)graph.k;
}

int pipeline_score_launch(gasm_ctx* ctx, DevReads& rd, DevPaths& dp, int kmer, const ScoreTable& tb, bool want_freq, bool want_pd,
                          ScoreState& ss, const BuildState* graph) {
    if (kmer < 0) { gasm_set_error("kmer must be >= 0"); return GASM_ERR_INVALID; }
    if (rd.n_segments != dp.n_segments) { gasm_set_error("reads and paths disagree on the number of segments"); return GASM_ERR_INVALID; }
    HIPCHK(hipSetDevice(ctx->device));
    GasmRange range("gasm:score");
    ss.valid = false;
    ss.n_table = tb.n_table;
    ss.want_freq = want_freq && tb.n_table;
    ss.want_pd = want_pd;
    ss.graph = nullptr;
    if (graph && !want_freq && !want_pd && pipeline_score_uses_graph(rd, *graph)) return score_launch_graph(ctx, rd, dp, kmer, tb, ss, *graph);
    // ---- arbitrary paths (or reads that do not all hold a k-mer): host-side sizes are needed — the caller has uploaded
    // the paths or, for a build's contigs, read the build's report (pipeline_contig_paths_host)
    ss.n_paths = dp.n_paths;
    const u32 S = rd.n_segments, P = dp.n_paths;
    const u64 TB = dp.total_bases;
    ss.stride = (size_t)P + 1;
    GCHK(ss.d_poscnt.ensure((TB + 1) * 4));
    GCHK(ss.d_total.ensure(((size_t)P + 1) * 4));
    GCHK(ss.d_out_f64.ensure(((size_t)P + 1) * 8 * 3));
    GCHK(ss.d_out_i32.ensure(((size_t)P + 1) * 4 * 2));
    HIPCHK(hipMemsetAsync(ss.d_poscnt.p, 0, (TB + 1) * 4, ctx->stream));
    HIPCHK(hipMemsetAsync(ss.d_total.p, 0, ((size_t)P + 1) * 4, ctx->stream));
    const PathSet ps = dp.view();
    const int w = (int)std::min<u32>(32, rd.min_len);
    if (P && TB && rd.n_reads > rd.n_empty && w >= 1) {
        // per-segment read tables (power of two, at least twice the reads) and the dense first-occurrence table
        std::vector<u64>& toff = ss.h_toff;
        toff.assign((size_t)S + 1, 0);
        std::vector<u64> foff((size_t)S + 1, 0);
        u64 max_bases = 0, max_reads = 0;
        for (u32 s = 0; s < S; ++s) {
            const u64 nbases = dp.h_seg_base_off[s + 1] - dp.h_seg_base_off[s];
            const u64 nreads = rd.h_seg_read_off[s + 1] - rd.h_seg_read_off[s];
            const u64 npaths = dp.h_seg_path_off[s + 1] - dp.h_seg_path_off[s];
            u64 slots = 2;
            while (slots < 2 * nreads) slots <<= 1;
            toff[s + 1] = toff[s] + slots;
            foff[s + 1] = foff[s] + npaths * nreads;
            max_bases = std::max(max_bases, nbases);
            max_reads = std::max(max_reads, nreads);
        }
        // the dense first-occurrence table holds paths x reads words: at most `budget` of them at a time, else the paths of
        // a segment are scored slice by slice (the read table is built once)
        const u64 budget = getenv("GASM_DBG_FIRST_BUDGET") ? (u64)atoll(getenv("GASM_DBG_FIRST_BUDGET")) : (2ull << 30);
        const bool sliced = foff[S] > budget;
        if (sliced && dp.h_p_off.size() != (size_t)P + 1) { gasm_set_error("paths x reads too large for one pass and the path offsets are not on the host"); return GASM_ERR_CAPACITY; }
        if (rd.n_reads > 0xFFFFFFF0ull || TB > 0xFFFFFFF0ull) { gasm_set_error("too many reads or path bases for 32-bit indices"); return GASM_ERR_CAPACITY; }
        GCHK(h2d(ctx, ss.d_tbl_off, toff.data(), toff.size() * 8));
        GCHK(h2d(ctx, ss.d_first_off, foff.data(), foff.size() * 8));
        GCHK(ss.d_seed.ensure(toff[S] * 8));
        GCHK(ss.d_gpos.ensure(toff[S] * 4));
        HIPCHK(hipMemsetAsync(ss.d_gpos.p, 0xFF, toff[S] * 4, ctx->stream));
        SeedTable st;
        st.seed = ss.d_seed.as<u64>();
        st.gpos = ss.d_gpos.as<u32>();
        st.tbl_off = ss.d_tbl_off.as<u64>();
        if (max_bases && max_reads) {
            GLAUNCH(ctx, "k_read_insert", k_read_insert, dim3(ceil_div_u64(max_reads, GASM_WG), S), dim3(GASM_WG), 0, rd.view(), st, w);
            if (!sliced) {
                GCHK(ss.d_first.ensure(foff[S] * 4 + 16));
                HIPCHK(hipMemsetAsync(ss.d_first.p, 0xFF, foff[S] * 4 + 16, ctx->stream));
                GLAUNCH(ctx, "k_path_scan", k_path_scan, dim3(ceil_div_u64(max_bases, GASM_WG), S), dim3(GASM_WG), 0, rd.view(), ps, st,
                        dp.seg_base_off_dev(), w, ss.d_first_off.as<u64>(), ss.d_first.as<u32>(), 0u, 0u, 0u);
                GLAUNCH(ctx, "k_first_to_poscnt", k_first_to_poscnt, dim3(ceil_div_u64(foff[S], GASM_WG)), dim3(GASM_WG), 0, ss.d_first.as<u32>(),
                        foff[S], ss.d_poscnt.as<u32>());
            } else {
                for (u32 s = 0; s < S; ++s) {
                    const u64 nreads = rd.h_seg_read_off[s + 1] - rd.h_seg_read_off[s];
                    const u32 p0 = dp.h_seg_path_off[s], p1 = dp.h_seg_path_off[s + 1];
                    if (!nreads || p0 == p1) continue;
                    const u32 step = (u32)std::max<u64>(1, std::min<u64>(budget / nreads, 0x7FFFFFFFull));
                    for (u32 a = p0; a < p1; a += step) {
                        const u32 e = (u32)std::min<u64>((u64)a + step, p1);
                        const u64 entries = (u64)(e - a) * nreads, bases = dp.h_p_off[e] - dp.h_p_off[a];
                        GCHK(ss.d_first.ensure(entries * 4 + 16));
                        HIPCHK(hipMemsetAsync(ss.d_first.p, 0xFF, entries * 4 + 16, ctx->stream));
                        if (bases) {
                            GLAUNCH(ctx, "k_path_scan", k_path_scan, dim3(ceil_div_u64(bases, GASM_WG), 1), dim3(GASM_WG), 0, rd.view(), ps, st,
                                    dp.seg_base_off_dev(), w, ss.d_first_off.as<u64>(), ss.d_first.as<u32>(), s, a, e);
                            GLAUNCH(ctx, "k_first_to_poscnt", k_first_to_poscnt, dim3(ceil_div_u64(entries, GASM_WG)), dim3(GASM_WG), 0,
                                    ss.d_first.as<u32>(), entries, ss.d_poscnt.as<u32>());
                        }
                    }
                }
            }
        }
    }
    if (P && rd.n_empty) {
        GCHK(h2d(ctx, ss.d_seg_empty, rd.h_seg_empty.data(), (size_t)S * 8));
        u32 maxp = 0;
        for (u32 s = 0; s < S; ++s) maxp = std::max(maxp, dp.h_seg_path_off[s + 1] - dp.h_seg_path_off[s]);
        if (maxp) hipLaunchKernelGGL(k_add_empty_reads, dim3(ceil_div_u64(maxp, GASM_WG), S), dim3(GASM_WG), 0, ctx->stream, ps,
                                     ss.d_seg_empty.as<u64>(), ss.d_poscnt.as<u32>(), ss.d_total.as<u32>());
    }
    double* o_bp = ss.d_out_f64.as<double>();
    double* o_nf = o_bp + (P + 1);
    double* o_nl = o_nf + (P + 1);
    int32_t* o_br = ss.d_out_i32.as<int32_t>();
    int32_t* o_ln = o_br + (P + 1);
    if (P) {
        GLAUNCH(ctx, "k_path_reduce", k_path_reduce, dim3(P), dim3(GASM_WG), 0, ps, ss.d_poscnt.as<u32>(),
                ss.d_total.as<u32>(), tb.d_prob.as<double>(), kmer, o_bp, o_nf, o_nl, o_br, o_ln, P);
    }
    if (P && ss.want_freq) {
        const size_t cells = (size_t)P * tb.n_table;
        GCHK(ss.d_freq.ensure(cells * 4));
        HIPCHK(hipMemsetAsync(ss.d_freq.p, 0, cells * 4, ctx->stream));
        GLAUNCH(ctx, "k_path_freq", k_path_freq, dim3(ceil_div_u64(P, GASM_WG / 64)), dim3(GASM_WG), 0, ps, ss.d_poscnt.as<u32>(),
                ss.d_total.as<u32>(), tb.d_row.as<int32_t>(), kmer, tb.n_table, ss.d_freq.as<u32>(), P);
    }
    ss.h_pd_off.clear();
    if (want_pd) {
        ss.h_pd_off.assign((size_t)P + 1, 0);
        for (u32 p = 0; p < P; ++p) {
            const u64 len = dp.h_p_off[p + 1] - dp.h_p_off[p];
            ss.h_pd_off[p + 1] = ss.h_pd_off[p] + (len >= (u64)kmer ? len - kmer + 1 : 0);
        }
        if (P) {
            GCHK(h2d(ctx, ss.d_pd_off, ss.h_pd_off.data(), ss.h_pd_off.size() * 8));
            GCHK(ss.d_pd.ensure((ss.h_pd_off[P] + 1) * 8));
            GLAUNCH(ctx, "k_prob_dist", k_prob_dist, dim3(ceil_div_u64(P, GASM_WG / 64)), dim3(GASM_WG), 0, ps, tb.d_prob.as<double>(), kmer,
                    ss.d_pd_off.as<u64>(), ss.d_pd.as<double>(), P);
        }
    }
    ss.launched = true;
    return GASM_OK;
}

int pipeline_score_fetch(gasm_ctx* ctx, ScoreState& ss) {
    if (!ss.launched) { gasm_set_error("no scoring has been queued"); return GASM_ERR_STATE; }
    if (ss.valid) return GASM_OK;
    if (ss.graph) {
        if (ss.graph->pending) { gasm_set_error("scores fetched before the build's report was read"); return GASM_ERR_STATE; }
        ss.n_paths = ss.graph->n_contigs;
        if (ss.verify) {
            u32 bad = 0;
            HIPCHK(hipMemcpyAsync(&bad, ss.graph->d_flags.as<u32>() + 2, 4, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            if (bad) { gasm_set_error("GASM_SCORE_VERIFY: a read that fits its contig by the graph differs from the contig's text"); return GASM_ERR_STATE; }
        }
    }
    const u32 P = ss.n_paths;
    double* o_bp = ss.d_out_f64.as<double>();
    double* o_nf = o_bp + ss.stride;
    double* o_nl = o_nf + ss.stride;
    int32_t* o_br = ss.d_out_i32.as<int32_t>();
    int32_t* o_ln = o_br + ss.stride;
    ss.h_bp.resize(P); ss.h_nf.resize(P); ss.h_nl.resize(P); ss.h_breaks.resize(P); ss.h_len.resize(P);
    ss.h_freq.clear(); ss.h_pd.clear();
    std::vector<u32> h_fc;
    if (P) {
        HIPCHK(hipMemcpyAsync(ss.h_bp.data(), o_bp, (size_t)P * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(ss.h_nf.data(), o_nf, (size_t)P * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(ss.h_nl.data(), o_nl, (size_t)P * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(ss.h_breaks.data(), o_br, (size_t)P * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(ss.h_len.data(), o_ln, (size_t)P * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (ss.want_freq) {
            h_fc.resize((size_t)P * ss.n_table);
            HIPCHK(hipMemcpyAsync(h_fc.data(), ss.d_freq.p, h_fc.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
        }
        if (ss.want_pd && ss.h_pd_off[P]) {
            ss.h_pd.resize(ss.h_pd_off[P]);
            HIPCHK(hipMemcpyAsync(ss.h_pd.data(), ss.d_pd.p, ss.h_pd_off[P] * 8, hipMemcpyDeviceToHost, ctx->stream));
        }
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (!h_fc.empty()) {
        // observed_breaks = count / total (lib/DeNovoAssembler.cpp:402); 0/0 = NaN as in the reference
        ss.h_freq.resize(h_fc.size());
        for (u32 p = 0; p < P; ++p) {
            const double tot = (double)ss.h_breaks[p];
            for (u32 j = 0; j < ss.n_table; ++j) ss.h_freq[(size_t)p * ss.n_table + j] = (double)h_fc[(size_t)p * ss.n_table + j] / tot;
        }
    }
    ss.valid = true;
    return GASM_OK;
}
