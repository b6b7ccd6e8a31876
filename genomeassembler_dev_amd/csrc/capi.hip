// capi.hip — the extern "C" surface declared in include/gasm.h.
#include <algorithm>
#include <atomic>
#include <functional>
#include <mutex>
#include <new>
#include <thread>

#include "pipeline.h"
#include "scaffolds.h"

struct gasm_strlist {
    std::vector<char> data;
    std::vector<u64> off;
};

struct gasm_contigs {
    std::vector<char> data;
    std::vector<u64> off;
    u64 rows = 0;
    std::vector<u32> perm;
    int words = 1;
    std::vector<u64> dkeys;
    std::vector<u32> dmult;
};

struct gasm_packed {
    gasm_host::PackedReads pr;
    std::vector<u64> seg;
    u64 dropped = 0;
    std::vector<u8> on_device;       // per file: parsed and packed by the device path (gasm_read_files_device)
};

struct gasm_scores {
    u64 n = 0;
    std::vector<int32_t> len, breaks, lev, startpos;
    std::vector<double> bp, nf, nl, freq, pd, ks;
    std::vector<u64> pd_off;
    bool has_freq = false, velvet = false, has_ks = false;
    int lev_device = 0;      // who computed lev: 0 nobody (not asked), 1 the GPU (k_levenshtein), 2 host threads (gasm_host::levenshtein)
};

// A batch runs as one or more sub-batches — contiguous blocks of its segments, each with its own reads, build and
// scoring state on its own stream (a lane of the context).  Segments are independent, so this changes nothing in the
// results; it lets the latency-bound graph phase of one block run under the bandwidth-bound partition/de-duplication
// phase of the next: block j's first kernel waits (stream event) until block j-1's streaming kernels are queued, and
// with builds queued ahead of their reports the blocks of consecutive steps interleave the same way.
// What one step (build + scoring) of a block owns: the graph, the contigs as paths, the scores — and the stream it runs on.
struct StepSlot {
    gasm_ctx* cx = nullptr;
    hipEvent_t ev_streamed = nullptr;       // its build's streaming kernels (partition, de-duplication) are done
    BuildState bs;
    DevPaths dp;
    ScoreState ss;
    bool paths_ready = false;
};

struct SubBatch {
    gasm_ctx* cx = nullptr;                 // the lane it runs on (lane 0 = the batch's context itself)
    u32 seg0 = 0, seg1 = 0;                 // its block of the batch's segments
    DevReads rd;
    ScoreTable tb;
    bool table_set = false;
    hipEvent_t ev_streamed = nullptr;
    GuidedState guided;
    // Two step slots, taken in turn (round 3; one-block batches, GASM_PINGPONG=0 switches it off): consecutive steps of a
    // resident pipeline are independent of each other — same reads, own graph, own scores — so step n + 1 is queued on the
    // other slot's stream with the other slot's buffers and its streaming kernels (partition, de-duplication: HBM and LDS)
    // run beside step n's graph and scoring kernels (latency-bound, a few waves per CU).  Results are always those of the
    // slot the last gasm_batch_build took.
    StepSlot slot[4];
    int cur = 0, n_slots = 2;
    bool pingpong = false;
    // GASM_PINGPONG=1 (default): whole steps on equal streams, GASM_STEP_SLOTS of them (2..4, default 3).
    // GASM_PINGPONG=2: every step's streaming kernels on the block's own stream, back to back, and each slot's graph, contigs
    // and scoring on a tail lane of its own, whose stream the dispatcher serves first.  Measured: the
    // persistent streaming workgroups hold the CUs' LDS, the small kernels beside them run several times slower, and the
    // next-but-one partition waits for them — 0.92-0.98 ms/step against 0.88 with equal streams.
    int pp_mode = 1;
    StepSlot& S() {
        StepSlot& x = slot[cur];
        if (!x.cx && pingpong && pp_mode == 2) x.cx = cx->tail_lane(cur);
        if (!x.cx) x.cx = cur ? cx->lane((size_t)cur) : cx;
        if (!x.cx) x.cx = cx;                // (no second stream to be had: both slots on the block's own)
        return x;
    }
    const StepSlot& S() const { return slot[cur]; }
    // Scoring on a stream of its own (one-block batches without ping-pong): the graph-indexed scoring of step n only reads what
    // build n left behind and what the NEXT build does not touch before its de-duplication (which rewrites the directories the
    // scorer searches) — so it runs on a lane beside the next build's partition instead of in front of it.  ev_built: the build
    // is queued in full (the scorer's lane waits for it); ev_scored: the scoring is done (the next de-duplication waits for it).
    gasm_ctx* scx = nullptr;                // the lane
    gasm_ctx* score_cx_last = nullptr;      // where the last scoring was queued (fetches read from there)
    hipEvent_t ev_built = nullptr, ev_scored = nullptr;
    bool lane_last = false;                 // the last build recorded ev_built for the lane (GASM_SCORE_LANE is read per build)
};

static bool score_lane_wanted(const gasm_batch* b);
// the lane and its events, on first use; false: no lane (more than one block, switched off, or no resources)
static bool score_lane(gasm_batch* b, SubBatch& sb);

struct gasm_batch {
    gasm_ctx* ctx = nullptr;
    std::vector<SubBatch> sub;
    u32 n_segments = 0;
    u64 n_reads = 0;
    bool built = false;
    std::vector<double> table_copy;
    bool table_given = false;
    // the last gasm_batch_score, kept to queue it again behind a build that had to be repeated
    bool scored = false;
    int score_kmer = 0;
    int last_k = 0;                         // k of the last gasm_batch_build
    // concatenated host results of the sub-batches (fetch)
    std::vector<u64> h_seg_doff, h_dk_key, h_seg_coff, h_c_off;
    std::vector<u32> h_dk_cnt, h_nxt;
    std::vector<u8> h_eflag;
    std::vector<char> h_contigs;
    std::vector<double> h_bp, h_nf, h_nl;
    std::vector<int32_t> h_breaks, h_len;
    // simulated batches: the start of every read in its genome
    DBuf d_read_start;
    std::vector<u32> h_read_start;
    std::vector<int64_t> h_fx;
    std::vector<u64> h_sim_seg_off;
};

// Read the report of a sub-batch's queued build (repeating the build if it failed, and then the scoring queued behind it).
static int sub_finish(gasm_batch* b, SubBatch& sb) {
    bool rebuilt = false;
    GCHK(pipeline_build_finish(sb.S().cx, sb.rd, sb.S().bs, &rebuilt));
    if (rebuilt) {
        sb.S().paths_ready = false;
        if (b->scored) {
            GCHK(pipeline_contig_paths(sb.S().cx, sb.rd, sb.S().bs, sb.S().dp));
            sb.S().paths_ready = true;
            pipeline_contig_paths_host(sb.rd, sb.S().bs, sb.S().dp);
            if (sb.scx) HIPCHK(hipStreamSynchronize(sb.scx->stream));      // (the first attempt's scoring: its buffers are reused)
            GCHK(pipeline_score_launch(sb.S().cx, sb.rd, sb.S().dp, b->score_kmer, sb.tb, false, false, sb.S().ss, &sb.S().bs));
            sb.score_cx_last = sb.S().cx;
        }
    }
    return GASM_OK;
}
static int batch_finish(gasm_batch* b) {
    for (SubBatch& sb : b->sub) {
        GCHK(sub_finish(b, sb));
        if (sb.scx) HIPCHK(hipStreamSynchronize(sb.scx->stream));      // whatever the lane still scores: results are asked for
    }
    return GASM_OK;
}

#define API_GUARD_BEGIN try {
#define API_GUARD_END                                                          \
    } catch (const std::bad_alloc&) {                                          \
        gasm_set_error("out of host memory");                                  \
        return GASM_ERR_CAPACITY;                                              \
    } catch (const std::exception& e) {                                        \
        gasm_set_error("internal error: %s", e.what());                        \
        return GASM_ERR_INVALID;                                               \
    }

static bool score_lane_wanted(const gasm_batch* b) {
    const char* v = getenv("GASM_SCORE_LANE");
    return b->sub.size() == 1 && !(v && *v == '0');
}
static bool score_lane(gasm_batch* b, SubBatch& sb) {
    if (!score_lane_wanted(b)) return false;
    if (sb.scx && sb.ev_built && sb.ev_scored) return true;
    if (hipSetDevice(sb.cx->device) != hipSuccess) return false;
    if (!sb.scx) sb.scx = sb.cx->lane(0);
    if (!sb.scx) return false;
    if (!sb.ev_built && hipEventCreateWithFlags(&sb.ev_built, hipEventDisableTiming) != hipSuccess) { sb.ev_built = nullptr; return false; }
    if (!sb.ev_scored && hipEventCreateWithFlags(&sb.ev_scored, hipEventDisableTiming) != hipSuccess) { sb.ev_scored = nullptr; return false; }
    return true;
}

static void strlist_from(const std::vector<std::string>& v, std::vector<char>& data, std::vector<u64>& off) {
    off.assign(v.size() + 1, 0);
    size_t tot = 0;
    for (size_t i = 0; i < v.size(); ++i) { tot += v[i].size(); off[i + 1] = tot; }
    data.resize(tot);
    for (size_t i = 0; i < v.size(); ++i) memcpy(data.data() + off[i], v[i].data(), v[i].size());
}

extern "C" {

static int breakscore_impl(gasm_ctx* ctx, DevPaths& dp, const std::function<std::string(u64)>& path_text, uint64_t n_paths, const char* reads,
                           const uint64_t* read_off, uint64_t n_reads, const char* true_solution, uint64_t true_len, int kmer,
                           const char* bp_kmer, const uint64_t* bp_off, uint64_t n_table, const double* bp_prob, int variant,
                           int flags, gasm_scores** out);

// ------------------------------------------------------------------------------------------------- get_contigs
// reads (ragged when read_off != nullptr, else n_reads reads of fixed_len) of ONE segment -> contigs + shuffle matrix
static int contigs_of_reads(gasm_ctx* ctx, const char* bases, const u64* read_off, u64 n_reads, u32 fixed_len, int dbg_kmer, int seed, int matrix_rows,
                            gasm_contigs** out) {
    DevReads rd;
    BuildState bs;
    const u64 seg_off[2] = {0, n_reads};
    int st = rd.upload(ctx, bases, read_off, n_reads, fixed_len, seg_off, 1);
    if (st == GASM_OK) st = pipeline_build(ctx, rd, dbg_kmer, 0, bs);
    if (st == GASM_OK) st = pipeline_fetch_distinct(ctx, rd, bs);
    if (st == GASM_OK) st = pipeline_fetch_contigs(ctx, rd, bs);
    gasm_contigs* c = nullptr;
    if (st == GASM_OK) {
        c = new gasm_contigs();
        c->data = bs.h_contigs;
        c->off = bs.h_c_off;
        c->words = bs.words;
        c->dkeys = bs.h_dk_key;
        c->dmult = bs.h_dk_cnt;
        c->rows = (u64)matrix_rows;
        // lib/DeNovoAssembler.cpp:195-203
        gasm_host::shuffle_perm(bs.n_contigs, seed, (u64)matrix_rows, c->perm);
    }
    rd.release();
    bs.release();
    if (st != GASM_OK) return st;
    *out = c;
    return GASM_OK;
}

int gasm_get_contigs(gasm_ctx* ctx, const char* kmers, uint64_t n_kmers, int dbg_kmer, int seed, int matrix_rows,
                     gasm_contigs** out) {
    API_GUARD_BEGIN
    if (!ctx || !out || (n_kmers && !kmers)) { gasm_set_error("gasm_get_contigs: null argument"); return GASM_ERR_INVALID; }
    if (matrix_rows < 0) { gasm_set_error("matrix_rows must be >= 0"); return GASM_ERR_INVALID; }
    *out = nullptr;
    // the exploded k-mers are reads of length k with one k-mer each
    return contigs_of_reads(ctx, kmers, nullptr, n_kmers, (u32)dbg_kmer, dbg_kmer, seed, matrix_rows, out);
    API_GUARD_END
}

int gasm_get_contigs_from_reads(gasm_ctx* ctx, const char* reads, const uint64_t* read_off, uint64_t n_reads, int dbg_kmer, int seed,
                                int matrix_rows, gasm_contigs** out) {
    API_GUARD_BEGIN
    if (!ctx || !out || (n_reads && (!reads || !read_off))) { gasm_set_error("gasm_get_contigs_from_reads: null argument"); return GASM_ERR_INVALID; }
    if (matrix_rows < 0) { gasm_set_error("matrix_rows must be >= 0"); return GASM_ERR_INVALID; }
    *out = nullptr;
    return contigs_of_reads(ctx, reads, read_off, n_reads, 0, dbg_kmer, seed, matrix_rows, out);
    API_GUARD_END
}

uint64_t gasm_contigs_count(const gasm_contigs* c) { return c ? c->off.size() - 1 : 0; }
const char* gasm_contigs_data(const gasm_contigs* c) { return c ? c->data.data() : nullptr; }
const uint64_t* gasm_contigs_offsets(const gasm_contigs* c) { return c ? c->off.data() : nullptr; }
uint64_t gasm_contigs_rows(const gasm_contigs* c) { return c ? c->rows : 0; }
const uint32_t* gasm_contigs_perm(const gasm_contigs* c) { return c ? c->perm.data() : nullptr; }
uint64_t gasm_contigs_distinct_count(const gasm_contigs* c) { return c ? c->dmult.size() : 0; }
int gasm_contigs_key_words(const gasm_contigs* c) { return c ? c->words : 0; }
const uint64_t* gasm_contigs_distinct_keys(const gasm_contigs* c) { return c ? c->dkeys.data() : nullptr; }
const uint32_t* gasm_contigs_distinct_mult(const gasm_contigs* c) { return c ? c->dmult.data() : nullptr; }
void gasm_contigs_free(gasm_contigs* c) { delete c; }

// -------------------------------------------------------------------------------------------- assemble_contigs
// the device route: greedy merge on contig indices (host threads), scaffolds expanded, ordered and de-duplicated on the GPU
// *used = false when the index form does not apply (the caller takes the host's string form)
static int assemble_device(gasm_ctx* ctx, const char* contigs, const u64* off, u64 n, const u32* perm, u64 rows, u64 row_len, int k, gasm_scaffolds** out,
                           bool* used) {
    *used = false;
    std::vector<std::string> c(n);
    for (u64 i = 0; i < n; ++i) c[i].assign(contigs + off[i], contigs + off[i + 1]);
    for (u64 i = 0; i < rows * row_len; ++i)
        if (perm[i] >= n) { gasm_set_error("perm[%llu] = %u out of range", (unsigned long long)i, perm[i]); return GASM_ERR_INVALID; }
    if (row_len != n) return GASM_OK;                       // (rows that are not permutations of all contigs: the string form)
    std::vector<std::string> sigs;
    bool on_gpu = false;
    // the merge itself on the GPU (a wave per permutation); GASM_ASM_HOST_MERGE=1: on host threads (same signatures)
    u64 rows_on_host = 0;
    if (!getenv("GASM_ASM_HOST_MERGE")) GCHK(assemble_signatures_device(ctx, c, perm, rows, row_len, k, sigs, &on_gpu, &rows_on_host));
    if (!on_gpu && !gasm_host::assemble_signatures(c, perm, rows, row_len, k, sigs)) return GASM_OK;
    *used = true;
    GCHK(scaffolds_from_signatures(ctx, c, sigs, out));
    // who ran the greedy merge (the results are the same; a caller can ask): the GPU, host threads, or both
    (*out)->rows_total = rows;
    (*out)->rows_on_host = on_gpu ? rows_on_host : rows;
    return GASM_OK;
}

static int assemble_common(gasm_ctx* ctx, const char* contigs, const u64* off, u64 n, const u32* perm, u64 rows, u64 row_len, int k, gasm_strlist** out) {
    if (ctx && !getenv("GASM_ASM_HOST")) {
        gasm_scaffolds* sc = nullptr;
        bool used = false;
        GCHK(assemble_device(ctx, contigs, off, n, perm, rows, row_len, k, &sc, &used));
        if (used) {
            gasm_strlist* s = new gasm_strlist();
            const int st = scaffolds_fetch(sc, s->data, s->off);
            gasm_scaffolds_free(sc);
            if (st != GASM_OK) { delete s; return st; }
            *out = s;
            return GASM_OK;
        }
    }
    std::vector<std::string> c(n);
    for (u64 i = 0; i < n; ++i) c[i].assign(contigs + off[i], contigs + off[i + 1]);
    for (u64 i = 0; i < rows * row_len; ++i)
        if (perm[i] >= n) { gasm_set_error("perm[%llu] = %u out of range", (unsigned long long)i, perm[i]); return GASM_ERR_INVALID; }
    std::vector<std::string> res;
    GCHK(gasm_host::assemble(c, perm, rows, row_len, k, res));
    gasm_strlist* s = new gasm_strlist();
    strlist_from(res, s->data, s->off);
    *out = s;
    return GASM_OK;
}

int gasm_assemble_contigs(gasm_ctx* ctx, const char* contigs, const uint64_t* off, uint64_t n, const uint32_t* perm,
                          uint64_t rows, uint64_t row_len, int dbg_kmer, gasm_strlist** out) {
    API_GUARD_BEGIN
    if (!out || !off || (n && !contigs) || (rows && row_len && !perm)) { gasm_set_error("gasm_assemble_contigs: null argument"); return GASM_ERR_INVALID; }
    *out = nullptr;
    return assemble_common(ctx, contigs, off, n, perm, rows, row_len, dbg_kmer, out);
    API_GUARD_END
}

int gasm_assemble_contigs_velvet(gasm_ctx* ctx, const char* contigs, const uint64_t* off, uint64_t n, int dbg_kmer, int seed,
                                 int rows, gasm_strlist** out) {
    API_GUARD_BEGIN
    if (!out || !off || (n && !contigs) || rows < 0) { gasm_set_error("gasm_assemble_contigs_velvet: bad argument"); return GASM_ERR_INVALID; }
    *out = nullptr;
    std::vector<u32> perm;
    gasm_host::shuffle_perm(n, seed, (u64)rows, perm);  // lib/BreakageScorer.cpp:86-94
    return assemble_common(ctx, contigs, off, n, perm.data(), (u64)rows, n, dbg_kmer, out);
    API_GUARD_END
}

// ---- the same with the scaffolds left on the device
int gasm_assemble_contigs_dev(gasm_ctx* ctx, const char* contigs, const uint64_t* off, uint64_t n, const uint32_t* perm, uint64_t rows, uint64_t row_len,
                              int dbg_kmer, gasm_scaffolds** out) {
    API_GUARD_BEGIN
    if (!ctx || !out || !off || (n && !contigs) || (rows && row_len && !perm)) { gasm_set_error("gasm_assemble_contigs_dev: null argument"); return GASM_ERR_INVALID; }
    *out = nullptr;
    bool used = false;
    GCHK(assemble_device(ctx, contigs, off, n, perm, rows, row_len, dbg_kmer, out, &used));
    if (!used) {
        // contigs shorter than k-1 (the reference compares whole strings or throws there), or rows that are not full
        // permutations: the host's string form, then the text goes to the device as a one-contig-per-scaffold chain set
        gasm_strlist* sl = nullptr;
        GCHK(assemble_common(nullptr, contigs, off, n, perm, rows, row_len, dbg_kmer, &sl));
        std::vector<std::string> txt(sl->off.size() - 1), sigs;
        for (size_t i = 0; i + 1 < sl->off.size(); ++i) txt[i].assign(sl->data.data() + sl->off[i], sl->data.data() + sl->off[i + 1]);
        gasm_strlist_free(sl);
        // (already sorted, distinct and in final order: hand them over as they are)
        gasm_scaffolds* sc = new gasm_scaffolds();
        sc->ctx = ctx; sc->n = (u32)txt.size();
        sc->h_off.assign(1, 0);
        std::string cat;
        for (auto& t : txt) { cat += t; sc->h_off.push_back(cat.size()); }
        DevPaths tmp;
        std::vector<u64> o(sc->h_off);
        const int st = tmp.upload_ascii(ctx, cat.data(), o.data(), sc->n);
        if (st != GASM_OK) { tmp.release(); delete sc; return st; }
        sc->d_words = tmp.d_words; tmp.d_words = DBuf();
        tmp.release();
        *out = sc;
    }
    return GASM_OK;
    API_GUARD_END
}

int gasm_assemble_contigs_velvet_dev(gasm_ctx* ctx, const char* contigs, const uint64_t* off, uint64_t n, int dbg_kmer, int seed, int rows,
                                     gasm_scaffolds** out) {
    API_GUARD_BEGIN
    if (!ctx || !out || !off || (n && !contigs) || rows < 0) { gasm_set_error("gasm_assemble_contigs_velvet_dev: bad argument"); return GASM_ERR_INVALID; }
    std::vector<u32> perm;
    gasm_host::shuffle_perm(n, seed, (u64)rows, perm);  // lib/BreakageScorer.cpp:86-94
    return gasm_assemble_contigs_dev(ctx, contigs, off, n, perm.data(), (u64)rows, n, dbg_kmer, out);
    API_GUARD_END
}

uint64_t gasm_scaffolds_count(const gasm_scaffolds* s) { return s ? s->n : 0; }
int gasm_scaffolds_merge_device(const gasm_scaffolds* s, uint64_t* rows_on_host) {
    if (rows_on_host) *rows_on_host = s ? s->rows_on_host : 0;
    if (!s) return 0;
    return s->rows_on_host == 0 ? 1 : (s->rows_on_host >= s->rows_total ? 2 : 3);
}
const uint64_t* gasm_scaffolds_offsets(const gasm_scaffolds* s) { return s ? s->h_off.data() : nullptr; }
int gasm_scaffolds_fetch(const gasm_scaffolds* s, gasm_strlist** out) {
    API_GUARD_BEGIN
    if (!s || !out) { gasm_set_error("gasm_scaffolds_fetch: null argument"); return GASM_ERR_INVALID; }
    gasm_strlist* l = new gasm_strlist();
    const int st = scaffolds_fetch(s, l->data, l->off);
    if (st != GASM_OK) { delete l; return st; }
    *out = l;
    return GASM_OK;
    API_GUARD_END
}
void gasm_scaffolds_free(gasm_scaffolds* s) {
    if (!s) return;
    if (s->ctx) { (void)hipSetDevice(s->ctx->device); (void)hipStreamSynchronize(s->ctx->stream); }
    s->d_words.release();
    delete s;
}

int gasm_calc_breakscore_dev(gasm_ctx* ctx, const gasm_scaffolds* paths, const char* reads, const uint64_t* read_off, uint64_t n_reads,
                             const char* true_solution, uint64_t true_len, int kmer, const char* bp_kmer, const uint64_t* bp_off, uint64_t n_table,
                             const double* bp_prob, int variant, int flags, gasm_scores** out) {
    API_GUARD_BEGIN
    if (!ctx || !out || !paths || !read_off || !bp_off || (n_table && (!bp_kmer || !bp_prob)) || (true_len && !true_solution)) {
        gasm_set_error("gasm_calc_breakscore_dev: null argument");
        return GASM_ERR_INVALID;
    }
    if (variant != GASM_SCORE_OWN && variant != GASM_SCORE_VELVET) { gasm_set_error("unknown variant %d", variant); return GASM_ERR_INVALID; }
    *out = nullptr;
    DevPaths dp;
    int st = scaffolds_as_paths(paths, dp);
    // text of single paths is rarely needed (velvet startpos, host Levenshtein): fetched once, on first use
    std::vector<char> txt;
    std::vector<u64> toff;
    std::mutex mu;
    bool have = false;
    auto path_text = [&](u64 p) {
        std::lock_guard<std::mutex> g(mu);
        if (!have) { (void)scaffolds_fetch(paths, txt, toff); have = true; }
        return toff.size() > p + 1 ? std::string(txt.data() + toff[p], txt.data() + toff[p + 1]) : std::string();
    };
    if (st == GASM_OK)
        st = breakscore_impl(ctx, dp, path_text, paths->n, reads, read_off, n_reads, true_solution, true_len, kmer, bp_kmer, bp_off, n_table, bp_prob,
                             variant, flags, out);
    dp.release();
    return st;
    API_GUARD_END
}

uint64_t gasm_strlist_count(const gasm_strlist* s) { return s ? s->off.size() - 1 : 0; }
const char* gasm_strlist_data(const gasm_strlist* s) { return s ? s->data.data() : nullptr; }
const uint64_t* gasm_strlist_offsets(const gasm_strlist* s) { return s ? s->off.data() : nullptr; }
void gasm_strlist_free(gasm_strlist* s) { delete s; }

// -------------------------------------------------------------------------------------------- calc_breakscore
int gasm_levenshtein(const char* query, uint64_t nq, const char* target, uint64_t nt, int infix, int32_t* out) {
    API_GUARD_BEGIN
    if (!out || (nq && !query) || (nt && !target)) { gasm_set_error("gasm_levenshtein: null argument"); return GASM_ERR_INVALID; }
    *out = gasm_host::levenshtein(query, nq, target, nt, infix != 0);
    return GASM_OK;
    API_GUARD_END
}

// calc_breakscore proper.  `dp` holds the paths on the device (uploaded text or the scaffolds of a device-side
// assemble_contigs); path_text(p) hands out path p as text for the few host-side steps that want it (the velvet variant's
// startpos find, the host Levenshtein routine for a target outside ACGT).
static int breakscore_impl(gasm_ctx* ctx, DevPaths& dp, const std::function<std::string(u64)>& path_text, uint64_t n_paths, const char* reads,
                           const uint64_t* read_off, uint64_t n_reads, const char* true_solution, uint64_t true_len, int kmer,
                           const char* bp_kmer, const uint64_t* bp_off, uint64_t n_table, const double* bp_prob, int variant,
                           int flags, gasm_scores** out) {
    const bool velvet = variant == GASM_SCORE_VELVET;
    DevReads rd;
    ScoreTable tb;
    ScoreState ss;
    const u64 seg_off[2] = {0, n_reads};
    static const char empty = 0;
    int st = rd.upload(ctx, reads ? reads : &empty, read_off, n_reads, 0, seg_off, 1);
    if (st == GASM_OK) st = tb.set(ctx, bp_kmer, bp_off, n_table, bp_prob);
    if (st == GASM_OK) st = pipeline_score_launch(ctx, rd, dp, kmer, tb, !velvet && (flags & GASM_WANT_FREQ), velvet, ss, nullptr);
    if (st == GASM_OK) st = pipeline_score_fetch(ctx, ss);
    gasm_scores* s = nullptr;
    if (st == GASM_OK) {
        s = new gasm_scores();
        s->n = n_paths;
        s->velvet = velvet;
        s->len = ss.h_len; s->breaks = ss.h_breaks; s->bp = ss.h_bp; s->nf = ss.h_nf; s->nl = ss.h_nl;
        s->lev.assign(n_paths, 0);
        if (!velvet && (flags & GASM_WANT_FREQ)) { s->freq = ss.h_freq; s->has_freq = true; }
        if (velvet) {
            s->pd = ss.h_pd;
            s->pd_off = ss.h_pd_off;
            // lib/BreakageScorer.cpp:273-274: start of the path inside the true solution, taken only when a read matched
            s->startpos.assign(n_paths, 0);
            const std::string truth(true_solution ? true_solution : "", true_len);
            for (u64 p = 0; p < n_paths; ++p) {
                if (ss.h_breaks[p] <= 0) continue;
                s->startpos[p] = (int32_t)(int)truth.find(path_text(p));
            }
        }
        if (flags & GASM_WANT_KS) {
            // lib/DeNovoAssembler.R:414-424: ks.test(path_freq, kmer_from_seq)$statistic per path
            st = pipeline_ks(ctx, dp, ss, tb, true_solution ? true_solution : "", true_len, kmer, s->ks);
            s->has_ks = st == GASM_OK;
        }
        bool lev_done = false;
        // GPU or host?  One wave walks a path's bands column by column (~0.25 us per column and band, whatever the number
        // of paths up to a few thousand), the host routine costs ~1.5 ns per 64 cells and runs 32 paths at a time: a
        // handful of contigs is quicker on the host, thousands of scaffolds 10-60x quicker on the GPU.
        bool lev_gpu = !getenv("GASM_LEV_HOST");
        if (lev_gpu && !getenv("GASM_LEV_GPU") && (flags & GASM_WANT_LEV)) {
            u64 max_bands = 0;
            double cells = 0;
            for (u64 p = 0; p < n_paths; ++p) {
                const u64 nq = dp.h_p_off[p + 1] - dp.h_p_off[p];
                max_bands = std::max<u64>(max_bands, (nq + 4095) / 4096);
                cells += (double)nq * (double)true_len;
            }
            // measured (k_levenshtein, round 2): 0.24 us per column and band for a wave alone on its SIMD, 0.58 us with four waves per
            // SIMD; k_levenshtein2 needs 0.6 of its instructions
            const double per_simd = (double)n_paths / 1024.0;
            const double share = std::max(1.0, 0.6 * std::min(per_simd, 4.0)), rounds = std::max(1.0, per_simd / 4.0);
            const double gpu_ms = 0.05 + (double)max_bands * (double)(true_len + 63) * 0.00015 * share * rounds;
            const double host_ms = cells / 64.0 * 1.5e-6 / (double)std::max<u64>(1, std::min<u64>(32, n_paths));
            lev_gpu = gpu_ms < host_ms;
        }
        if (st == GASM_OK && (flags & GASM_WANT_LEV) && lev_gpu) {
            // lib/DeNovoAssembler.cpp:463 (global) / lib/BreakageScorer.cpp:339 (infix): one wave per path on the GPU
            st = pipeline_levenshtein(ctx, dp, true_solution, true_len, velvet, s->lev, &lev_done);
            if (lev_done) s->lev_device = 1;
        }
        if (st == GASM_OK && (flags & GASM_WANT_LEV) && !lev_done) {
            // target with bytes outside ACGT (or GASM_LEV_HOST set): the host routine, threads over paths
            s->lev_device = 2;
            std::atomic<u64> next(0);
            unsigned nt = std::thread::hardware_concurrency();
            nt = std::max(1u, std::min(nt, 32u));
            if (n_paths < 2) nt = 1;
            auto work = [&]() {
                while (true) {
                    const u64 p = next.fetch_add(1);
                    if (p >= n_paths) break;
                    const std::string q = path_text(p);
                    s->lev[p] = gasm_host::levenshtein(q.data(), q.size(), true_solution, true_len, velvet);
                }
            };
            if (nt == 1) work();
            else {
                std::vector<std::thread> th;
                for (unsigned t = 0; t < nt; ++t) th.emplace_back(work);
                for (auto& t : th) t.join();
            }
        }
    }
    rd.release(); tb.release(); ss.release();
    if (st != GASM_OK) { delete s; return st; }
    *out = s;
    return GASM_OK;
}

int gasm_calc_breakscore(gasm_ctx* ctx, const char* paths, const uint64_t* path_off, uint64_t n_paths, const char* reads,
                         const uint64_t* read_off, uint64_t n_reads, const char* true_solution, uint64_t true_len, int kmer,
                         const char* bp_kmer, const uint64_t* bp_off, uint64_t n_table, const double* bp_prob, int variant,
                         int flags, gasm_scores** out) {
    API_GUARD_BEGIN
    if (!ctx || !out || !path_off || !read_off || !bp_off || (n_table && (!bp_kmer || !bp_prob)) || (true_len && !true_solution)) {
        gasm_set_error("gasm_calc_breakscore: null argument");
        return GASM_ERR_INVALID;
    }
    if (variant != GASM_SCORE_OWN && variant != GASM_SCORE_VELVET) { gasm_set_error("unknown variant %d", variant); return GASM_ERR_INVALID; }
    if (n_paths > 0xFFFFFFF0ull) { gasm_set_error("too many paths"); return GASM_ERR_CAPACITY; }
    *out = nullptr;
    static const char empty = 0;
    DevPaths dp;
    int st = dp.upload_ascii(ctx, paths ? paths : &empty, path_off, (u32)n_paths);
    if (st == GASM_OK)
        st = breakscore_impl(ctx, dp, [&](u64 p) { return std::string(paths + path_off[p], paths + path_off[p + 1]); }, n_paths, reads, read_off, n_reads,
                             true_solution, true_len, kmer, bp_kmer, bp_off, n_table, bp_prob, variant, flags, out);
    dp.release();
    return st;
    API_GUARD_END
}

uint64_t gasm_scores_count(const gasm_scores* s) { return s ? s->n : 0; }
const int32_t* gasm_scores_sequence_len(const gasm_scores* s) { return s ? s->len.data() : nullptr; }
const double* gasm_scores_bp_score(const gasm_scores* s) { return s ? s->bp.data() : nullptr; }
const double* gasm_scores_norm_by_break_freqs(const gasm_scores* s) { return s ? s->nf.data() : nullptr; }
const double* gasm_scores_norm_by_len(const gasm_scores* s) { return s ? s->nl.data() : nullptr; }
const int32_t* gasm_scores_kmer_breaks(const gasm_scores* s) { return s ? s->breaks.data() : nullptr; }
const int32_t* gasm_scores_lev_dist(const gasm_scores* s) { return s ? s->lev.data() : nullptr; }
const double* gasm_scores_path_freq(const gasm_scores* s) { return s && s->has_freq ? s->freq.data() : nullptr; }
const int32_t* gasm_scores_startpos(const gasm_scores* s) { return s && s->velvet ? s->startpos.data() : nullptr; }
const double* gasm_scores_prob_dist(const gasm_scores* s) { return s && s->velvet ? s->pd.data() : nullptr; }
const uint64_t* gasm_scores_prob_dist_offsets(const gasm_scores* s) { return s && s->velvet ? s->pd_off.data() : nullptr; }
const double* gasm_scores_ks(const gasm_scores* s) { return s && s->has_ks ? s->ks.data() : nullptr; }
int gasm_scores_lev_device(const gasm_scores* s) { return s ? s->lev_device : 0; }

int gasm_coverage_percent(gasm_ctx* ctx, const int64_t* start, const int64_t* len, uint64_t n, int64_t seq_len, double* percent) {
    API_GUARD_BEGIN
    if (!ctx || !percent || (n && (!start || !len))) { gasm_set_error("gasm_coverage_percent: null argument"); return GASM_ERR_INVALID; }
    return pipeline_coverage(ctx, reinterpret_cast<const long long*>(start), reinterpret_cast<const long long*>(len), n, (long long)seq_len, percent);
    API_GUARD_END
}
void gasm_scores_free(gasm_scores* s) { delete s; }

// ------------------------------------------------------------------------------------------------------ batches
static u32 sub_batches_for(u32 n_segments) {
    // GASM_SUBBATCHES=n forces the number of blocks (1 = the whole batch on the context's own stream)
    if (const char* v = getenv("GASM_SUBBATCHES")) return (u32)std::max(1, std::min(8, atoi(v)));
    (void)n_segments;
    return 1u;      // measured (cfg2, 2-4 blocks): 1.32-1.35 ms per step against 1.34 — the half-size streaming kernels lose what the overlap gains
}

int gasm_batch_create(gasm_ctx* ctx, const char* reads, const uint64_t* read_off, uint64_t n_reads, uint32_t fixed_len,
                      const uint64_t* seg_read_off, uint32_t n_segments, gasm_batch** out) {
    API_GUARD_BEGIN
    if (!ctx || !out) { gasm_set_error("gasm_batch_create: null argument"); return GASM_ERR_INVALID; }
    *out = nullptr;
    if (n_segments == 0 || !seg_read_off) { gasm_set_error("need at least one segment and seg_read_off"); return GASM_ERR_INVALID; }
    if (seg_read_off[0] != 0 || seg_read_off[n_segments] != n_reads) { gasm_set_error("seg_read_off must run from 0 to n_reads"); return GASM_ERR_INVALID; }
    for (u32 s = 0; s < n_segments; ++s) if (seg_read_off[s] > seg_read_off[s + 1]) { gasm_set_error("seg_read_off not monotone"); return GASM_ERR_INVALID; }
    gasm_batch* b = new gasm_batch();
    b->ctx = ctx;
    b->n_segments = n_segments;
    b->n_reads = n_reads;
    const u32 nsub = std::min<u32>(sub_batches_for(n_segments), n_segments);
    b->sub.resize(nsub);
    // blocks of segments with about the same number of reads each
    u32 seg = 0;
    int st = GASM_OK;
    for (u32 j = 0; j < nsub && st == GASM_OK; ++j) {
        SubBatch& sb = b->sub[j];
        sb.seg0 = seg;
        const u64 want = n_reads * (u64)(j + 1) / nsub;
        u32 e = seg + 1;
        while (e < n_segments - (nsub - 1 - j) && seg_read_off[e] < want) ++e;
        if (j + 1 == nsub) e = n_segments;
        sb.seg1 = e;
        seg = e;
        sb.cx = j == 0 ? ctx : ctx->lane(j - 1);
        if (!sb.cx) { st = GASM_ERR_HIP; break; }
        std::vector<u64> so(sb.seg1 - sb.seg0 + 1);
        const u64 r0 = seg_read_off[sb.seg0];
        for (u32 i = 0; i < so.size(); ++i) so[i] = seg_read_off[sb.seg0 + i] - r0;
        const char* rbase = reads;
        const u64* ro = nullptr;
        if (read_off) ro = read_off + r0;            // DevReads::upload rebases ragged offsets to ro[0]
        else rbase = reads ? reads + r0 * (u64)fixed_len : reads;
        st = sb.rd.upload(sb.cx, rbase, ro, so.back(), fixed_len, so.data(), (u32)so.size() - 1);
        if (st == GASM_OK && nsub > 1 && hipEventCreateWithFlags(&sb.ev_streamed, hipEventDisableTiming) != hipSuccess) {
            gasm_set_error("hipEventCreate failed");
            st = GASM_ERR_HIP;
        }
    }
    if (st != GASM_OK) { gasm_batch_free(b); return st; }
    for (u32 j = 0; j < nsub; ++j) {
        b->sub[j].slot[0].bs.ev_streamed = b->sub[j].ev_streamed;
        b->sub[j].slot[0].bs.ev_wait = j ? b->sub[j - 1].ev_streamed : nullptr;
    }
    *out = b;
    return GASM_OK;
    API_GUARD_END
}

// one block on the context's own stream, from reads that are packed already
static int batch_from_packed(gasm_ctx* ctx, const u64* words, const u64* read_off, u64 n_reads, u32 fixed_len, const u64* seg_read_off,
                             u32 n_segments, gasm_batch** out) {
    gasm_batch* b = new gasm_batch();
    b->ctx = ctx;
    b->n_segments = n_segments;
    b->n_reads = n_reads;
    b->sub.resize(1);
    SubBatch& sb = b->sub[0];
    sb.cx = ctx; sb.seg0 = 0; sb.seg1 = n_segments;
    const int st = sb.rd.upload_packed(ctx, words, read_off, n_reads, fixed_len, seg_read_off, n_segments);
    if (st != GASM_OK) { gasm_batch_free(b); return st; }
    *out = b;
    return GASM_OK;
}

int gasm_batch_create_packed(gasm_ctx* ctx, const uint64_t* words, const uint64_t* read_off, uint64_t n_reads, uint32_t fixed_len,
                             const uint64_t* seg_read_off, uint32_t n_segments, gasm_batch** out) {
    API_GUARD_BEGIN
    if (!ctx || !out) { gasm_set_error("gasm_batch_create_packed: null argument"); return GASM_ERR_INVALID; }
    *out = nullptr;
    return batch_from_packed(ctx, words, read_off, n_reads, fixed_len, seg_read_off, n_segments, out);
    API_GUARD_END
}

static int parse_files(const char* const* paths, uint32_t n_files, int on_non_acgt, gasm_packed& g) {
    g.pr.read_off.assign(1, 0);
    g.seg.assign((size_t)n_files + 1, 0);
    g.dropped = 0;
    for (u32 f = 0; f < n_files; ++f) {
        if (!paths[f]) { gasm_set_error("paths[%u] is null", f); return GASM_ERR_INVALID; }
        u64 kept = 0;
        GCHK(gasm_host::read_sequence_file(paths[f], on_non_acgt != 0, g.pr, &kept, &g.dropped));
        g.seg[f + 1] = g.seg[f] + kept;
    }
    return GASM_OK;
}

int gasm_read_files(const char* const* paths, uint32_t n_files, int on_non_acgt, gasm_packed** out) {
    API_GUARD_BEGIN
    if (!out || !paths || n_files == 0) { gasm_set_error("gasm_read_files: bad argument"); return GASM_ERR_INVALID; }
    *out = nullptr;
    gasm_packed* g = new gasm_packed();
    const int st = parse_files(paths, n_files, on_non_acgt, *g);
    if (st != GASM_OK) { delete g; return st; }
    *out = g;
    return GASM_OK;
    API_GUARD_END
}
uint64_t gasm_packed_n_reads(const gasm_packed* g) { return g ? g->pr.read_off.size() - 1 : 0; }
uint32_t gasm_packed_n_segments(const gasm_packed* g) { return g ? (uint32_t)(g->seg.size() - 1) : 0; }
const uint64_t* gasm_packed_words(const gasm_packed* g) { return g ? g->pr.words.data() : nullptr; }
const uint64_t* gasm_packed_read_off(const gasm_packed* g) { return g ? g->pr.read_off.data() : nullptr; }
const uint64_t* gasm_packed_seg_read_off(const gasm_packed* g) { return g ? g->seg.data() : nullptr; }
uint64_t gasm_packed_dropped(const gasm_packed* g) { return g ? g->dropped : 0; }
void gasm_packed_free(gasm_packed* g) { delete g; }

// the same files through the device path (ingest.hip: record scan + 2-bit pack on the GPU, the host only inflates), results
// copied back: what gasm_batch_from_files builds its batch from without the copy
int gasm_read_files_device(gasm_ctx* ctx, const char* const* paths, uint32_t n_files, int on_non_acgt, gasm_packed** out) {
    API_GUARD_BEGIN
    if (!ctx || !out || !paths || n_files == 0) { gasm_set_error("gasm_read_files_device: bad argument"); return GASM_ERR_INVALID; }
    *out = nullptr;
    gasm_packed* g = new gasm_packed();
    DBuf d_words;
    struct Rel { DBuf& b; ~Rel() { b.release(); } } rel{d_words};
    const int st = ingest_files_device(ctx, paths, n_files, on_non_acgt != 0, d_words, g->pr.read_off, g->seg, &g->dropped, g->on_device);
    if (st != GASM_OK) { delete g; return st; }
    g->pr.total_bases = g->pr.read_off.back();
    const u64 nw = (g->pr.total_bases + 31) / 32;
    g->pr.words.resize(nw);
    if (nw && hipMemcpy(g->pr.words.data(), d_words.p, nw * 8, hipMemcpyDeviceToHost) != hipSuccess) { delete g; gasm_set_error("copy of the packed reads failed"); return GASM_ERR_HIP; }
    *out = g;
    return GASM_OK;
    API_GUARD_END
}
int gasm_packed_parsed_on_device(const gasm_packed* g, uint32_t file) { return g && file < g->on_device.size() ? g->on_device[file] : 0; }

int gasm_batch_from_files(gasm_ctx* ctx, const char* const* paths, uint32_t n_files, int on_non_acgt, gasm_batch** out, uint64_t* dropped_reads) {
    API_GUARD_BEGIN
    if (!ctx || !out || !paths || n_files == 0) { gasm_set_error("gasm_batch_from_files: bad argument"); return GASM_ERR_INVALID; }
    *out = nullptr;
    // record scan and 2-bit packing on the device (ingest.hip); the packed stream never leaves it
    DBuf d_words;
    struct Rel { DBuf& b; ~Rel() { b.release(); } } rel{d_words};
    std::vector<u64> read_off, seg;
    std::vector<u8> on_device;
    u64 dropped = 0;
    GCHK(ingest_files_device(ctx, paths, n_files, on_non_acgt != 0, d_words, read_off, seg, &dropped, on_device));
    if (dropped_reads) *dropped_reads = dropped;
    const u64 n = read_off.size() - 1;
    // fixed-length reads (the usual case) need no offset array on the device
    u32 flen = n ? (u32)std::min<u64>(read_off[1], 0xFFFFFFFFull) : 0;
    bool fixed = n > 0 && flen > 0;
    for (u64 r = 0; fixed && r < n; ++r) fixed = read_off[r + 1] - read_off[r] == flen;
    gasm_batch* b = new gasm_batch();
    b->ctx = ctx;
    b->n_segments = n_files;
    b->n_reads = n;
    b->sub.resize(1);
    SubBatch& sb = b->sub[0];
    sb.cx = ctx; sb.seg0 = 0; sb.seg1 = n_files;
    const int st = sb.rd.adopt_packed(ctx, d_words, fixed ? nullptr : read_off.data(), n, fixed ? flen : 0, seg.data(), n_files);
    if (st != GASM_OK) { gasm_batch_free(b); return st; }
    *out = b;
    return GASM_OK;
    API_GUARD_END
}

int gasm_batch_simulate(gasm_ctx* ctx, const char* genomes, const uint64_t* genome_off, uint32_t n_segments, uint32_t read_len, double coverage,
                        uint64_t seed, int kmer, const double* table, gasm_batch** out) {
    API_GUARD_BEGIN
    if (!ctx || !out || !genomes || !genome_off) { gasm_set_error("gasm_batch_simulate: null argument"); return GASM_ERR_INVALID; }
    *out = nullptr;
    gasm_batch* b = new gasm_batch();
    b->ctx = ctx;
    b->n_segments = n_segments;
    b->sub.resize(1);
    SubBatch& sb = b->sub[0];
    sb.cx = ctx; sb.seg0 = 0; sb.seg1 = n_segments;
    const int st = sb.rd.simulate(ctx, genomes, genome_off, n_segments, read_len, coverage, seed, kmer, table, b->d_read_start);
    if (st != GASM_OK) { gasm_batch_free(b); return st; }
    b->n_reads = sb.rd.n_reads;
    b->h_sim_seg_off = sb.rd.h_seg_read_off;
    *out = b;
    return GASM_OK;
    API_GUARD_END
}

int gasm_batch_fetch_read_starts(gasm_batch* b, const uint64_t** seg_read_off, const uint32_t** starts) {
    API_GUARD_BEGIN
    if (!b || !seg_read_off || !starts) { gasm_set_error("null argument"); return GASM_ERR_INVALID; }
    if (b->h_sim_seg_off.empty()) { gasm_set_error("not a simulated batch"); return GASM_ERR_STATE; }
    b->h_read_start.resize(b->n_reads);
    HIPCHK(hipSetDevice(b->ctx->device));
    if (b->n_reads) HIPCHK(hipMemcpyAsync(b->h_read_start.data(), b->d_read_start.p, b->n_reads * 4, hipMemcpyDeviceToHost, b->ctx->stream));
    HIPCHK(hipStreamSynchronize(b->ctx->stream));
    *seg_read_off = b->h_sim_seg_off.data();
    *starts = b->h_read_start.data();
    return GASM_OK;
    API_GUARD_END
}

void gasm_batch_free(gasm_batch* b) {
    if (!b) return;
    b->d_read_start.release();
    for (SubBatch& sb : b->sub) {
        if (sb.cx) { (void)hipSetDevice(sb.cx->device); (void)hipStreamSynchronize(sb.cx->stream); }
        if (sb.scx) (void)hipStreamSynchronize(sb.scx->stream);
        if (sb.ev_built) (void)hipEventDestroy(sb.ev_built);
        if (sb.ev_scored) (void)hipEventDestroy(sb.ev_scored);
        for (StepSlot& x : sb.slot) { if (x.cx && x.cx != sb.cx) (void)hipStreamSynchronize(x.cx->stream); if (x.ev_streamed) (void)hipEventDestroy(x.ev_streamed); x.bs.release(); x.dp.release(); x.ss.release(); }
        sb.rd.release(); sb.tb.release(); sb.guided.release();
        if (sb.ev_streamed) (void)hipEventDestroy(sb.ev_streamed);
    }
    delete b;
}

int gasm_batch_build(gasm_batch* b, int k, uint64_t genome_len_hint) {
    API_GUARD_BEGIN
    if (!b) { gasm_set_error("batch is null"); return GASM_ERR_INVALID; }
    b->built = false; b->scored = false;
    for (SubBatch& sb : b->sub) {
        // consecutive steps take the two slots in turn (one-block batches): this build does not wait for the last step's graph
        // and scoring, it runs beside them.  A change of k rewrites the tile tables both slots read: everything drains first.
        const int pp = env_int("GASM_PINGPONG", 1);
        sb.pingpong = b->sub.size() == 1 && pp != 0;
        if (sb.pingpong && sb.slot[0].cx == nullptr && sb.slot[1].cx == nullptr) {       // (fixed with the first build)
            sb.pp_mode = pp;
            sb.n_slots = std::max(2, std::min(4, env_int("GASM_STEP_SLOTS", 3)));
        }
        if (sb.pingpong) {
            if (b->last_k && b->last_k != k) for (StepSlot& x : sb.slot) if (x.cx) HIPCHK(hipStreamSynchronize(x.cx->stream));
            sb.cur = (sb.cur + 1) % sb.n_slots;
        }
        StepSlot& st = sb.S();
        st.paths_ready = false; st.ss.valid = false; st.ss.launched = false;
        if (sb.pingpong) {
            // the streaming kernels of consecutive steps take turns (two of them at once only share the HBM they are both bound by):
            // this build's partition waits for the other slot's de-duplication, and runs beside that slot's graph and scoring
            StepSlot& other = sb.slot[(sb.cur + sb.n_slots - 1) % sb.n_slots];
            const bool chain = sb.pp_mode != 2 && env_flag("GASM_PINGPONG_CHAIN", false);
            st.bs.stream_ctx = sb.pp_mode == 2 ? sb.cx : nullptr;
            if (chain && !st.ev_streamed && hipEventCreateWithFlags(&st.ev_streamed, hipEventDisableTiming) != hipSuccess) st.ev_streamed = nullptr;
            st.bs.ev_streamed = chain ? st.ev_streamed : nullptr;
            st.bs.ev_wait = chain ? other.ev_streamed : nullptr;      // (never recorded yet: no wait)
        }
        const bool lane = !sb.pingpong && score_lane(b, sb);
        const bool lane_before = sb.lane_last;      // (the lane switched off between two builds: its last scoring is still waited for)
        sb.lane_last = lane;
        st.bs.ev_before_dedup = (lane || lane_before) ? sb.ev_scored : nullptr;      // the last step's scoring still searches the directories this build's de-duplication rewrites
        GCHK(pipeline_build(st.cx, sb.rd, k, genome_len_hint, st.bs));
        if (lane) HIPCHK(hipEventRecord(sb.ev_built, st.cx->stream));
    }
    b->last_k = k;
    b->built = true;
    return GASM_OK;
    API_GUARD_END
}

int gasm_batch_score(gasm_batch* b, int kmer, const double* table) {
    API_GUARD_BEGIN
    if (!b || !table) { gasm_set_error("gasm_batch_score: null argument"); return GASM_ERR_INVALID; }
    if (!b->built) { gasm_set_error("gasm_batch_score before gasm_batch_build"); return GASM_ERR_STATE; }
    const bool new_table = !b->table_given || memcmp(b->table_copy.data(), table, GASM_TABLE_ROWS * sizeof(double)) != 0;
    if (new_table) {
        b->table_copy.assign(table, table + GASM_TABLE_ROWS);
        b->table_given = true;
    }
    for (SubBatch& sb : b->sub) {
        if (new_table || !sb.table_set) {
            for (StepSlot& x : sb.slot) if (x.cx) HIPCHK(hipStreamSynchronize(x.cx->stream));      // (whatever still scores with the old table)
            GCHK(sb.tb.set_standard(sb.cx, table));
            sb.table_set = true;
        }
        // reads shorter than k (or none): the general scorer, which sizes its tables on the host — after the build's report
        const bool through_graph = pipeline_score_uses_graph(sb.rd, sb.S().bs);
        if (!through_graph) GCHK(sub_finish(b, sb));
        // through the graph: on the lane, behind the build (its queue, not its completion: stream order does the rest)
        gasm_ctx* const own = sb.S().cx;
        gasm_ctx* const cx = (sb.lane_last && through_graph && !getenv("GASM_SCORE_VERIFY")) ? sb.scx : own;
        if (cx != own) HIPCHK(hipStreamWaitEvent(cx->stream, sb.ev_built, 0));
        else if (sb.scx) HIPCHK(hipStreamSynchronize(sb.scx->stream));
        if (!sb.S().paths_ready) {
            GCHK(pipeline_contig_paths(cx, sb.rd, sb.S().bs, sb.S().dp));
            sb.S().paths_ready = true;
        }
        if (!through_graph) pipeline_contig_paths_host(sb.rd, sb.S().bs, sb.S().dp);
        GCHK(pipeline_score_launch(cx, sb.rd, sb.S().dp, kmer, sb.tb, false, false, sb.S().ss, &sb.S().bs));
        sb.score_cx_last = cx;
        if (sb.ev_scored) HIPCHK(hipEventRecord(sb.ev_scored, cx->stream));
    }
    b->scored = true;
    b->score_kmer = kmer;
    return GASM_OK;
    API_GUARD_END
}

// ---- row A16: breakage-score-guided traversal of a built + scored batch
int gasm_batch_guided(gasm_batch* b) {
    API_GUARD_BEGIN
    if (!b) { gasm_set_error("batch is null"); return GASM_ERR_INVALID; }
    if (!b->built || !b->scored) { gasm_set_error("gasm_batch_guided needs gasm_batch_build and gasm_batch_score first"); return GASM_ERR_STATE; }
    if (b->sub.size() != 1) { gasm_set_error("gasm_batch_guided: batches split into sub-batches are not supported"); return GASM_ERR_STATE; }
    GCHK(batch_finish(b));
    SubBatch& sb = b->sub[0];
    GCHK(pipeline_score_fetch(sb.score_cx_last ? sb.score_cx_last : sb.S().cx, sb.S().ss));
    return guided_build(sb.S().cx, sb.rd, sb.S().bs, sb.S().dp, sb.S().ss, sb.tb, b->score_kmer, sb.guided);
    API_GUARD_END
}

int gasm_batch_fetch_guided(gasm_batch* b, const uint64_t** seg_off, const uint64_t** off, const char** data, const double** bp_score,
                            const double** norm_by_len, const int32_t** kmer_breaks) {
    API_GUARD_BEGIN
    if (!b || !seg_off || !off || !data || !bp_score || !norm_by_len || !kmer_breaks) { gasm_set_error("null argument"); return GASM_ERR_INVALID; }
    if (b->sub.size() != 1 || !b->sub[0].guided.valid) { gasm_set_error("fetch before gasm_batch_guided"); return GASM_ERR_STATE; }
    GuidedState& g = b->sub[0].guided;
    GCHK(guided_fetch_text(b->sub[0].S().cx, g));
    *seg_off = g.h_seg_off.data(); *off = g.h_text_off.data(); *data = g.h_text.data();
    *bp_score = g.ss.h_bp.data(); *norm_by_len = g.ss.h_nl.data(); *kmer_breaks = g.ss.h_breaks.data();
    return GASM_OK;
    API_GUARD_END
}

// the fixed-point breakage sums behind the last gasm_batch_score: bp_score[c] = fx[c] * 2^-shift exactly (what the guided
// traversal compares, and what its CPU restatement recomputes)
int gasm_batch_fetch_score_fixed(gasm_batch* b, const int64_t** fx, int* shift) {
    API_GUARD_BEGIN
    if (!b || !fx || !shift) { gasm_set_error("null argument"); return GASM_ERR_INVALID; }
    if (b->sub.size() != 1 || !b->scored) { gasm_set_error("gasm_batch_fetch_score_fixed needs a scored, unsplit batch"); return GASM_ERR_STATE; }
    GCHK(batch_finish(b));
    SubBatch& sb = b->sub[0];
    if (!sb.S().ss.graph) { gasm_set_error("the batch was not scored through its graph"); return GASM_ERR_STATE; }
    const u32 P = sb.S().bs.n_contigs;
    b->h_fx.resize(P);
    const size_t fx_off = (sb.S().ss.stride * 4 + 15) & ~(size_t)15;
    HIPCHK(hipSetDevice(sb.cx->device));
    if (P) HIPCHK(hipMemcpyAsync(b->h_fx.data(), static_cast<const char*>(sb.S().ss.d_total.p) + fx_off, (size_t)P * 8, hipMemcpyDeviceToHost, sb.S().cx->stream));
    HIPCHK(hipStreamSynchronize(sb.S().cx->stream));
    *fx = b->h_fx.data();
    *shift = sb.tb.fix_shift;
    return GASM_OK;
    API_GUARD_END
}

uint64_t gasm_batch_total_kmers(const gasm_batch* b) {
    u64 n = 0;
    if (b) for (const SubBatch& sb : b->sub) n += sb.S().bs.n_kmers;
    return n;
}
uint64_t gasm_batch_total_reads(const gasm_batch* b) { return b ? b->n_reads : 0; }

int gasm_batch_fetch_distinct(gasm_batch* b, const uint64_t** seg_off, const uint64_t** keys, const uint32_t** mult, int* words) {
    API_GUARD_BEGIN
    if (!b || !seg_off || !keys || !mult || !words) { gasm_set_error("null argument"); return GASM_ERR_INVALID; }
    if (!b->built) { gasm_set_error("fetch before build"); return GASM_ERR_STATE; }
    GCHK(batch_finish(b));
    for (SubBatch& sb : b->sub) GCHK(pipeline_fetch_distinct(sb.S().cx, sb.rd, sb.S().bs));
    *words = b->sub[0].S().bs.words;
    if (b->sub.size() == 1) {
        BuildState& bs = b->sub[0].S().bs;
        *seg_off = bs.h_seg_doff.data(); *keys = bs.h_dk_key.data(); *mult = bs.h_dk_cnt.data();
        return GASM_OK;
    }
    b->h_seg_doff.assign((size_t)b->n_segments + 1, 0);
    b->h_dk_key.clear(); b->h_dk_cnt.clear();
    u64 base = 0;
    for (SubBatch& sb : b->sub) {
        for (u32 s = sb.seg0; s <= sb.seg1; ++s) b->h_seg_doff[s] = base + sb.S().bs.h_seg_doff[s - sb.seg0];
        base += sb.S().bs.d_total;
        b->h_dk_key.insert(b->h_dk_key.end(), sb.S().bs.h_dk_key.begin(), sb.S().bs.h_dk_key.end());
        b->h_dk_cnt.insert(b->h_dk_cnt.end(), sb.S().bs.h_dk_cnt.begin(), sb.S().bs.h_dk_cnt.end());
    }
    *seg_off = b->h_seg_doff.data(); *keys = b->h_dk_key.data(); *mult = b->h_dk_cnt.data();
    return GASM_OK;
    API_GUARD_END
}

int gasm_batch_fetch_graph(gasm_batch* b, const uint8_t** edge_flags, const uint32_t** edge_next) {
    API_GUARD_BEGIN
    if (!b || !edge_flags || !edge_next) { gasm_set_error("null argument"); return GASM_ERR_INVALID; }
    if (!b->built) { gasm_set_error("fetch before build"); return GASM_ERR_STATE; }
    GCHK(batch_finish(b));
    for (SubBatch& sb : b->sub) GCHK(pipeline_fetch_graph(sb.S().cx, sb.rd, sb.S().bs));
    if (b->sub.size() == 1) {
        *edge_flags = b->sub[0].S().bs.h_eflag.data(); *edge_next = b->sub[0].S().bs.h_nxt.data();
        return GASM_OK;
    }
    b->h_eflag.clear(); b->h_nxt.clear();
    u32 base = 0;
    for (SubBatch& sb : b->sub) {
        b->h_eflag.insert(b->h_eflag.end(), sb.S().bs.h_eflag.begin(), sb.S().bs.h_eflag.end());
        for (u32 v : sb.S().bs.h_nxt) b->h_nxt.push_back(v == 0xFFFFFFFFu ? v : v + base);
        base += sb.S().bs.d_total;
    }
    *edge_flags = b->h_eflag.data(); *edge_next = b->h_nxt.data();
    return GASM_OK;
    API_GUARD_END
}

int gasm_batch_fetch_contigs(gasm_batch* b, const uint64_t** seg_contig_off, const uint64_t** off, const char** data) {
    API_GUARD_BEGIN
    if (!b || !seg_contig_off || !off || !data) { gasm_set_error("null argument"); return GASM_ERR_INVALID; }
    if (!b->built) { gasm_set_error("fetch before build"); return GASM_ERR_STATE; }
    GCHK(batch_finish(b));
    for (SubBatch& sb : b->sub) GCHK(pipeline_fetch_contigs(sb.S().cx, sb.rd, sb.S().bs));
    if (b->sub.size() == 1) {
        BuildState& bs = b->sub[0].S().bs;
        *seg_contig_off = bs.h_seg_coff.data(); *off = bs.h_c_off.data(); *data = bs.h_contigs.data();
        return GASM_OK;
    }
    b->h_seg_coff.assign((size_t)b->n_segments + 1, 0);
    b->h_c_off.clear(); b->h_contigs.clear();
    u64 cbase = 0, bbase = 0;
    for (SubBatch& sb : b->sub) {
        for (u32 s = sb.seg0; s <= sb.seg1; ++s) b->h_seg_coff[s] = cbase + sb.S().bs.h_seg_coff[s - sb.seg0];
        for (u32 c = 0; c < sb.S().bs.n_contigs; ++c) b->h_c_off.push_back(bbase + sb.S().bs.h_c_off[c]);
        cbase += sb.S().bs.n_contigs;
        bbase += sb.S().bs.contig_bases;
        b->h_contigs.insert(b->h_contigs.end(), sb.S().bs.h_contigs.begin(), sb.S().bs.h_contigs.end());
    }
    b->h_c_off.push_back(bbase);
    *seg_contig_off = b->h_seg_coff.data(); *off = b->h_c_off.data(); *data = b->h_contigs.data();
    return GASM_OK;
    API_GUARD_END
}

int gasm_batch_fetch_scores(gasm_batch* b, const double** bp_score, const double** norm_by_break_freqs, const double** norm_by_len,
                            const int32_t** kmer_breaks, const int32_t** sequence_len) {
    API_GUARD_BEGIN
    if (!b || !bp_score || !norm_by_break_freqs || !norm_by_len || !kmer_breaks || !sequence_len) { gasm_set_error("null argument"); return GASM_ERR_INVALID; }
    GCHK(batch_finish(b));
    for (SubBatch& sb : b->sub) GCHK(pipeline_score_fetch(sb.score_cx_last ? sb.score_cx_last : sb.S().cx, sb.S().ss));
    if (b->sub.size() == 1) {
        ScoreState& ss = b->sub[0].S().ss;
        *bp_score = ss.h_bp.data(); *norm_by_break_freqs = ss.h_nf.data(); *norm_by_len = ss.h_nl.data();
        *kmer_breaks = ss.h_breaks.data(); *sequence_len = ss.h_len.data();
        return GASM_OK;
    }
    b->h_bp.clear(); b->h_nf.clear(); b->h_nl.clear(); b->h_breaks.clear(); b->h_len.clear();
    for (SubBatch& sb : b->sub) {
        ScoreState& ss = sb.S().ss;
        b->h_bp.insert(b->h_bp.end(), ss.h_bp.begin(), ss.h_bp.end());
        b->h_nf.insert(b->h_nf.end(), ss.h_nf.begin(), ss.h_nf.end());
        b->h_nl.insert(b->h_nl.end(), ss.h_nl.begin(), ss.h_nl.end());
        b->h_breaks.insert(b->h_breaks.end(), ss.h_breaks.begin(), ss.h_breaks.end());
        b->h_len.insert(b->h_len.end(), ss.h_len.begin(), ss.h_len.end());
    }
    *bp_score = b->h_bp.data(); *norm_by_break_freqs = b->h_nf.data(); *norm_by_len = b->h_nl.data();
    *kmer_breaks = b->h_breaks.data(); *sequence_len = b->h_len.data();
    return GASM_OK;
    API_GUARD_END
}

}  // extern "C"
