// exchange.hip — the pooled (multi-GPU) step with its all-to-alls INSIDE the library: gasm_comm_* and gasm_pool_exchange_build
// of include/gasm.h.  SURVEY §8(e) mode 2 / the north star's "RCCL all-to-all over xGMI to bucket k-mers by hash before the
// global edge-list merge"; the reference has no counterpart (it loops over segments on one thread,
// scripts/02_Real_vs_rand_prob_own.R:33-53).
//
// Round 2 drove the same stages from Python (genomeassembler_dev_amd/pooled.py): run lengths went to the host after every
// stage, numpy computed who gets what and where it lands, torch.distributed moved the buffers — a dozen host syncs per step
// and a second HIP runtime (PyTorch's) in the process.  Here the run-length tables stay on the device, the plans (offsets,
// per-peer totals, run directories of the merges) are kernels (kernels_pool.hip, k_x1_plan / k_x2_plan), and every exchange
// is one ncclGroupStart ... ncclSend / ncclRecv ... ncclGroupEnd on the context's stream: on a fully connected xGMI node
// every pair of ranks has its own link, so the grouped point-to-point form is link-parallel (no ring).  What the host still
// waits for is ONE small pinned report per record exchange — the per-peer totals, because ncclSend / ncclRecv take their
// counts from the host — and nothing at all for the third exchange (the reads: their sizes are known when the pools are set up).
//
//   stage 10  local runs            reads -> k-mers -> sorted distinct (key, count) run per (segment, bucket)      [launch_distinct]
//   stage 11  all-gather            every rank's run lengths + its overflow flag; plan 1; report 1 (host wait #1)
//   stage 12  all-to-all #1         runs -> bucket_owner(segment, prefix): keys and counts in one group
//   stage 13  merge                 k_bucket_merge: one run per owned bucket, counts added
//   stage 21  all-reduce            merged lengths of all buckets + overflow flags; plan 2; report 2 (host wait #2)
//   stage 22  all-to-all #2         merged runs -> segment_owner(segment)
//   stage 23  placement + graph     k_bucket_merge (one source per bucket: copy + fine directory), graph, contigs
//   stage 31  all-to-all #3         the segment's reads (2-bit, word-aligned pieces) -> segment_owner(segment)
//   stage 32  scoring               queued behind the graph
// Capacity failures are collective by construction: the overflow flags travel with the length tables, so every rank sees
// every rank's flags in the same report and takes the same step of the same ladder (two-pass partition, larger tables, two
// more bucket bits, GASM_ERR_CAPACITY) — no rank can walk into the next all-to-all alone.
//
// Two communicators behind the same code: RCCL (one rank per process; librccl is loaded on first use, so single-GPU users
// never touch it) and a virtual one (W ranks of one process on one GPU, the exchanges are device copies on the same stream):
// the virtual form is what the one-GPU test box can run, and it runs everything except the ncclSend / ncclRecv calls
// themselves.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>

#include "pool.h"

// ---------------------------------------------------------------------------------------------------------------
// librccl, resolved at first use.  "librccl.so.1" is also the soname of the copy PyTorch ships: in a process that has
// imported torch the loader hands back that copy (and libgasm's libamdhip64.so.7 is torch's copy too, _lib.py), otherwise
// ROCm's own.
// ---------------------------------------------------------------------------------------------------------------
struct RcclApi {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
};

static const RcclApi* rccl_api() {
    static RcclApi api;
    static bool tried = false, ok = false;
    if (!tried) {
        tried = true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.h) break;
        }
        if (!api.h) { gasm_set_error("librccl.so.1 could not be loaded: %s", dlerror()); return nullptr; }
        bool all = true;
        auto sym = [&](const char* n) { void* p = dlsym(api.h, n); if (!p) all = false; return p; };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
        api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
        api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
        ok = all;
        if (!ok) gasm_set_error("librccl lacks a symbol this library needs");
    }
    return ok ? &api : nullptr;
}

#define NCHK(expr)                                                                                                          \
    do {                                                                                                                    \
        ncclResult_t _r = (expr);                                                                                           \
        if (_r != ncclSuccess) {                                                                                            \
            gasm_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, rccl_api()->GetErrorString(_r));              \
            return GASM_ERR_HIP;                                                                                            \
        }                                                                                                                   \
    } while (0)

// ---------------------------------------------------------------------------------------------------------------
// ownership: pure functions of (segment, prefix, world size), the same on every rank (restated from round 2's pooled.py,
// which keeps its numpy form for the CPU tests: tests/test_cabi_and_host.py compares the two)
// ---------------------------------------------------------------------------------------------------------------
static inline u32 bucket_owner_of(u64 seg, u64 pre, u32 world) {
    u64 h = seg * 0x9E3779B97F4A7C15ull + pre * 0xC2B2AE3D27D4EB4Full + 0x165667B19E3779F9ull;
    h ^= h >> 29;
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 32;
    return (u32)(h % world);
}
static void segment_bounds(u32 S, u32 W, std::vector<u32>& first) {
    first.assign((size_t)W + 1, 0);
    const u32 base = S / W, extra = S % W;
    for (u32 r = 0; r < W; ++r) first[r + 1] = first[r] + base + (r < extra ? 1u : 0u);
}

// ---------------------------------------------------------------------------------------------------------------
// communicator
// ---------------------------------------------------------------------------------------------------------------
struct RankX {                 // exchange state of one local rank
    DBuf d_mine, d_lens_all, d_send_off, d_send_tot, d_recv_tot, d_run_off, d_run_len, d_bstart_new, d_flags_or, d_info, d_src_base, d_G;
    DBuf send_keys, send_cnt, recv_keys, recv_cnt;
    DBuf d_nr, d_nr_all, d_rdir, send_words, recv_words;
    DBuf d_part_saved;         // the partition's region layout (BuildState::d_bstart) while the merges use that slot
    bool part_saved = false;
    u64* rep = nullptr;        // pinned: two reports of 2 W + 6 words
    u64* h_base = nullptr;     // pinned: W source bases (uploaded asynchronously)
    // reads plan (exchange 3): valid for reads_id
    u64 reads_id = 0;
    bool reads_ready = false;
    std::vector<u64> send_woff, recv_woff;     // W + 1 word offsets
    u32 max_piece_words = 0;
    void release() {
        for (DBuf* b : {&d_mine, &d_lens_all, &d_send_off, &d_send_tot, &d_recv_tot, &d_run_off, &d_run_len, &d_bstart_new, &d_flags_or, &d_info, &d_src_base, &d_G,
                        &send_keys, &send_cnt, &recv_keys, &recv_cnt, &d_nr, &d_nr_all, &d_rdir, &send_words, &recv_words, &d_part_saved})
            b->release();
        if (rep) { (void)hipHostFree(rep); rep = nullptr; }
        if (h_base) { (void)hipHostFree(h_base); h_base = nullptr; }
    }
};

struct gasm_comm {
    gasm_ctx* ctx = nullptr;
    int world = 1, rank = -1;          // rank < 0: virtual (all `world` ranks live in this process, on ctx)
    ncclComm_t nccl = nullptr;
    std::atomic<int> stage{0};         // what the last gasm_pool_exchange_build is doing (watchdogs read it from another thread)
    // plan of the current (n_segments, bbits): who owns what
    u32 S = 0;
    int bbits = -1;
    std::vector<u16> h_own1;
    std::vector<u32> h_seg_first, h_dst_first, h_order;
    std::vector<std::vector<u32>> h_mine;
    DBuf d_own1, d_order, d_dst_first, d_seg_first, d_iota, d_tmp;
    std::vector<RankX> rx;             // one per local rank
    u64 ticket = 1;
    u32 n_local() const { return rank < 0 ? (u32)world : 1u; }
    u32 global_rank(u32 li) const { return rank < 0 ? li : (u32)rank; }
};

static int upload(gasm_ctx* ctx, DBuf& b, const void* src, size_t bytes) {
    GCHK(b.ensure(bytes ? bytes : 8));
    if (bytes) HIPCHK(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return GASM_OK;
}

// ---- transport: the three collectives, for both kinds of communicator.  Arrays are indexed by local rank.
static int x_allgather(gasm_comm* c, const void* const* send, void* const* recv, size_t bytes) {
    gasm_ctx* ctx = c->ctx;
    if (c->rank >= 0) {
        NCHK(rccl_api()->AllGather(send[0], recv[0], bytes, ncclUint8, c->nccl, ctx->stream));
        return GASM_OK;
    }
    for (int d = 0; d < c->world; ++d)
        for (int s = 0; s < c->world; ++s)
            HIPCHK(hipMemcpyAsync(static_cast<char*>(recv[d]) + (size_t)s * bytes, send[s], bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return GASM_OK;
}

static int x_allreduce_u32(gasm_comm* c, u32* const* buf, u64 n) {
    gasm_ctx* ctx = c->ctx;
    if (c->rank >= 0) {
        NCHK(rccl_api()->AllReduce(buf[0], buf[0], n, ncclUint32, ncclSum, c->nccl, ctx->stream));
        return GASM_OK;
    }
    GCHK(c->d_tmp.ensure(n * 4 + 8));
    HIPCHK(hipMemsetAsync(c->d_tmp.p, 0, n * 4, ctx->stream));
    const u32 grid = std::max(1u, std::min<u32>(ceil_div_u64(n, GASM_WG), (u32)ctx->n_cu * 8u));
    for (int s = 0; s < c->world; ++s) GLAUNCH(ctx, "k_x_add_u32", k_x_add_u32, dim3(grid), dim3(GASM_WG), 0, c->d_tmp.as<u32>(), buf[s], n);
    for (int d = 0; d < c->world; ++d) HIPCHK(hipMemcpyAsync(buf[d], c->d_tmp.p, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
    return GASM_OK;
}

// One all-to-all of `n_streams` parallel buffers that share their record offsets (keys + counts of the same records):
// rank li sends records [soff[li][d], soff[li][d + 1]) of send[t][li] to rank d, where they become records
// [roff[d][li], roff[d][li + 1]) of recv[t][d]; elem[t] bytes per record.  RCCL: one group of sends and receives; the
// part a rank keeps is a device copy.
struct XStream { const void* const* send; void* const* recv; size_t elem; };
static int x_alltoallv(gasm_comm* c, const XStream* st, int n_streams, const std::vector<std::vector<u64>>& soff, const std::vector<std::vector<u64>>& roff,
                       u64* bytes_total, u64* bytes_remote) {
    gasm_ctx* ctx = c->ctx;
    const int W = c->world;
    if (c->rank >= 0) {
        const int me = c->rank;
        const RcclApi* api = rccl_api();
        for (int t = 0; t < n_streams; ++t) {
            const u64 n = soff[0][me + 1] - soff[0][me];
            if (n != roff[0][me + 1] - roff[0][me]) { gasm_set_error("all-to-all plan: a rank's own part differs between its send and receive side"); return GASM_ERR_STATE; }
            if (n) HIPCHK(hipMemcpyAsync(static_cast<char*>(st[t].recv[0]) + roff[0][me] * st[t].elem, static_cast<const char*>(st[t].send[0]) + soff[0][me] * st[t].elem,
                                         n * st[t].elem, hipMemcpyDeviceToDevice, ctx->stream));
            if (bytes_total) *bytes_total += n * st[t].elem;
        }
        NCHK(api->GroupStart());
        for (int p = 0; p < W; ++p) {
            if (p == me) continue;
            for (int t = 0; t < n_streams; ++t) {
                const u64 ns = soff[0][p + 1] - soff[0][p], nr = roff[0][p + 1] - roff[0][p];
                if (ns) NCHK(api->Send(static_cast<const char*>(st[t].send[0]) + soff[0][p] * st[t].elem, ns * st[t].elem, ncclUint8, p, c->nccl, ctx->stream));
                if (nr) NCHK(api->Recv(static_cast<char*>(st[t].recv[0]) + roff[0][p] * st[t].elem, nr * st[t].elem, ncclUint8, p, c->nccl, ctx->stream));
                if (bytes_total) *bytes_total += ns * st[t].elem;
                if (bytes_remote) *bytes_remote += ns * st[t].elem;
            }
        }
        NCHK(api->GroupEnd());
        return GASM_OK;
    }
    for (int s = 0; s < W; ++s)
        for (int d = 0; d < W; ++d) {
            const u64 n = soff[s][d + 1] - soff[s][d];
            if (n != roff[d][s + 1] - roff[d][s]) { gasm_set_error("all-to-all plan: rank %d sends %llu records to rank %d, which expects %llu", s, (unsigned long long)n, d, (unsigned long long)(roff[d][s + 1] - roff[d][s])); return GASM_ERR_STATE; }
            if (!n) continue;
            for (int t = 0; t < n_streams; ++t) {
                HIPCHK(hipMemcpyAsync(static_cast<char*>(st[t].recv[d]) + roff[d][s] * st[t].elem, static_cast<const char*>(st[t].send[s]) + soff[s][d] * st[t].elem,
                                      n * st[t].elem, hipMemcpyDeviceToDevice, ctx->stream));
                if (s == 0) { if (bytes_total) *bytes_total += n * st[t].elem; if (bytes_remote && d != 0) *bytes_remote += n * st[t].elem; }
            }
        }
    return GASM_OK;
}

// GASM_X_SYNC=1 (diagnostic): wait for the stream after every stage and say which stage it was
static int stage_done(gasm_comm* c, int stage) {
    static const bool on = getenv("GASM_X_SYNC") != nullptr;
    if (!on) return GASM_OK;
    const hipError_t e = hipStreamSynchronize(c->ctx->stream);
    fprintf(stderr, "[gasm exchange] stage %d done: %s\n", stage, hipGetErrorString(e));
    if (e != hipSuccess) { gasm_set_error("stage %d failed: %s", stage, hipGetErrorString(e)); return GASM_ERR_HIP; }
    return GASM_OK;
}

// ---- the ownership plan of (S, bbits) on the device
static int comm_plan(gasm_comm* c, u32 S, int bbits) {
    if (c->S == S && c->bbits == bbits) return GASM_OK;
    gasm_ctx* ctx = c->ctx;
    const u32 W = (u32)c->world, nb = 1u << bbits;
    const u64 nbt64 = (u64)S << bbits;
    if (nbt64 >= 0x7FFFFFF0ull) { gasm_set_error("too many buckets (%u segments x %u)", S, nb); return GASM_ERR_CAPACITY; }
    const u32 nbt = (u32)nbt64;
    c->h_own1.resize(nbt);
    c->h_mine.assign(W, {});
    for (u32 gb = 0; gb < nbt; ++gb) {
        const u32 o = bucket_owner_of(gb >> bbits, gb & (nb - 1), W);
        c->h_own1[gb] = (u16)o;
        c->h_mine[o].push_back(gb);
    }
    segment_bounds(S, W, c->h_seg_first);
    c->h_dst_first.assign((size_t)W + 1, 0);
    c->h_order.clear();
    c->h_order.reserve(nbt);
    for (u32 d = 0; d < W; ++d) {
        c->h_order.insert(c->h_order.end(), c->h_mine[d].begin(), c->h_mine[d].end());
        c->h_dst_first[d + 1] = (u32)c->h_order.size();
    }
    std::vector<u32> iota(nbt + 1);
    for (u32 i = 0; i <= nbt; ++i) iota[i] = i;
    GCHK(upload(ctx, c->d_own1, c->h_own1.data(), (size_t)nbt * 2));
    GCHK(upload(ctx, c->d_order, c->h_order.data(), (size_t)nbt * 4));
    GCHK(upload(ctx, c->d_dst_first, c->h_dst_first.data(), ((size_t)W + 1) * 4));
    GCHK(upload(ctx, c->d_seg_first, c->h_seg_first.data(), ((size_t)W + 1) * 4));
    GCHK(upload(ctx, c->d_iota, iota.data(), iota.size() * 4));
    for (u32 li = 0; li < c->n_local(); ++li) {
        const std::vector<u32>& m = c->h_mine[c->global_rank(li)];
        GCHK(upload(ctx, c->rx[li].d_mine, m.data(), m.size() * 4));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));      // (host vectors above go out of scope / are reused)
    c->S = S; c->bbits = bbits;
    return GASM_OK;
}

// ---- stage 10: the local runs of one rank, queued (gasm_pool_local_runs without its read-back)
static int local_runs_queue(gasm_pool* p, RankX& x, int k, int bbits, bool small_tbl, bool single_pass) {
    gasm_ctx* ctx = p->ctx;
    BuildState& bs = p->bs;
    GCHK(plan_build(ctx, p->rd, k, 0, bs));
    if (x.part_saved) { std::swap(bs.d_bstart, x.d_part_saved); bs.part_valid = true; x.part_saved = false; }     // (checked against the reads / k / bbits by launch_distinct)
    if (bbits < 0 || bbits > bs.bb_cap) { gasm_set_error("bbits = %d out of range (0..%d for k = %d)", bbits, bs.bb_cap, k); return GASM_ERR_INVALID; }
    const u32 S = p->rd.n_segments;
    bs.bbits = bbits;
    bs.have_actual = false; bs.multi_pass = false; bs.rank_global = false;
    bs.small_tbl = bs.words == 2 ? true : small_tbl;
    bs.single_pass = single_pass;
    p->graphed = false; p->paths_ready = false; p->ss.launched = false; p->ss.valid = false; p->scored = false;
    p->n_runs = S << bbits;
    GCHK(bs.d_bucket_d.ensure(((size_t)p->n_runs + 2) * 4));       // (+ the flag word that travels with the lengths)
    if (bs.n_kmers == 0) {
        GCHK(bs.d_bstart.ensure(((size_t)p->n_runs + 1) * 8));
        GCHK(bs.d_keys.ensure(64)); GCHK(bs.d_mult.ensure(64));
        HIPCHK(hipMemsetAsync(bs.d_bucket_d.p, 0, ((size_t)p->n_runs + 2) * 4, ctx->stream));
        bs.part_valid = false;
        HIPCHK(hipMemsetAsync(bs.d_bstart.p, 0, ((size_t)p->n_runs + 1) * 8, ctx->stream));
        HIPCHK(hipMemsetAsync(bs.d_flags.p, 0, 256, ctx->stream));
        return GASM_OK;
    }
    distinct_caps(bs, S);
    GCHK(launch_distinct(ctx, p->rd, bs));
    hipLaunchKernelGGL(k_x_flag_word, dim3(1), dim3(64), 0, ctx->stream, bs.d_flags.as<u32>(), bs.d_bucket_d.as<u32>() + p->n_runs);
    HIPCHK(hipGetLastError());
    return GASM_OK;
}

template <class K, int TBL>
static int launch_merge(gasm_ctx* ctx, BuildState& bs, RankX& x, u32 n_out, u32 W) {
    GLAUNCH(ctx, "k_bucket_merge", (k_bucket_merge<K, TBL>), dim3(n_out), dim3(GASM_WG), 0, x.recv_keys.as<K>(), x.recv_cnt.as<u32>(), x.d_run_off.as<u64>(),
            x.d_run_len.as<u32>(), W, bs.d_keys.as<K>(), bs.d_mult.as<u32>(), bs.d_bstart.as<u64>(), bs.d_bucket_d.as<u32>(), bs.d_flags.as<u32>(),
            bs.d_fdir.as<u16>(), 2 * bs.k - bs.bbits, x.d_src_base.as<u64>());
    return GASM_OK;
}

// the merge of what an exchange delivered: `n_out` output runs with capacity layout x.d_bstart_new (cap_total records in all)
static int merge_received(gasm_pool* p, RankX& x, u32 n_out, u32 W, u64 cap_total, const u64* recv_tot) {
    gasm_ctx* ctx = p->ctx;
    BuildState& bs = p->bs;
    const size_t KB = 8 * (size_t)bs.words;
    u64 run = 0;
    for (u32 s = 0; s < W; ++s) { x.h_base[s] = run; run += recv_tot[s]; }       // first record of every source in the receive buffers
    GCHK(x.d_src_base.ensure((size_t)W * 8));
    HIPCHK(hipMemcpyAsync(x.d_src_base.p, x.h_base, (size_t)W * 8, hipMemcpyHostToDevice, ctx->stream));
    GCHK(bs.d_keys.ensure(std::max<u64>(cap_total, 1) * KB));
    GCHK(bs.d_mult.ensure(std::max<u64>(cap_total, 1) * 4));
    if (bs.part_valid) {
        // the region layout of the one-pass partition is a function of the reads alone: set aside, back in place for the next
        // step's local runs (uploading it again would be a host sync per step)
        std::swap(bs.d_bstart, x.d_part_saved);
        x.part_saved = true;
        bs.part_valid = false;
    }
    std::swap(bs.d_bstart, x.d_bstart_new);         // the plan kernel wrote the merged runs' layout; the packed runs' layout is no longer needed
    GCHK(bs.d_bucket_d.ensure(((size_t)n_out + 2) * 4));
    bs.small_tbl = bs.words == 2;                   // the merge uses 4096-slot tables for 64-bit keys, 2048-slot ones for 128-bit keys
    bs.fbits = bs.words == 1 ? 10 : 9;
    GCHK(bs.d_fdir.ensure(((size_t)n_out + 1) * ((1u << bs.fbits) + 1) * 2));
    HIPCHK(hipMemsetAsync(bs.d_flags.p, 0, 256, ctx->stream));
    if (n_out) {
        if (bs.words == 1) GCHK((launch_merge<u64, 4096>(ctx, bs, x, n_out, W)));
        else GCHK((launch_merge<K128, 2048>(ctx, bs, x, n_out, W)));
    }
    p->n_runs = n_out;
    p->graphed = false;
    return GASM_OK;
}

template <class K>
static int launch_pack(gasm_ctx* ctx, BuildState& bs, RankX& x, u32 n, const u32* d_list) {
    GLAUNCH(ctx, "k_pack_runs", k_pack_runs<K>, dim3(n), dim3(GASM_WG), 0, bs.d_keys.as<K>(), bs.d_mult.as<u32>(), bs.d_bstart.as<u64>(), bs.d_bucket_d.as<u32>(),
            d_list, x.d_send_off.as<u64>(), x.send_keys.as<K>(), x.send_cnt.as<u32>());
    return GASM_OK;
}

// ---- exchange 3 set-up: who holds how many reads of which segment (once per set of pools)
static int reads_setup(gasm_comm* c, gasm_pool* const* pools) {
    gasm_ctx* ctx = c->ctx;
    const u32 W = (u32)c->world, nl = c->n_local(), S = c->S;
    bool ready = true;
    for (u32 li = 0; li < nl; ++li) ready = ready && c->rx[li].reads_ready && c->rx[li].reads_id == pools[li]->rd.upload_id;
    // (all ranks decide alike: the reads of a pool change only when the pool is created anew, on every rank)
    if (ready) return GASM_OK;
    std::vector<const void*> snd(nl);
    std::vector<void*> rcv(nl);
    std::vector<std::vector<u64>> nr(nl, std::vector<u64>(S));
    for (u32 li = 0; li < nl; ++li) {
        const DevReads& rd = pools[li]->rd;
        for (u32 s = 0; s < S; ++s) nr[li][s] = rd.h_seg_read_off[s + 1] - rd.h_seg_read_off[s];
        GCHK(upload(ctx, c->rx[li].d_nr, nr[li].data(), (size_t)S * 8));
        GCHK(c->rx[li].d_nr_all.ensure((size_t)W * S * 8 + 8));
        snd[li] = c->rx[li].d_nr.p; rcv[li] = c->rx[li].d_nr_all.p;
    }
    GCHK(x_allgather(c, snd.data(), rcv.data(), (size_t)S * 8));
    std::vector<std::vector<u64>> all(nl, std::vector<u64>((size_t)W * S));
    for (u32 li = 0; li < nl; ++li) HIPCHK(hipMemcpyAsync(all[li].data(), c->rx[li].d_nr_all.p, (size_t)W * S * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (u32 li = 0; li < nl; ++li) {
        gasm_pool* p = pools[li];
        RankX& x = c->rx[li];
        const u32 r = c->global_rank(li), flen = p->rd.fixed_len;
        auto nw = [&](u64 n) { return (n * (u64)flen + 31) / 32; };
        // what leaves: one word-aligned piece per segment, in segment order (= by destination)
        std::vector<u64> dir((size_t)S * 3);
        x.send_woff.assign((size_t)W + 1, 0);
        u64 woff = 0;
        x.max_piece_words = 0;
        for (u32 d = 0; d < W; ++d) {
            for (u32 s = c->h_seg_first[d]; s < c->h_seg_first[d + 1]; ++s) {
                const u64 b0 = p->rd.h_seg_read_off[s] * (u64)flen, b1 = p->rd.h_seg_read_off[s + 1] * (u64)flen;
                dir[3 * (size_t)s] = b0; dir[3 * (size_t)s + 1] = b1; dir[3 * (size_t)s + 2] = woff;
                woff += nw(nr[li][s]);
                x.max_piece_words = std::max<u32>(x.max_piece_words, (u32)std::min<u64>(nw(nr[li][s]), 0xFFFFFFFFu));
            }
            x.send_woff[d + 1] = woff;
        }
        GCHK(upload(ctx, x.d_rdir, dir.data(), dir.size() * 8));
        GCHK(x.send_words.ensure(std::max<u64>(woff, 1) * 8));
        // what arrives: from every source the pieces of this rank's segments, back to back
        const u32 a = c->h_seg_first[r], b = c->h_seg_first[r + 1];
        x.recv_woff.assign((size_t)W + 1, 0);
        for (u32 s = 0; s < W; ++s) {
            u64 w = 0;
            for (u32 i = a; i < b; ++i) w += nw(all[li][(size_t)s * S + i]);
            x.recv_woff[s + 1] = x.recv_woff[s] + w;
        }
        HIPCHK(hipStreamSynchronize(ctx->stream));      // (`dir` goes out of scope)
        x.reads_id = p->rd.upload_id;
    }
    // the first exchange goes through gasm_pool_set_reads (piece directory -> the reads' positions); later ones land in
    // the same place directly
    std::vector<const void*> s3(nl);
    std::vector<void*> r3(nl);
    std::vector<std::vector<u64>> so(nl), ro(nl);
    for (u32 li = 0; li < nl; ++li) {
        RankX& x = c->rx[li];
        gasm_pool* p = pools[li];
        if (S > 65535) { gasm_set_error("at most 65535 segments per pooled build"); return GASM_ERR_CAPACITY; }
        if (x.send_woff[W]) GLAUNCH(ctx, "k_repack_reads", k_repack_reads, dim3(std::max(1u, std::min<u32>(ceil_div_u64(std::max<u32>(x.max_piece_words, 1), GASM_WG), 64u)), S),
                                    dim3(GASM_WG), 0, p->rd.d_words.as<u64>(), x.d_rdir.as<u64>(), x.send_words.as<u64>());
        GCHK(x.recv_words.ensure(std::max<u64>(x.recv_woff[W], 1) * 8));
        s3[li] = x.send_words.p; r3[li] = x.recv_words.p; so[li] = x.send_woff; ro[li] = x.recv_woff;
    }
    const XStream st{s3.data(), r3.data(), 8};
    GCHK(x_alltoallv(c, &st, 1, so, ro, nullptr, nullptr));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (u32 li = 0; li < nl; ++li) {
        RankX& x = c->rx[li];
        gasm_pool* p = pools[li];
        const u32 r = c->global_rank(li), a = c->h_seg_first[r], b = c->h_seg_first[r + 1], flen = p->rd.fixed_len;
        auto nw = [&](u64 n) { return (n * (u64)flen + 31) / 32; };
        std::vector<u32> pseg;
        std::vector<u64> preads, pword, run(W);
        for (u32 s = 0; s < W; ++s) run[s] = x.recv_woff[s];
        for (u32 i = a; i < b; ++i)
            for (u32 s = 0; s < W; ++s) {
                const u64 n = all[li][(size_t)s * S + i];
                pseg.push_back(i - a); preads.push_back(n); pword.push_back(run[s]);
                run[s] += nw(n);
            }
        GCHK(gasm_pool_set_reads(p, x.recv_words.p, x.recv_woff[W], (u32)pseg.size(), pseg.data(), preads.data(), pword.data()));
        x.reads_ready = true;
    }
    return GASM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
extern "C" {

int gasm_comm_unique_id(void* id) {
    if (!id) { gasm_set_error("gasm_comm_unique_id: null argument"); return GASM_ERR_INVALID; }
    static_assert(sizeof(ncclUniqueId) == GASM_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    const RcclApi* api = rccl_api();
    if (!api) return GASM_ERR_NO_DEVICE;
    ncclUniqueId u;
    NCHK(api->GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return GASM_OK;
}

static gasm_comm* comm_new(gasm_ctx* ctx, int world, int rank) {
    gasm_comm* c = new gasm_comm();
    c->ctx = ctx; c->world = world; c->rank = rank;
    c->rx.resize(c->n_local());
    for (RankX& x : c->rx) {
        if (hipHostMalloc((void**)&x.rep, (size_t)(2 * (2 * world + 6)) * 8, hipHostMallocCoherent) != hipSuccess ||
            hipHostMalloc((void**)&x.h_base, (size_t)(world + 1) * 8, hipHostMallocCoherent) != hipSuccess) {
            gasm_set_error("hipHostMalloc failed");
            for (RankX& y : c->rx) y.release();
            delete c;
            return nullptr;
        }
        memset(x.rep, 0, (size_t)(2 * (2 * world + 6)) * 8);
    }
    return c;
}

int gasm_comm_create(gasm_ctx* ctx, const void* id, int rank, int world, gasm_comm** out) {
    if (!ctx || !id || !out) { gasm_set_error("gasm_comm_create: null argument"); return GASM_ERR_INVALID; }
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world || world > 4096) { gasm_set_error("gasm_comm_create: rank %d of %d", rank, world); return GASM_ERR_INVALID; }
    const RcclApi* api = rccl_api();
    if (!api) return GASM_ERR_NO_DEVICE;
    HIPCHK(hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t nc = nullptr;
    NCHK(api->CommInitRank(&nc, world, u, rank));
    gasm_comm* c = comm_new(ctx, world, rank);
    if (!c) { (void)api->CommDestroy(nc); return GASM_ERR_HIP; }
    c->nccl = nc;
    *out = c;
    return GASM_OK;
}

int gasm_comm_create_virtual(gasm_ctx* ctx, int world, gasm_comm** out) {
    if (!ctx || !out) { gasm_set_error("gasm_comm_create_virtual: null argument"); return GASM_ERR_INVALID; }
    *out = nullptr;
    if (world < 1 || world > 4096) { gasm_set_error("gasm_comm_create_virtual: world = %d", world); return GASM_ERR_INVALID; }
    HIPCHK(hipSetDevice(ctx->device));
    gasm_comm* c = comm_new(ctx, world, -1);
    if (!c) return GASM_ERR_HIP;
    *out = c;
    return GASM_OK;
}

void gasm_comm_destroy(gasm_comm* c) {
    if (!c) return;
    if (c->ctx) { (void)hipSetDevice(c->ctx->device); (void)hipStreamSynchronize(c->ctx->stream); }
    if (c->nccl && rccl_api()) (void)rccl_api()->CommDestroy(c->nccl);
    for (RankX& x : c->rx) x.release();
    for (DBuf* b : {&c->d_own1, &c->d_order, &c->d_dst_first, &c->d_seg_first, &c->d_iota, &c->d_tmp}) b->release();
    delete c;
}

int gasm_comm_world(const gasm_comm* c) { return c ? c->world : 0; }
int gasm_comm_rank(const gasm_comm* c) { return c ? c->rank : -1; }
int gasm_comm_stage(const gasm_comm* c) { return c ? c->stage.load() : 0; }

int gasm_pool_bucket_owner(uint32_t n_segments, int bbits, uint32_t world, uint32_t* owner) {
    if (!owner || world == 0 || bbits < 0 || bbits > 10) { gasm_set_error("gasm_pool_bucket_owner: bad argument"); return GASM_ERR_INVALID; }
    const u64 nbt = (u64)n_segments << bbits;
    for (u64 gb = 0; gb < nbt; ++gb) owner[gb] = bucket_owner_of(gb >> bbits, gb & ((1u << bbits) - 1), world);
    return GASM_OK;
}

int gasm_pool_segment_bounds(uint32_t n_segments, uint32_t world, uint32_t* first) {
    if (!first || world == 0) { gasm_set_error("gasm_pool_segment_bounds: bad argument"); return GASM_ERR_INVALID; }
    std::vector<u32> f;
    segment_bounds(n_segments, world, f);
    memcpy(first, f.data(), f.size() * 4);
    return GASM_OK;
}

int gasm_pool_exchange_build(gasm_comm* c, gasm_pool* const* pools, uint32_t n_pools, int k, int bbits, int kmer, const double* table, uint64_t* stats) {
    try {
    if (!c || !pools) { gasm_set_error("gasm_pool_exchange_build: null argument"); return GASM_ERR_INVALID; }
    const u32 nl = c->n_local(), W = (u32)c->world;
    if (n_pools != nl) { gasm_set_error("gasm_pool_exchange_build: %u pools for %u local rank(s)", n_pools, nl); return GASM_ERR_INVALID; }
    gasm_ctx* ctx = c->ctx;
    for (u32 li = 0; li < nl; ++li) {
        if (!pools[li] || pools[li]->ctx != ctx) { gasm_set_error("gasm_pool_exchange_build: pool %u does not belong to the communicator's context", li); return GASM_ERR_INVALID; }
        if (pools[li]->rd.n_segments != pools[0]->rd.n_segments || pools[li]->rd.fixed_len != pools[0]->rd.fixed_len) { gasm_set_error("the pools disagree on segments or read length"); return GASM_ERR_INVALID; }
    }
    HIPCHK(hipSetDevice(ctx->device));
    GasmRange range("gasm:pool exchange + build");
    const u32 S = pools[0]->rd.n_segments;
    u64 st_bytes[3] = {0, 0, 0}, st_remote[3] = {0, 0, 0};
    int attempts = 0;
    bool small_tbl = true, single_pass = true;
    { const char* v = getenv("GASM_SINGLE_PASS"); if (v && *v && atoi(v) == 0) single_pass = false; }
    const int bb_cap = std::min(10, 2 * (k - 1));
    const int words = k <= 31 ? 1 : 2;
    const size_t KB = 8 * (size_t)words;
    const u32 limit = words == 1 ? GASM_TBL_LIMIT : GASM_TBL_LIMIT / 2;
    std::vector<const void*> cs(nl);
    std::vector<void*> cr(nl);
    std::vector<u32*> cu(nl);
    for (;;) {
        ++attempts;
        if (attempts > 16) { gasm_set_error("the pooled build did not settle on a configuration"); return GASM_ERR_CAPACITY; }
        GCHK(comm_plan(c, S, bbits));
        const u32 nbt = S << bbits, nb = 1u << bbits;
        const u64 stride = (u64)nbt + 2;                                 // run lengths + flag word + padding
        // ---- stage 10
        c->stage = 10;
        for (u32 li = 0; li < nl; ++li) GCHK(local_runs_queue(pools[li], c->rx[li], k, bbits, small_tbl, single_pass));
        // ---- stage 11: lengths + flags of everybody, plan 1, report 1
        GCHK(stage_done(c, 10));
        c->stage = 11;
        for (u32 li = 0; li < nl; ++li) {
            RankX& x = c->rx[li];
            GCHK(x.d_lens_all.ensure((size_t)W * stride * 4));
            cs[li] = pools[li]->bs.d_bucket_d.p; cr[li] = x.d_lens_all.p;
        }
        GCHK(x_allgather(c, cs.data(), cr.data(), (size_t)stride * 4));
        const u64 ticket = ++c->ticket;
        for (u32 li = 0; li < nl; ++li) {
            RankX& x = c->rx[li];
            const u32 r = c->global_rank(li), n_mine = (u32)c->h_mine[r].size();
            GCHK(x.d_send_off.ensure(((size_t)nbt + 2) * 8));
            GCHK(x.d_send_tot.ensure((size_t)W * 8 + 8)); GCHK(x.d_recv_tot.ensure((size_t)W * 8 + 8));
            GCHK(x.d_run_off.ensure(std::max<size_t>((size_t)std::max(n_mine, S * nb) * W, 1) * 8));
            GCHK(x.d_run_len.ensure(std::max<size_t>((size_t)std::max(n_mine, S * nb) * W, 1) * 4));
            GCHK(x.d_bstart_new.ensure(((size_t)std::max(n_mine, nbt) + 2) * 8));
            GCHK(x.d_flags_or.ensure(8)); GCHK(x.d_info.ensure(64));
            GLAUNCH(ctx, "k_x1_plan", k_x1_plan, dim3(W + 2), dim3(1024), 0, x.d_lens_all.as<u32>(), stride, nbt, c->d_order.as<u32>(), c->d_dst_first.as<u32>(),
                    x.d_mine.as<u32>(), n_mine, W, r, limit, x.d_send_off.as<u64>(), x.d_send_tot.as<u64>(), x.d_run_off.as<u64>(), x.d_run_len.as<u32>(),
                    x.d_recv_tot.as<u64>(), x.d_bstart_new.as<u64>(), x.d_flags_or.as<u32>());
            GLAUNCH(ctx, "k_x_report", k_x_report, dim3(1), dim3(64), 0, x.d_send_tot.as<u64>(), x.d_recv_tot.as<u64>(), W, x.d_bstart_new.as<u64>() + n_mine, 1u,
                    x.d_flags_or.as<u32>(), x.rep, ticket);
        }
        u32 flags = 0;
        for (u32 li = 0; li < nl; ++li) {
            GCHK(gasm_wait_word64(ctx, c->rx[li].rep + 2 * W + 5, ticket));
            flags |= (u32)c->rx[li].rep[2 * W + 4];
        }
        if (flags & 3u) {
            // every rank reads the same OR of everybody's flags: the same step of the ladder everywhere
            if ((flags & 2u) && single_pass) single_pass = false;                  // a region of the one-pass partition overflowed: exact layout
            else if (small_tbl && words == 1) small_tbl = false;                    // larger tables
            else if (bbits < bb_cap) bbits = std::min(bb_cap, bbits + 2);           // more buckets (changes the ownership: all ranks alike)
            else { gasm_set_error("a k-mer bucket of some rank holds more than %u distinct k-mers even with %d bucket bits", limit, bbits); return GASM_ERR_CAPACITY; }
            continue;
        }
        // ---- stage 12: all-to-all #1
        GCHK(stage_done(c, 11));
        c->stage = 12;
        std::vector<std::vector<u64>> soff(nl, std::vector<u64>(W + 1, 0)), roff(nl, std::vector<u64>(W + 1, 0));
        std::vector<const void*> sk(nl), sc(nl);
        std::vector<void*> rk(nl), rc(nl);
        for (u32 li = 0; li < nl; ++li) {
            RankX& x = c->rx[li];
            BuildState& bs = pools[li]->bs;
            for (u32 d = 0; d < W; ++d) { soff[li][d + 1] = soff[li][d] + x.rep[d]; roff[li][d + 1] = roff[li][d] + x.rep[W + d]; }
            GCHK(x.send_keys.ensure(std::max<u64>(soff[li][W], 1) * KB)); GCHK(x.send_cnt.ensure(std::max<u64>(soff[li][W], 1) * 4));
            GCHK(x.recv_keys.ensure(std::max<u64>(roff[li][W], 1) * KB)); GCHK(x.recv_cnt.ensure(std::max<u64>(roff[li][W], 1) * 4));
            if (soff[li][W]) {
                if (words == 1) GCHK(launch_pack<u64>(ctx, bs, x, nbt, c->d_order.as<u32>()));
                else GCHK(launch_pack<K128>(ctx, bs, x, nbt, c->d_order.as<u32>()));
            }
            sk[li] = x.send_keys.p; sc[li] = x.send_cnt.p; rk[li] = x.recv_keys.p; rc[li] = x.recv_cnt.p;
        }
        {
            const XStream st[2] = {{sk.data(), rk.data(), KB}, {sc.data(), rc.data(), 4}};
            GCHK(x_alltoallv(c, st, 2, soff, roff, &st_bytes[0], &st_remote[0]));
        }
        // ---- stage 13: merge
        GCHK(stage_done(c, 12));
        c->stage = 13;
        for (u32 li = 0; li < nl; ++li) {
            RankX& x = c->rx[li];
            const u32 r = c->global_rank(li), n_mine = (u32)c->h_mine[r].size();
            GCHK(merge_received(pools[li], x, n_mine, W, x.rep[2 * W], x.rep + W));
        }
        // ---- stage 21: merged lengths + flags of everybody, plan 2, report 2
        GCHK(stage_done(c, 13));
        c->stage = 21;
        for (u32 li = 0; li < nl; ++li) {
            RankX& x = c->rx[li];
            BuildState& bs = pools[li]->bs;
            const u32 r = c->global_rank(li), n_mine = (u32)c->h_mine[r].size();
            GCHK(x.d_G.ensure((size_t)stride * 4));
            HIPCHK(hipMemsetAsync(x.d_G.p, 0, (size_t)stride * 4, ctx->stream));
            GLAUNCH(ctx, "k_x2_fill", k_x2_fill, dim3(std::max(1u, ceil_div_u64(n_mine, GASM_WG))), dim3(GASM_WG), 0, x.d_G.as<u32>(), nbt, x.d_mine.as<u32>(), n_mine,
                    bs.d_bucket_d.as<u32>(), bs.d_flags.as<u32>());
            cu[li] = x.d_G.as<u32>();
        }
        GCHK(x_allreduce_u32(c, cu.data(), stride));
        const u64 ticket2 = ++c->ticket;
        for (u32 li = 0; li < nl; ++li) {
            RankX& x = c->rx[li];
            const u32 r = c->global_rank(li), n_mine = (u32)c->h_mine[r].size();
            const u32 a = c->h_seg_first[r], b = c->h_seg_first[r + 1];
            GCHK(x.d_bstart_new.ensure(((size_t)std::max(n_mine, nbt) + 2) * 8));     // (the merge took the last one for the pool's runs)
            HIPCHK(hipMemcpyAsync(x.d_flags_or.p, x.d_G.as<u32>() + nbt, 4, hipMemcpyDeviceToDevice, ctx->stream));
            GLAUNCH(ctx, "k_x2_plan", k_x2_plan, dim3(W + 2), dim3(1024), 0, x.d_G.as<u32>(), c->d_own1.as<u16>(), W, r, a * nb, (b - a) * nb, bbits, x.d_mine.as<u32>(), n_mine,
                    c->d_seg_first.as<u32>(), x.d_send_off.as<u64>(), x.d_send_tot.as<u64>(), x.d_run_off.as<u64>(), x.d_run_len.as<u32>(), x.d_recv_tot.as<u64>(),
                    x.d_bstart_new.as<u64>(), x.d_info.as<u64>());
            GLAUNCH(ctx, "k_x_report", k_x_report, dim3(1), dim3(64), 0, x.d_send_tot.as<u64>(), x.d_recv_tot.as<u64>(), W, x.d_info.as<u64>(), 2u, x.d_flags_or.as<u32>(),
                    x.rep + (2 * W + 6), ticket2);
        }
        flags = 0;
        for (u32 li = 0; li < nl; ++li) {
            const u64* rep2 = c->rx[li].rep + (2 * W + 6);
            GCHK(gasm_wait_word64(ctx, rep2 + 2 * W + 5, ticket2));
            flags |= rep2[2 * W + 4] ? 1u : 0u;
        }
        if (flags) {
            if (bbits < bb_cap) { bbits = std::min(bb_cap, bbits + 2); continue; }
            gasm_set_error("a merged k-mer bucket holds more than %u distinct k-mers even with %d bucket bits", limit, bbits);
            return GASM_ERR_CAPACITY;
        }
        // ---- stage 22: all-to-all #2
        GCHK(stage_done(c, 21));
        c->stage = 22;
        for (u32 li = 0; li < nl; ++li) {
            RankX& x = c->rx[li];
            BuildState& bs = pools[li]->bs;
            const u64* rep2 = x.rep + (2 * W + 6);
            const u32 r = c->global_rank(li), n_mine = (u32)c->h_mine[r].size();
            for (u32 d = 0; d < W; ++d) { soff[li][d + 1] = soff[li][d] + rep2[d]; roff[li][d + 1] = roff[li][d] + rep2[W + d]; }
            GCHK(x.send_keys.ensure(std::max<u64>(soff[li][W], 1) * KB)); GCHK(x.send_cnt.ensure(std::max<u64>(soff[li][W], 1) * 4));
            GCHK(x.recv_keys.ensure(std::max<u64>(roff[li][W], 1) * KB)); GCHK(x.recv_cnt.ensure(std::max<u64>(roff[li][W], 1) * 4));
            if (soff[li][W] && n_mine) {
                if (words == 1) GCHK(launch_pack<u64>(ctx, bs, x, n_mine, c->d_iota.as<u32>()));
                else GCHK(launch_pack<K128>(ctx, bs, x, n_mine, c->d_iota.as<u32>()));
            }
            sk[li] = x.send_keys.p; sc[li] = x.send_cnt.p; rk[li] = x.recv_keys.p; rc[li] = x.recv_cnt.p;
        }
        {
            const XStream st[2] = {{sk.data(), rk.data(), KB}, {sc.data(), rc.data(), 4}};
            GCHK(x_alltoallv(c, st, 2, soff, roff, &st_bytes[1], &st_remote[1]));
        }
        // ---- stage 23: placement, graph, contigs of the rank's own segments
        GCHK(stage_done(c, 22));
        c->stage = 23;
        for (u32 li = 0; li < nl; ++li) {
            RankX& x = c->rx[li];
            const u64* rep2 = x.rep + (2 * W + 6);
            const u32 r = c->global_rank(li), a = c->h_seg_first[r], b = c->h_seg_first[r + 1];
            GCHK(merge_received(pools[li], x, (b - a) * nb, W, rep2[2 * W], rep2 + W));
            GCHK(pool_graph_launch(pools[li], b - a, rep2[2 * W], rep2[2 * W + 1]));
        }
        break;
    }
    // ---- stage 31: the reads of a segment to the segment's owner
    if (table) {
        GCHK(stage_done(c, 23));
        c->stage = 31;
        bool first = false;
        for (u32 li = 0; li < nl; ++li) first = first || !(c->rx[li].reads_ready && c->rx[li].reads_id == pools[li]->rd.upload_id);
        if (first) GCHK(reads_setup(c, pools));
        else {
            std::vector<const void*> s3(nl);
            std::vector<void*> r3(nl);
            std::vector<std::vector<u64>> so(nl), ro(nl);
            for (u32 li = 0; li < nl; ++li) {
                RankX& x = c->rx[li];
                gasm_pool* p = pools[li];
                if (x.send_woff[W]) GLAUNCH(ctx, "k_repack_reads", k_repack_reads, dim3(std::max(1u, std::min<u32>(ceil_div_u64(std::max<u32>(x.max_piece_words, 1), GASM_WG), 64u)), S),
                                            dim3(GASM_WG), 0, p->rd.d_words.as<u64>(), x.d_rdir.as<u64>(), x.send_words.as<u64>());
                s3[li] = x.send_words.p; r3[li] = p->own.d_words.p; so[li] = x.send_woff; ro[li] = x.recv_woff;
            }
            const XStream st{s3.data(), r3.data(), 8};
            GCHK(x_alltoallv(c, &st, 1, so, ro, &st_bytes[2], &st_remote[2]));
        }
        if (first) for (u32 d = 0; d < W; ++d) {        // (the set-up's exchange was the step's: account for it)
            const u64 n = c->rx[0].send_woff[d + 1] - c->rx[0].send_woff[d];
            st_bytes[2] += n * 8;
            if (d != c->global_rank(0)) st_remote[2] += n * 8;
        }
        // ---- stage 32: scoring, queued behind the graph
        GCHK(stage_done(c, 31));
        c->stage = 32;
        for (u32 li = 0; li < nl; ++li) GCHK(pool_score_launch(pools[li], kmer, table, false));
    }
    GCHK(stage_done(c, 32));
    c->stage = 0;
    if (stats) {
        for (int i = 0; i < 3; ++i) { stats[i] = st_bytes[i]; stats[3 + i] = st_remote[i]; }
        stats[6] = (u64)attempts; stats[7] = (u64)bbits;
    }
    return GASM_OK;
    } catch (const std::bad_alloc&) {
        gasm_set_error("out of host memory");
        return GASM_ERR_CAPACITY;
    } catch (const std::exception& e) {
        gasm_set_error("internal error: %s", e.what());
        return GASM_ERR_INVALID;
    }
}

}  // extern "C"
