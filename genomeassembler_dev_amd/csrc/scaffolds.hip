// scaffolds.hip — assemble_contigs with its result kept on the device.  The greedy merge itself runs
// on contig indices (host_algos.cpp: same visiting order as lib/DeNovoAssembler.cpp:233-266, threads over permutations,
// ~4 ms for 10 000 permutations); what used to cost the time — building 3.8e8 characters of scaffold text, handing them
// to the caller and uploading them again for calc_breakscore — is replaced by chains expanded to 2-bit on the GPU
// (kernels_asm.hip), sorted and de-duplicated there as strings, and passed on behind a gasm_scaffolds handle.
#include <algorithm>
#include <thread>

#include "pipeline.h"
#include "scaffolds.h"

// GASM_ASM_TIMING=1 (diagnostic): host wall time of the phases of assemble_contigs' device route, to stderr
#include <chrono>
struct AsmLap {
    bool on = getenv("GASM_ASM_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void operator()(gasm_ctx* ctx, const char* what) {
        if (!on) return;
        (void)hipStreamSynchronize(ctx->stream);
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[asm] %-34s %8.2f ms\n", what, std::chrono::duration<double>(n - t).count() * 1e3);
        t = n;
    }
};

static int up64(gasm_ctx* ctx, DBuf& b, const void* src, size_t bytes) {
    GCHK(b.ensure(bytes ? bytes : 8));
    if (bytes) HIPCHK(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return GASM_OK;
}

struct ChainCsr {
    std::vector<u64> sig_off, elem_pos, out_off;
    std::vector<u32> elem_contig, elem_skip;
    void clear() { sig_off.assign(1, 0); out_off.assign(1, 0); elem_pos.clear(); elem_contig.clear(); elem_skip.clear(); }
    // signature = u32 contig, (u32 overlap, u32 contig)*.  order[i] = the signature of scaffold i; threads over the scaffolds
    // (440 000 elements for one 50 kb experiment: 5 ms on one thread)
    void build(const std::vector<std::string>& sigs, const u32* order, size_t m, const std::vector<std::string>& contigs) {
        sig_off.assign(m + 1, 0);
        out_off.assign(m + 1, 0);
        for (size_t i = 0; i < m; ++i) sig_off[i + 1] = sig_off[i] + (sigs[order ? order[i] : i].size() + 4) / 8;
        const size_t ne = sig_off[m];
        elem_pos.resize(ne); elem_contig.resize(ne); elem_skip.resize(ne);
        std::vector<u64> len(m);
        unsigned nt = std::thread::hardware_concurrency();
        nt = std::max(1u, std::min(nt, 16u));
        if (m < 2048) nt = 1;
        auto work = [&](unsigned t) {
            for (size_t i = m * t / nt; i < m * (t + 1) / nt; ++i) {
                const std::string& sg = sigs[order ? order[i] : i];
                size_t e = sig_off[i];
                u32 x;
                memcpy(&x, sg.data(), 4);
                u64 pos = 0;
                elem_contig[e] = x; elem_skip[e] = 0; elem_pos[e] = 0; ++e;
                pos += contigs[x].size();
                for (size_t q = 4; q + 8 <= sg.size(); q += 8, ++e) {
                    u32 ov, y;
                    memcpy(&ov, sg.data() + q, 4);
                    memcpy(&y, sg.data() + q + 4, 4);
                    elem_contig[e] = y; elem_skip[e] = ov; elem_pos[e] = pos;
                    pos += contigs[y].size() - ov;
                }
                len[i] = pos;
            }
        };
        if (nt == 1) work(0);
        else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; ++t) th.emplace_back(work, t);
            for (auto& t : th) t.join();
        }
        for (size_t i = 0; i < m; ++i) out_off[i + 1] = out_off[i] + len[i];
    }
};

static int expand(gasm_ctx* ctx, const ChainCsr& c, const u64* d_cwords, const u64* d_coff, DBuf& d_out, DBuf (&tmp)[5]) {
    const u32 m = (u32)(c.sig_off.size() - 1);
    const u64 T = c.out_off[m], nw = (T + 31) / 32 + 4;
    GCHK(up64(ctx, tmp[0], c.sig_off.data(), c.sig_off.size() * 8));
    GCHK(up64(ctx, tmp[1], c.elem_contig.data(), c.elem_contig.size() * 4));
    GCHK(up64(ctx, tmp[2], c.elem_skip.data(), c.elem_skip.size() * 4));
    GCHK(up64(ctx, tmp[3], c.elem_pos.data(), c.elem_pos.size() * 8));
    GCHK(up64(ctx, tmp[4], c.out_off.data(), c.out_off.size() * 8));
    GCHK(d_out.ensure(nw * 8));
    GLAUNCH(ctx, "k_chain_expand", k_chain_expand, dim3(std::max(1u, std::min<u32>(ceil_div_u64(nw, GASM_WG), (u32)ctx->n_cu * 16u))), dim3(GASM_WG), 0, d_cwords,
            d_coff, tmp[0].as<u64>(), tmp[1].as<u32>(), tmp[2].as<u32>(), tmp[3].as<u64>(), tmp[4].as<u64>(), m, d_out.as<u64>(), nw);
    return GASM_OK;
}

int scaffolds_from_signatures(gasm_ctx* ctx, const std::vector<std::string>& contigs, const std::vector<std::string>& sigs, gasm_scaffolds** out) {
    HIPCHK(hipSetDevice(ctx->device));
    const u32 n = (u32)contigs.size(), m = (u32)sigs.size();
    DBuf ascii, err, cwords, d_coff, d_w1, d_idx, d_same, tmp[5];
    struct Rel { std::vector<DBuf*> v; ~Rel() { for (DBuf* b : v) b->release(); } } rel{{&ascii, &err, &cwords, &d_coff, &d_w1, &d_idx, &d_same, &tmp[0], &tmp[1],
                                                                                         &tmp[2], &tmp[3], &tmp[4]}};
    // ---- contigs, packed
    std::vector<u64> coff((size_t)n + 1, 0);
    std::string cat;
    for (u32 i = 0; i < n; ++i) { cat += contigs[i]; coff[i + 1] = cat.size(); }
    GCHK(err.ensure(8));
    HIPCHK(hipMemsetAsync(err.p, 0, 8, ctx->stream));
    GCHK(up64(ctx, ascii, cat.data(), cat.size()));
    {
        const u64 nw = (cat.size() + 31) / 32;
        GCHK(cwords.ensure((nw + 4) * 8));
        GLAUNCH(ctx, "k_pack_ascii", k_pack_ascii, dim3(std::max(1u, std::min<u32>(ceil_div_u64(nw + 4, GASM_WG), (u32)ctx->n_cu * 32u))), dim3(GASM_WG), 0, ascii.as<u8>(),
                (u64)cat.size(), (const u64*)nullptr, cwords.as<u64>(), err.as<u32>());
    }
    GCHK(up64(ctx, d_coff, coff.data(), coff.size() * 8));
    AsmLap lap;
    lap(ctx, "scaffolds: contigs packed");
    // ---- every distinct chain once, as text-in-2-bit; sort and de-duplicate as strings (lib/DeNovoAssembler.cpp:275-286)
    ChainCsr c1;
    c1.build(sigs, nullptr, sigs.size(), contigs);
    lap(ctx, "scaffolds: chain CSR (host)");
    GCHK(expand(ctx, c1, cwords.as<u64>(), d_coff.as<u64>(), d_w1, tmp));
    lap(ctx, "scaffolds: expand 1");
    u32 p2 = 1;
    while (p2 < m) p2 <<= 1;
    std::vector<u32> idx(p2, GASM_NONE32);
    for (u32 i = 0; i < m; ++i) idx[i] = i;
    GCHK(up64(ctx, d_idx, idx.data(), (size_t)p2 * 4));
    const ChainSigs cs{getenv("GASM_ASM_PLAIN_SORT") ? nullptr : tmp[0].as<u64>(), tmp[1].as<u32>(), tmp[2].as<u32>(), tmp[3].as<u64>()};
    // bitonic network on the index array; the steps whose partners lie inside one 512-element block (j <= 256) of a stage run as
    // ONE launch with the indices in LDS (120 launches -> 36 for 32 768 scaffolds)
    for (u32 kk = 2; kk <= p2; kk <<= 1) {
        u32 j = kk >> 1;
        for (; j > GASM_WG; j >>= 1)
            GLAUNCH(ctx, "k_str_bitonic", k_str_bitonic, dim3(std::max(1u, ceil_div_u64(p2 >> 1, GASM_WG))), dim3(GASM_WG), 0, d_w1.as<u64>(), tmp[4].as<u64>(),
                    d_idx.as<u32>(), p2, kk, j, cs);
        GLAUNCH(ctx, "k_str_bitonic_block", k_str_bitonic_block, dim3(std::max(1u, ceil_div_u64(p2 >> 1, GASM_WG))), dim3(GASM_WG), 0, d_w1.as<u64>(), tmp[4].as<u64>(),
                d_idx.as<u32>(), p2, kk, j, cs);
    }
    GCHK(d_same.ensure(std::max<u32>(m, 1)));
    if (m) GLAUNCH(ctx, "k_str_adjacent_eq", k_str_adjacent_eq, dim3(ceil_div_u64(m, GASM_WG)), dim3(GASM_WG), 0, d_w1.as<u64>(), tmp[4].as<u64>(), d_idx.as<u32>(), m,
                   d_same.as<u8>(), cs);
    std::vector<u8> same(m);
    u32 herr = 0;
    HIPCHK(hipMemcpyAsync(&herr, err.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    if (m) {
        HIPCHK(hipMemcpyAsync(idx.data(), d_idx.p, (size_t)m * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(same.data(), d_same.p, m, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    lap(ctx, "scaffolds: string sort + unique");
    if (herr) { gasm_set_error("contigs contain a base outside upper-case ACGT"); return GASM_ERR_NON_ACGT; }
    // ---- the reference's last step: the sorted distinct strings, then std::sort by length, longest first
    // (lib/DeNovoAssembler.cpp:289-294: not a stable sort; the same libstdc++ call on the same initial order makes the same
    // comparisons and moves, whatever the elements carry)
    struct Ent { u64 len; u32 id; };
    std::vector<Ent> fin;
    for (u32 i = 0; i < m; ++i) if (!same[i]) fin.push_back(Ent{c1.out_off[idx[i] + 1] - c1.out_off[idx[i]], idx[i]});
    std::sort(fin.begin(), fin.end(), [](const Ent& a, const Ent& b) { return a.len > b.len; });
    // ---- the distinct scaffolds in their final order, once more, as the handle's stream
    ChainCsr c2;
    {
        std::vector<u32> order(fin.size());
        for (size_t i = 0; i < fin.size(); ++i) order[i] = fin[i].id;
        c2.build(sigs, order.data(), order.size(), contigs);
    }
    gasm_scaffolds* sc = new gasm_scaffolds();
    sc->ctx = ctx;
    sc->n = (u32)fin.size();
    sc->h_off = c2.out_off;
    lap(ctx, "scaffolds: length sort + CSR 2 (host)");
    int st = expand(ctx, c2, cwords.as<u64>(), d_coff.as<u64>(), sc->d_words, tmp);
    if (st == GASM_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) { gasm_set_error("scaffold expansion failed"); st = GASM_ERR_HIP; }
    lap(ctx, "scaffolds: expand 2");
    if (st != GASM_OK) { sc->d_words.release(); delete sc; return st; }
    *out = sc;
    return GASM_OK;
}

// The greedy merge of every permutation on the GPU (k_asm_match + k_asm_merge), its chains as sorted distinct signatures
// (what gasm_host::assemble_signatures gives).  Permutations that ran into two chains of equal length (the reference's
// full-string test) are redone by the host routine.  *used = false: the index form does not apply (host string form).
int assemble_signatures_device(gasm_ctx* ctx, const std::vector<std::string>& contigs, const u32* perm, u64 rows, u64 row_len, int k,
                               std::vector<std::string>& sigs, bool* used, u64* rows_on_host) {
    *used = false;
    if (rows_on_host) *rows_on_host = 0;
    const u64 n = row_len;
    size_t min_len = ~(size_t)0;
    for (const std::string& s : contigs) min_len = std::min(min_len, s.size());
    if (!(n > 0 && n == contigs.size() && k >= 2 && k <= 255 && min_len >= (size_t)(k - 1) && n <= 2048 && rows > 0)) return GASM_OK;
    HIPCHK(hipSetDevice(ctx->device));
    DBuf ascii, err, cwords, d_coff, d_clen, d_perm, d_match, d_ra, d_la, d_next, d_ov, d_heads, d_nch, d_need;
    struct Rel { std::vector<DBuf*> v; ~Rel() { for (DBuf* b : v) b->release(); } } rel{{&ascii, &err, &cwords, &d_coff, &d_clen, &d_perm, &d_match, &d_ra, &d_la,
                                                                                         &d_next, &d_ov, &d_heads, &d_nch, &d_need}};
    AsmLap lap;
    std::vector<u64> coff(n + 1, 0);
    std::vector<u32> clen(n);
    std::string cat;
    for (u64 i = 0; i < n; ++i) { cat += contigs[i]; coff[i + 1] = cat.size(); clen[i] = (u32)contigs[i].size(); }
    GCHK(err.ensure(8));
    HIPCHK(hipMemsetAsync(err.p, 0, 8, ctx->stream));
    GCHK(up64(ctx, ascii, cat.data(), cat.size()));
    const u64 nw = (cat.size() + 31) / 32;
    GCHK(cwords.ensure((nw + 4) * 8));
    GLAUNCH(ctx, "k_pack_ascii", k_pack_ascii, dim3(std::max(1u, std::min<u32>(ceil_div_u64(nw + 4, GASM_WG), (u32)ctx->n_cu * 32u))), dim3(GASM_WG), 0, ascii.as<u8>(),
            (u64)cat.size(), (const u64*)nullptr, cwords.as<u64>(), err.as<u32>());
    GCHK(up64(ctx, d_coff, coff.data(), coff.size() * 8));
    GCHK(up64(ctx, d_clen, clen.data(), clen.size() * 4));
    GCHK(up64(ctx, d_perm, perm, rows * n * 4));
    const u64 msz = (u64)k * n * n;
    GCHK(d_match.ensure(msz));
    GCHK(d_ra.ensure((u64)k * n));
    GCHK(d_la.ensure((u64)k + 8));
    HIPCHK(hipMemsetAsync(d_ra.p, 0, (u64)k * n, ctx->stream));
    HIPCHK(hipMemsetAsync(d_la.p, 0, (u64)k + 8, ctx->stream));
    GLAUNCH(ctx, "k_asm_match", k_asm_match, dim3(std::max(1u, std::min<u32>(ceil_div_u64((u64)(k - 1) * n * n, GASM_WG), (u32)ctx->n_cu * 64u))), dim3(GASM_WG), 0,
            cwords.as<u64>(), d_coff.as<u64>(), (u32)n, k, d_match.as<u8>(), d_ra.as<u8>(), d_la.as<u8>());
    GCHK(d_next.ensure(rows * n * 4));
    GCHK(d_ov.ensure(rows * n));
    GCHK(d_heads.ensure(rows * n * 4));
    GCHK(d_nch.ensure(rows * 4));
    GCHK(d_need.ensure(rows));
    const size_t lds = (size_t)n * 13 + 16;
    GLAUNCH(ctx, "k_asm_merge", k_asm_merge, dim3((u32)std::min<u64>(rows, (u64)ctx->n_cu * 32u)), dim3(64), lds, d_perm.as<u32>(), (u32)rows, (u32)n, k, d_clen.as<u32>(),
            d_match.as<u8>(), d_ra.as<u8>(), d_la.as<u8>(), cwords.as<u64>(), d_coff.as<u64>(), d_next.as<u32>(), d_ov.as<u8>(), d_heads.as<u32>(), d_nch.as<u32>(), d_need.as<u8>());
    std::vector<u32> next(rows * n), heads(rows * n), nch(rows);
    std::vector<u8> ov(rows * n), need(rows);
    u32 herr = 0;
    HIPCHK(hipMemcpyAsync(&herr, err.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(next.data(), d_next.p, rows * n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(ov.data(), d_ov.p, rows * n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(heads.data(), d_heads.p, rows * n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(nch.data(), d_nch.p, rows * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(need.data(), d_need.p, rows, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    lap(ctx, "merge: uploads + kernels + results");
    if (herr) { gasm_set_error("contigs contain a base outside upper-case ACGT"); return GASM_ERR_NON_ACGT; }
    // ---- chains -> signatures (u32 contig, u32 overlap, u32 contig, ...); rows the kernel handed back go to the host merge.
    // Threads over the rows; every signature comes with a 64-bit hash, and the distinct ones are found by sorting the hashes
    // (equal hashes are compared in full: a collision costs a comparison, never a scaffold).  The order of `sigs` is of no
    // consequence: the scaffolds are ordered as strings on the device.
    std::vector<u32> redo;
    for (u64 r = 0; r < rows; ++r) if (need[r]) redo.insert(redo.end(), perm + r * n, perm + (r + 1) * n);
    unsigned nt = std::thread::hardware_concurrency();
    nt = std::max(1u, std::min(nt, 16u));
    if (rows < 256) nt = 1;
    struct Part { std::vector<std::string> sig; std::vector<u64> hash; };
    std::vector<Part> parts(nt);
    auto work = [&](unsigned t) {
        Part& P = parts[t];
        std::string sig;
        const u64 r0 = rows * t / nt, r1 = rows * (t + 1) / nt;
        for (u64 r = r0; r < r1; ++r) {
            if (need[r]) continue;
            for (u32 q = 0; q < nch[r]; ++q) {
                sig.clear();
                u64 h = 0xCBF29CE484222325ull;
                for (u32 x = heads[r * n + q];; x = next[r * n + x]) {
                    sig.append(reinterpret_cast<const char*>(&x), 4);
                    h = (h ^ x) * 0x100000001B3ull;
                    if (next[r * n + x] == GASM_NONE32) break;
                    const u32 o = ov[r * n + x];
                    sig.append(reinterpret_cast<const char*>(&o), 4);
                    h = (h ^ (0x9E3779B9ull + o)) * 0x100000001B3ull;
                }
                P.sig.push_back(sig);
                P.hash.push_back(h ^ (h >> 29));
            }
        }
    };
    if (nt == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t) th.emplace_back(work, t);
        for (auto& t : th) t.join();
    }
    std::vector<std::string> all;
    if (!redo.empty()) {
        if (!gasm_host::assemble_signatures(contigs, redo.data(), redo.size() / n, n, k, all)) return GASM_OK;      // (cannot happen: same preconditions)
        if (rows_on_host) *rows_on_host = redo.size() / n;
    }
    lap(ctx, "merge: signatures (host threads)");
    {
        struct Ref { u64 h; u32 part, i; };
        std::vector<Ref> refs;
        size_t total = 0;
        for (const Part& P : parts) total += P.sig.size();
        refs.reserve(total);
        for (u32 t = 0; t < nt; ++t) for (u32 i = 0; i < parts[t].sig.size(); ++i) refs.push_back(Ref{parts[t].hash[i], t, i});
        std::sort(refs.begin(), refs.end(), [](const Ref& a, const Ref& b) { return a.h < b.h; });
        const size_t host_sigs = all.size();
        for (size_t a = 0; a < refs.size();) {
            size_t b = a + 1;
            while (b < refs.size() && refs[b].h == refs[a].h) ++b;
            // the distinct strings among refs[a, b) (one, unless two signatures share a hash)
            const size_t first_kept = all.size();
            for (size_t x = a; x < b; ++x) {
                std::string& sx = parts[refs[x].part].sig[refs[x].i];
                bool seen = false;
                for (size_t y = first_kept; y < all.size() && !seen; ++y) seen = all[y] == sx;       // (against the group's distinct strings only: a popular chain comes 10 000 times)
                if (!seen) all.push_back(std::move(sx));
            }
            a = b;
        }
        if (host_sigs) { std::sort(all.begin(), all.end()); all.erase(std::unique(all.begin(), all.end()), all.end()); }      // (rows merged on the host may repeat a chain)
    }
    lap(ctx, "merge: distinct signatures");
    sigs.swap(all);
    *used = true;
    return GASM_OK;
}

int scaffolds_fetch(const gasm_scaffolds* sc, std::vector<char>& data, std::vector<u64>& off) {
    gasm_ctx* ctx = sc->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    off = sc->h_off;
    const u64 T = sc->h_off[sc->n];
    data.resize(T);
    if (T == 0) return GASM_OK;
    DBuf d;
    GCHK(d.ensure(T));
    hipLaunchKernelGGL(k_unpack_ascii, dim3(std::min<u32>(ceil_div_u64(T, GASM_WG), (u32)ctx->n_cu * 32u)), dim3(GASM_WG), 0, ctx->stream, sc->d_words.as<u64>(), T, d.as<u8>());
    hipError_t e = hipMemcpyAsync(data.data(), d.p, T, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    d.release();
    if (e != hipSuccess) { gasm_set_error("scaffold fetch failed: %s", hipGetErrorString(e)); return GASM_ERR_HIP; }
    return GASM_OK;
}

// the scaffolds as the paths of a scoring call
int scaffolds_as_paths(const gasm_scaffolds* sc, DevPaths& dp) {
    gasm_ctx* ctx = sc->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    dp.n_segments = 1;
    dp.n_paths = sc->n;
    dp.b_p_off = nullptr; dp.b_seg_path_off = nullptr; dp.b_seg_base_off = nullptr;
    dp.h_p_off = sc->h_off;
    dp.total_bases = sc->h_off[sc->n];
    if (dp.total_bases >= 0xFFFFFFF0ull) { gasm_set_error("paths exceed 2^32 bases"); return GASM_ERR_CAPACITY; }
    dp.h_seg_path_off = {0u, sc->n};
    const u64 nw = (dp.total_bases + 31) / 32 + 4;
    GCHK(dp.d_words.ensure(nw * 8));
    HIPCHK(hipMemcpyAsync(dp.d_words.p, sc->d_words.p, nw * 8, hipMemcpyDeviceToDevice, ctx->stream));
    return dp.upload_dirs(ctx);
}

// ---------------------------------------------------------------------------------------------------------------
// Row A16: breakage-score-guided traversal (kernels_asm.hip, k_guided_chain; specification in DESIGN.md §8)
// ---------------------------------------------------------------------------------------------------------------
int guided_build(gasm_ctx* ctx, DevReads& rd, BuildState& bs, DevPaths& cp, ScoreState& cs, const ScoreTable& tb, int kmer, GuidedState& g) {
    g.valid = false;
    if (!cs.graph || cs.graph != &bs) { gasm_set_error("guided traversal needs the batch scored through its graph (reads of at least k bases)"); return GASM_ERR_STATE; }
    if (rd.n_empty) { gasm_set_error("guided traversal: empty reads are not supported"); return GASM_ERR_INVALID; }
    HIPCHK(hipSetDevice(ctx->device));
    const u32 S = rd.n_segments, P = bs.n_contigs;
    const int k = bs.k;
    u32 max_c = 1;
    for (u32 s = 0; s < S; ++s) max_c = std::max(max_c, bs.h_seg_cstart[s + 1] - bs.h_seg_cstart[s]);
    if (max_c > 60000) { gasm_set_error("guided traversal: more than 60000 contigs in a segment"); return GASM_ERR_CAPACITY; }
    DBuf d_next, d_prev, tmp[5];
    struct Rel { std::vector<DBuf*> v; ~Rel() { for (DBuf* b : v) b->release(); } } rel{{&d_next, &d_prev, &tmp[0], &tmp[1], &tmp[2], &tmp[3], &tmp[4]}};
    GCHK(d_next.ensure(std::max<u32>(P, 1) * 4));
    GCHK(d_prev.ensure(std::max<u32>(P, 1) * 4));
    const size_t fx_off = (cs.stride * 4 + 15) & ~(size_t)15;
    const unsigned long long* d_fx = reinterpret_cast<const unsigned long long*>(static_cast<const char*>(cs.d_total.p) + fx_off);
    AsmLap lap;
    DBuf d_order;
    struct RelO { DBuf* b; ~RelO() { b->release(); } } relo{&d_order};
    const bool dbg_order = getenv("GASM_DBG_GUIDED") != nullptr;
    if (dbg_order) { GCHK(d_order.ensure(((size_t)P + 1) * 4)); HIPCHK(hipMemsetAsync(d_order.p, 0, ((size_t)P + 1) * 4, ctx->stream)); }
    GLAUNCH(ctx, "k_guided_chain", k_guided_chain, dim3(S), dim3(64), (size_t)max_c + 16, cp.view(), d_fx, k, d_next.as<u32>(), d_prev.as<u32>(),
            dbg_order ? d_order.as<u32>() : (u32*)nullptr);
    std::vector<u32> next(P), prev(P);
    if (P) {
        HIPCHK(hipMemcpyAsync(next.data(), d_next.p, (size_t)P * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(prev.data(), d_prev.p, (size_t)P * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    lap(ctx, "guided: chain kernel + links back");
    if (const char* dump = getenv("GASM_DBG_GUIDED")) {        // diagnostic: the links and the sums the kernel saw
        std::vector<unsigned long long> hfx(P);
        if (P) HIPCHK(hipMemcpy(hfx.data(), d_fx, (size_t)P * 8, hipMemcpyDeviceToHost));
        std::vector<u64> hoff((size_t)P + 1);
        if (P) HIPCHK(hipMemcpy(hoff.data(), cp.view().p_off, ((size_t)P + 1) * 8, hipMemcpyDeviceToHost));
        if (FILE* f = fopen(dump, "wb")) {
            std::vector<u32> hord((size_t)P + 1);      // per contig: (when it joined a path) << 2 | 0 seed, 1 right, 2 left
            HIPCHK(hipMemcpy(hord.data(), d_order.p, hord.size() * 4, hipMemcpyDeviceToHost));
            fwrite(&P, 4, 1, f); fwrite(next.data(), 4, P, f); fwrite(prev.data(), 4, P, f); fwrite(hfx.data(), 8, P, f); fwrite(hoff.data(), 8, (size_t)P + 1, f);
            fwrite(hord.data(), 4, P, f);
            fclose(f);
        }
    }
    // ---- chains -> scaffolds, per segment by descending length, ties by first contig (= lexicographic: contigs are sorted
    // and begin with distinct k-mers)
    std::vector<u64> clen(P);
    GCHK(pipeline_fetch_contigs(ctx, rd, bs));
    lap(ctx, "guided: contigs fetched");
    for (u32 c = 0; c < P; ++c) clen[c] = bs.h_c_off[c + 1] - bs.h_c_off[c];
    ChainCsr csr;
    csr.clear();
    g.h_seg_off.assign((size_t)S + 1, 0);
    struct Ch { u64 len; u32 head; };
    for (u32 s = 0; s < S; ++s) {
        std::vector<Ch> chains;
        for (u32 c = bs.h_seg_cstart[s]; c < bs.h_seg_cstart[s + 1]; ++c) {
            if (prev[c] != GASM_NONE32) continue;
            u64 len = clen[c];
            for (u32 x = c; next[x] != GASM_NONE32; x = next[x]) len += clen[next[x]] - (u64)(k - 1);
            chains.push_back(Ch{len, c});
        }
        std::stable_sort(chains.begin(), chains.end(), [](const Ch& a, const Ch& b) { return a.len > b.len; });      // heads ascend inside equal lengths
        for (const Ch& ch : chains) {
            u64 pos = 0;
            csr.elem_contig.push_back(ch.head); csr.elem_skip.push_back(0); csr.elem_pos.push_back(0);
            pos += clen[ch.head];
            for (u32 x = ch.head; next[x] != GASM_NONE32; x = next[x]) {
                csr.elem_contig.push_back(next[x]); csr.elem_skip.push_back((u32)(k - 1)); csr.elem_pos.push_back(pos);
                pos += clen[next[x]] - (u64)(k - 1);
            }
            csr.sig_off.push_back(csr.elem_contig.size());
            csr.out_off.push_back(csr.out_off.back() + pos);
        }
        g.h_seg_off[s + 1] = csr.sig_off.size() - 1;
    }
    const u32 G = (u32)(csr.sig_off.size() - 1);
    lap(ctx, "guided: chains -> CSR (host)");
    DevPaths& dp = g.dp;
    dp.n_segments = S; dp.n_paths = G;
    dp.b_p_off = nullptr; dp.b_seg_path_off = nullptr; dp.b_seg_base_off = nullptr;
    dp.h_p_off = csr.out_off;
    dp.total_bases = csr.out_off[G];
    if (dp.total_bases >= 0xFFFFFFF0ull) { gasm_set_error("guided scaffolds exceed 2^32 bases"); return GASM_ERR_CAPACITY; }
    dp.h_seg_path_off.assign((size_t)S + 1, 0);
    for (u32 s = 0; s <= S; ++s) dp.h_seg_path_off[s] = (u32)g.h_seg_off[s];
    // contigs are packed already (the scorer's copy) with their offsets on the device
    GCHK(expand(ctx, csr, cp.d_words.as<u64>(), bs.d_c_off.as<u64>(), dp.d_words, tmp));
    GCHK(dp.upload_dirs(ctx));
    lap(ctx, "guided: expand + directories");
    GCHK(pipeline_score_launch(ctx, rd, dp, kmer, tb, false, false, g.ss, nullptr));
    lap(ctx, "guided: scoring launched");
    GCHK(pipeline_score_fetch(ctx, g.ss));
    lap(ctx, "guided: scores fetched");
    g.h_text.clear(); g.h_text_off.clear();
    g.valid = true;
    return GASM_OK;
}

int guided_fetch_text(gasm_ctx* ctx, GuidedState& g) {
    if (!g.h_text_off.empty()) return GASM_OK;
    gasm_scaffolds view;
    view.ctx = ctx; view.n = g.dp.n_paths; view.h_off = g.dp.h_p_off; view.d_words = g.dp.d_words;      // (borrowed)
    const int st = scaffolds_fetch(&view, g.h_text, g.h_text_off);
    view.d_words = DBuf();
    return st;
}
