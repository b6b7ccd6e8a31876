// host_algos.cpp — the parts of the drop-in surface that stay on the host in this version:
//   * the shuffle matrix (lib/DeNovoAssembler.cpp:195-203): libstdc++'s std::shuffle + std::mt19937 define the
//     permutations, so the same calls are made here (on indices instead of strings);
//   * assemble_contigs' greedy merge (lib/DeNovoAssembler.cpp:228-304, lib/BreakageScorer.cpp:85-172) — SURVEY §8
//     row A8 / F1: host C++; same visiting order, same std::sort calls, threads over permutations.  The merge runs
//     on contig indices (suffix/prefix matches precomputed once per contig pair and overlap), strings are built once
//     per distinct chain; the string form of the loop is kept for contigs shorter than k-1;
//   * Levenshtein distance (lib/DeNovoAssembler.cpp:41-55 / lib/BreakageScorer.cpp:41-55 use edlib): Myers'
//     bit-parallel algorithm in 64-row blocks, global (NW) and infix (HW) — SURVEY §8 row A17 / F2 (the batch of
//     paths of calc_breakscore goes to k_levenshtein on the GPU; this routine serves single pairs and non-ACGT targets).
#include <algorithm>
#include <atomic>
#include <random>
#include <thread>
#include <chrono>
#include <cstdio>

#include "gasm_internal.h"

namespace gasm_host {

void shuffle_perm(u64 n, int seed, u64 rows, std::vector<u32>& perm) {
    perm.resize(rows * n);
    std::mt19937 eng(seed);
    std::vector<u32> idx(n);
    for (u64 r = 0; r < rows; ++r) {
        for (u64 i = 0; i < n; ++i) idx[i] = (u32)i;
        std::shuffle(idx.begin(), idx.end(), eng);
        std::copy(idx.begin(), idx.end(), perm.begin() + r * n);
    }
}

// ---- greedy merge of one permutation ----------------------------------------------------------------------------
// Same control flow as the reference loop; the suffix/prefix test is done on memcmp with a first-byte reject instead
// of two substr temporaries.  Returns false where the reference's substr(size-ov) would throw.
static bool merge_one(std::vector<std::string>& c, int k) {
    for (int ov = k - 1; ov > 0; --ov) {
        bool shrunk = true;
        while (shrunk) {
            const size_t before = c.size();
            for (size_t i = 0; i < c.size(); ++i) {
                if (c[i].empty()) continue;
                for (long j = (long)c.size() - 1; j >= 0; --j) {
                    const std::string& cj = c[j];
                    std::string& ci = c[i];
                    if ((long)i == j) continue;                     // c[i] != c[j] is false
                    if (ci.size() < (size_t)ov) {
                        if (ci != cj) return false;                 // substr(size - ov) out of range
                        continue;
                    }
                    if (cj.size() < (size_t)ov) continue;           // prefix shorter than the suffix: never equal
                    const char* tail = ci.data() + ci.size() - ov;
                    if (tail[0] != cj[0] || memcmp(tail, cj.data(), ov) != 0) continue;
                    if (ci == cj) continue;
                    ci.append(cj, ov, std::string::npos);
                    c[j].clear();
                }
            }
            size_t w = 0;
            for (size_t i = 0; i < c.size(); ++i)
                if (!c[i].empty()) { if (w != i) c[w] = std::move(c[i]); ++w; }
            c.resize(w);
            shrunk = before != c.size();
        }
    }
    return true;
}

// ---- the same merge on contig indices ------------------------------------------------------------------------------
// A merged string is a chain of whole contigs: c_i + c_j[ov:] ends with all of c_j (the overlap is shared), and it still
// begins with all of c_i.  As long as no contig is shorter than the overlap being tried, "suffix(c_i, ov) == prefix(c_j, ov)"
// is therefore a statement about the chain's LAST contig and the other chain's FIRST contig only — it does not depend
// on what was merged before.  So the suffix/prefix tests of all (contig, contig, overlap) triples are done once per
// call (match[ov][a][b]), and a permutation's merge becomes the reference's loop over small integers: no string is
// built until the end, and equal chains (same contigs, same overlaps — most permutations end in the same few dozen
// scaffolds) are materialised once.  The full-string test `c[i] != c[j]` can only fail for chains of equal length;
// those are built and compared (rare).  Contigs shorter than k-1 take the string version above (merge_one).
struct ChainLinks {
    std::vector<u32> next, link_ov;      // per contig: the contig appended to it and the overlap dropped, or NONE
};
static const u32 NONE = 0xFFFFFFFFu;
struct Chain { u32 head, tail; u64 len; };

static void chain_string(const std::vector<std::string>& contigs, const ChainLinks& L, u32 head, std::string& out) {
    out = contigs[head];
    for (u32 c = head; L.next[c] != NONE; c = L.next[c]) out.append(contigs[L.next[c]], L.link_ov[c], std::string::npos);
}

// match[(ov * n + a) * n + b] for ov in [1, k-1]
// row_any[ov * n + a]: contig a's suffix of length ov is some contig's prefix (most rows and most overlaps have no match
// at all — real overlaps are the k-1 bases two contigs share at a branching node — and are skipped outright)
static void merge_indices(const std::vector<std::string>& contigs, const std::vector<u8>& match, const std::vector<u8>& row_any,
                          const std::vector<u8>& level_any, u64 n, const u32* row, int k,
                          ChainLinks& L, std::vector<Chain>& c, std::string& tmp_a, std::string& tmp_b) {
    L.next.assign(n, NONE);
    L.link_ov.assign(n, 0);
    c.resize(n);
    std::vector<u8> empty(n, 0);
    for (u64 p = 0; p < n; ++p) c[p] = Chain{row[p], row[p], (u64)contigs[row[p]].size()};
    for (int ov = k - 1; ov > 0; --ov) {
        if (!level_any[ov]) continue;                 // nothing can merge at this overlap: the reference's pass changes nothing
        const u8* M = &match[(size_t)ov * n * n];
        const u8* RA = &row_any[(size_t)ov * n];
        bool shrunk = true;
        while (shrunk) {
            const size_t before = c.size();
            empty.assign(c.size(), 0);
            for (size_t i = 0; i < c.size(); ++i) {
                if (empty[i] || !RA[c[i].tail]) continue;
                for (long j = (long)c.size() - 1; j >= 0; --j) {
                    if ((long)i == j || empty[j]) continue;               // (an emptied string is shorter than any overlap)
                    if (!M[(size_t)c[i].tail * n + c[j].head]) continue;
                    if (c[i].len == c[j].len) {                           // c[i] == c[j]?  Only possible at equal length
                        chain_string(contigs, L, c[i].head, tmp_a);
                        chain_string(contigs, L, c[j].head, tmp_b);
                        if (tmp_a == tmp_b) continue;
                    }
                    L.next[c[i].tail] = c[j].head;
                    L.link_ov[c[i].tail] = (u32)ov;
                    c[i].tail = c[j].tail;
                    c[i].len += c[j].len - (u64)ov;
                    empty[j] = 1;
                    if (!RA[c[i].tail]) break;                            // the new tail matches nothing: the rest of the scan is idle
                }
            }
            size_t w = 0;
            for (size_t i = 0; i < c.size(); ++i)
                if (!empty[i]) { c[w] = c[i]; ++w; }
            c.resize(w);
            shrunk = before != c.size();
        }
    }
}

bool assemble_signatures(const std::vector<std::string>& contigs, const u32* perm, u64 rows, u64 row_len, int k, std::vector<std::string>& all) {
    const u64 n = row_len;
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    if (nt > 32) nt = 32;
    if (rows < 64) nt = 1;
    size_t min_len = ~(size_t)0;
    for (const std::string& s : contigs) min_len = std::min(min_len, s.size());
    all.clear();
    if (!(n > 0 && k >= 2 && min_len >= (size_t)(k - 1) && n <= 4096)) return false;
    std::atomic<u64> next(0);
    {
        const bool timing = false;
        auto tnow = []() { return std::chrono::steady_clock::now(); };
        auto t0 = tnow();
        auto lap = [&](const char* what) { if (timing) { auto t1 = tnow(); fprintf(stderr, "[assemble] %-22s %8.1f ms\n", what, std::chrono::duration<double>(t1 - t0).count() * 1e3); t0 = t1; } };
        // suffix/prefix matches of every (overlap, contig, contig), once
        std::vector<u8> match((size_t)k * n * n, 0), row_any((size_t)k * n, 0), level_any((size_t)k, 0);
        for (int ov = 1; ov < k; ++ov)
            for (u64 a = 0; a < n; ++a) {
                const char* tail = contigs[a].data() + contigs[a].size() - ov;
                for (u64 b = 0; b < n; ++b) {
                    const bool m = tail[0] == contigs[b][0] && memcmp(tail, contigs[b].data(), ov) == 0;
                    match[((size_t)ov * n + a) * n + b] = m;
                    if (m && a != b) { row_any[(size_t)ov * n + a] = 1; level_any[ov] = 1; }
                }
            }
        // every permutation's final chains as signatures (contig, overlap, contig, ...): equal signatures = equal strings
        std::vector<std::vector<std::string>> sigs(nt);
        auto work = [&](unsigned t) {
            ChainLinks L;
            std::vector<Chain> c;
            std::string ta, tb, sig;
            while (true) {
                const u64 r = next.fetch_add(1);
                if (r >= rows) break;
                merge_indices(contigs, match, row_any, level_any, n, perm + r * n, k, L, c, ta, tb);
                for (const Chain& ch : c) {
                    sig.clear();
                    for (u32 x = ch.head;; x = L.next[x]) {
                        sig.append(reinterpret_cast<const char*>(&x), 4);
                        if (L.next[x] == NONE) break;
                        sig.append(reinterpret_cast<const char*>(&L.link_ov[x]), 4);
                    }
                    sigs[t].push_back(sig);
                }
            }
        };
        if (nt == 1) work(0);
        else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; ++t) th.emplace_back(work, t);
            for (auto& t : th) t.join();
        }
        lap("match + merges");
        for (auto& v : sigs) for (auto& s : v) all.push_back(std::move(s));
        std::sort(all.begin(), all.end());
        all.erase(std::unique(all.begin(), all.end()), all.end());
        lap("signature sort+unique");
    }
    return true;
}

// The index-form merge alone: every permutation's final chains as signatures (contig, overlap, contig, ... as 32-bit
// words), sorted and de-duplicated.  false = the index form does not apply (a contig shorter than k-1, or more than 4096
// contigs): the caller takes the string form.
bool assemble_signatures(const std::vector<std::string>& contigs, const u32* perm, u64 rows, u64 row_len, int k, std::vector<std::string>& sigs_out);

int assemble(const std::vector<std::string>& contigs, const u32* perm, u64 rows, u64 row_len, int k, std::vector<std::string>& out) {
    const u64 n = row_len;
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    if (nt > 32) nt = 32;
    if (rows < 64) nt = 1;
    size_t min_len = ~(size_t)0;
    for (const std::string& s : contigs) min_len = std::min(min_len, s.size());
    const bool by_index = n > 0 && k >= 2 && min_len >= (size_t)(k - 1) && n <= 4096;
    std::atomic<u64> next(0);
    std::atomic<int> bad(0);
    out.clear();
    if (!by_index) {
        // contigs shorter than an overlap (the reference throws there or compares whole strings): the string version
        std::vector<std::vector<std::string>> per(rows);
        auto work = [&]() {
            std::vector<std::string> c;
            while (true) {
                const u64 r = next.fetch_add(1);
                if (r >= rows || bad.load()) break;
                c.resize(n);
                for (u64 j = 0; j < n; ++j) c[j] = contigs[perm[r * n + j]];
                if (!merge_one(c, k)) { bad.store(1); break; }
                per[r] = c;
            }
        };
        if (nt == 1) work();
        else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; ++t) th.emplace_back(work);
            for (auto& t : th) t.join();
        }
        if (bad.load()) {
            gasm_set_error("assemble_contigs: a contig is shorter than the overlap being tried (the reference throws std::out_of_range here)");
            return GASM_ERR_RANGE;
        }
        for (auto& v : per) for (auto& s : v) out.push_back(std::move(s));
    } else {
        const bool timing = getenv("GASM_ASM_TIMING") != nullptr;
        auto tnow = []() { return std::chrono::steady_clock::now(); };
        auto t0 = tnow();
        auto lap = [&](const char* what) { if (timing) { auto t1 = tnow(); fprintf(stderr, "[assemble] %-22s %8.1f ms\n", what, std::chrono::duration<double>(t1 - t0).count() * 1e3); t0 = t1; } };
        std::vector<std::string> all;
        assemble_signatures(contigs, perm, rows, row_len, k, all);
        lap("signatures");
        // one string per distinct signature (threads over signatures)
        out.resize(all.size());
        std::atomic<u64> nx(0);
        auto build = [&]() {
            while (true) {
                const u64 i = nx.fetch_add(1);
                if (i >= all.size()) break;
                const std::string& sg = all[i];
                std::string& o = out[i];
                u32 x;
                memcpy(&x, sg.data(), 4);
                size_t total = contigs[x].size();
                for (size_t q = 4; q + 8 <= sg.size(); q += 8) {
                    u32 ov, y;
                    memcpy(&ov, sg.data() + q, 4);
                    memcpy(&y, sg.data() + q + 4, 4);
                    total += contigs[y].size() - ov;
                }
                o.clear();
                o.reserve(total);
                o = contigs[x];
                for (size_t q = 4; q + 8 <= sg.size(); q += 8) {
                    u32 ov, y;
                    memcpy(&ov, sg.data() + q, 4);
                    memcpy(&y, sg.data() + q + 4, 4);
                    o.append(contigs[y], ov, std::string::npos);
                }
            }
        };
        if (nt == 1) build();
        else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; ++t) th.emplace_back(build);
            for (auto& t : th) t.join();
        }
        lap("strings built");
    }
    // lib/DeNovoAssembler.cpp:275-294: flatten, sort+unique, then the same (non-stable) std::sort by length.  (Dropping
    // duplicates earlier does not change the sorted distinct list the last sort starts from.)
    std::sort(out.begin(), out.end());
    out.erase(std::unique(out.begin(), out.end()), out.end());
    std::sort(out.begin(), out.end(), [](const std::string& a, const std::string& b) { return a.length() > b.length(); });
    return GASM_OK;
}

// ---- Myers bit-parallel edit distance ------------------------------------------------------------------------------
int levenshtein(const char* q, u64 nq, const char* t, u64 nt, bool infix) {
    if (nq == 0 || nt == 0) return 0;  // edlib reports an error, the reference then returns 0
    const u64 nblk = (nq + 63) / 64;
    // match masks per block and symbol (bytes folded to 256 symbols)
    std::vector<u64> peq(nblk * 256, 0);
    for (u64 i = 0; i < nq; ++i) peq[(i / 64) * 256 + (unsigned char)q[i]] |= 1ull << (i % 64);
    std::vector<u64> pv(nblk, ~0ull), mv(nblk, 0);
    const u64 top_last = 1ull << ((nq - 1) % 64);
    long long score = (long long)nq, best = (long long)nq;
    for (u64 j = 0; j < nt; ++j) {
        const unsigned char ch = (unsigned char)t[j];
        int hin = infix ? 0 : 1;
        for (u64 b = 0; b < nblk; ++b) {
            u64 eq = peq[b * 256 + ch];
            const u64 Pv = pv[b], Mv = mv[b];
            const u64 xv = eq | Mv;
            if (hin < 0) eq |= 1;
            const u64 xh = (((eq & Pv) + Pv) ^ Pv) | eq;
            u64 ph = Mv | ~(xh | Pv);
            u64 mh = Pv & xh;
            const u64 top = b + 1 == nblk ? top_last : (1ull << 63);
            int hout = 0;
            if (ph & top) hout = 1;
            else if (mh & top) hout = -1;
            ph <<= 1;
            mh <<= 1;
            if (hin < 0) mh |= 1;
            else if (hin > 0) ph |= 1;
            pv[b] = mh | ~(xv | ph);
            mv[b] = ph & xv;
            hin = hout;
        }
        score += hin;
        if (score < best) best = score;
    }
    return (int)(infix ? best : score);
}

}  // namespace gasm_host
