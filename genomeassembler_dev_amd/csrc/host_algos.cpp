// host_algos.cpp — the parts of the drop-in surface that stay on the host in this version:
//   * the shuffle matrix (lib/DeNovoAssembler.cpp:195-203): libstdc++'s std::shuffle + std::mt19937 define the
//     permutations, so the same calls are made here (on indices instead of strings);
//   * assemble_contigs' greedy merge (lib/DeNovoAssembler.cpp:228-304, lib/BreakageScorer.cpp:85-172) — SURVEY §8
//     row A8 / F1: host C++ now, a GPU kernel later; same visiting order, same std::sort calls, threads over
//     permutations;
//   * Levenshtein distance (lib/DeNovoAssembler.cpp:41-55 / lib/BreakageScorer.cpp:41-55 use edlib): Myers'
//     bit-parallel algorithm in 64-row blocks, global (NW) and infix (HW) — SURVEY §8 row A17 / F2.
#include <algorithm>
#include <atomic>
#include <random>
#include <thread>

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

int assemble(const std::vector<std::string>& contigs, const u32* perm, u64 rows, u64 row_len, int k, std::vector<std::string>& out) {
    const u64 n = row_len;
    std::vector<std::vector<std::string>> per(rows);
    std::atomic<u64> next(0);
    std::atomic<int> bad(0);
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    if (nt > 32) nt = 32;
    if (rows < 64) nt = 1;
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
    // lib/DeNovoAssembler.cpp:275-294: flatten, sort+unique, then the same (non-stable) std::sort by length
    out.clear();
    for (auto& v : per) for (auto& s : v) out.push_back(std::move(s));
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
