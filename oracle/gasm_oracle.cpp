// ============================================================================
// gasm_oracle.cpp — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
//
// A single-threaded std::string + hash-map restatement of the reference hot
// path (SahakyanLab/GenomeAssembler_dev, lib/DeNovoAssembler.cpp and
// lib/BreakageScorer.cpp, plus the three R idioms that feed it).  It keeps the
// reference's data structures and asymptotics on purpose, so that it can be
// (a) the checker for the HIP path in tests/ and __graft_entry__.smoke(), and
// (b) the "port" CPU baseline timed by bench.py's cpu_baseline leg.
// Nothing under genomeassembler_dev_amd/ may import, link or call this file.
//
// PARITY STATUS: the reference ships no tests, fixtures or golden outputs, and
// it cannot be built in this image (it needs Rcpp, gtl/phmap.hpp and edlib,
// none of which are present; stand-in headers are not allowed).  The only
// reference-produced output available is the toy known-answer vector recorded
// in SURVEY.md §8(c); tests/test_oracle_kat.py pins this oracle to it.
// Everything beyond that vector is "parity unpinned" (see DESIGN.md §3).
//
// std::unordered_map stands in for gtl::flat_hash_map.  The two differ only
// in iteration order, which leaks into (i) the FP summation order of
// bp_score / norm (≤1e-12, inside the 1e-9 tolerance) and (ii) the element
// order of path_freq (compare as a multiset or via the by-input-order view).
//
// Every function cites the reference lines it restates (paths relative to
// the reference root).
// ============================================================================
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace orc {

using StrVec = std::vector<std::string>;

// ---------------------------------------------------------------------------
// A1 — lib/DeNovoAssembler.R:109-130 (get_kmers_from_reads): every k-mer of
// every read, read-major, left to right, duplicates kept.  R's 1:end with
// end < 1 counts downwards; reads shorter than k are not produced by the
// reference simulator, here they simply yield nothing.
// ---------------------------------------------------------------------------
StrVec kmers_from_reads(const StrVec& reads, int k) {
    StrVec out;
    for (const std::string& r : reads) {
        if ((int)r.size() < k) continue;
        for (size_t p = 0; p + k <= r.size(); ++p) out.emplace_back(r, p, k);
    }
    return out;
}

// lib/DeNovoAssembler.cpp:62-71 (remove_duplicates): sort + unique, plain
// byte-wise lexicographic order, shorter prefix first.
template <class T>
static void sort_unique(std::vector<T>& v) {
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
}

struct DbgDetail {
    StrVec edge_prefix, edge_suffix;        // distinct edges, sorted by (prefix,suffix)
    StrVec node;                            // every node, sorted
    std::vector<int> node_in, node_out;     // degrees over distinct edges
    StrVec branch;                          // sorted branching nodes that have out-edges
};

// ---------------------------------------------------------------------------
// A2–A6 — lib/DeNovoAssembler.cpp:91-192.  Returns the sorted unique contigs.
// ---------------------------------------------------------------------------
StrVec dbg_contigs(const StrVec& read_kmers, int k, DbgDetail* detail) {
    const size_t n = read_kmers.size();
    // :91-101  (k-1)-prefix and (k-1)-suffix of each k-mer; substr clamps.
    StrVec pre(n), suf(n);
    for (size_t i = 0; i < n; ++i) {
        pre[i] = read_kmers[i].substr(0, k - 1);
        suf[i] = read_kmers[i].size() >= 1 ? read_kmers[i].substr(1, k) : std::string();
    }
    // :104-122  adjacency: node -> distinct successors in first-seen order.
    std::unordered_map<std::string, StrVec> adj;
    for (size_t i = 0; i < n; ++i) {
        StrVec& lst = adj[pre[i]];
        if (std::find(lst.begin(), lst.end(), suf[i]) == lst.end()) lst.push_back(suf[i]);
    }
    // :125-142  node set = unique prefixes followed by unique suffixes.
    sort_unique(pre);
    sort_unique(suf);
    StrVec nodes = pre;
    nodes.insert(nodes.end(), suf.begin(), suf.end());
    std::unordered_map<std::string, std::pair<int, int>> deg;  // (in, out)
    for (const std::string& v : nodes) deg[v] = {0, 0};
    // :146-158  degrees over distinct edges.
    for (const auto& kv : adj) {
        deg[kv.first].second += (int)kv.second.size();
        for (const std::string& w : kv.second) deg[w].first++;
    }
    // :161-169  branching nodes: (in != 1 or out != 1) and the node has out-edges.
    StrVec branch;
    for (const auto& kv : deg)
        if ((kv.second.first != 1 || kv.second.second != 1) && adj.count(kv.first))
            branch.push_back(kv.first);
    // :172-189  one contig per (branching node, out-edge): follow the single
    // successor until a branching node or a node without successors; the
    // contig is the start node plus the last character of every node visited.
    // (Linear std::find over the branching list, as in the reference.)
    StrVec contigs;
    for (const std::string& start : branch) {
        const StrVec outs = adj[start];
        for (const std::string& first : outs) {
            std::string cur = first, path = start;
            while (std::find(branch.begin(), branch.end(), cur) == branch.end()) {
                auto it = adj.find(cur);
                if (it == adj.end() || it->second.empty()) break;
                path.push_back(cur.back());
                cur = it->second[0];
            }
            path.push_back(cur.back());
            contigs.push_back(path);
        }
    }
    // :192
    sort_unique(contigs);

    if (detail) {
        std::vector<std::pair<std::string, std::string>> e;
        for (const auto& kv : adj)
            for (const std::string& w : kv.second) e.emplace_back(kv.first, w);
        std::sort(e.begin(), e.end());
        for (auto& p : e) { detail->edge_prefix.push_back(p.first); detail->edge_suffix.push_back(p.second); }
        StrVec nn = nodes;
        sort_unique(nn);
        for (const std::string& v : nn) {
            detail->node.push_back(v);
            detail->node_in.push_back(deg[v].first);
            detail->node_out.push_back(deg[v].second);
        }
        detail->branch = branch;
        std::sort(detail->branch.begin(), detail->branch.end());
    }
    return contigs;
}

// A7 — lib/DeNovoAssembler.cpp:195-203: `rows` permutations of the contig
// list, one std::mt19937(seed) engine shared by consecutive std::shuffle calls.
std::vector<StrVec> shuffle_matrix(const StrVec& contigs, int seed, int rows) {
    std::mt19937 eng(seed);
    std::vector<StrVec> m(rows);
    for (int i = 0; i < rows; ++i) {
        StrVec c = contigs;
        std::shuffle(c.begin(), c.end(), eng);
        m[i] = std::move(c);
    }
    return m;
}

// ---------------------------------------------------------------------------
// A8 — lib/DeNovoAssembler.cpp:228-272 (identical loop in
// lib/BreakageScorer.cpp:105-149): greedy suffix/prefix merge of one
// permutation, overlap k-1 … 1, repeated until the list stops shrinking.
// std::string::substr(pos) with pos > size() throws std::out_of_range, as in
// the reference (where Rcpp turns it into an R error).
// ---------------------------------------------------------------------------
static StrVec greedy_merge(StrVec c, int k) {
    for (int ov = k - 1; ov > 0; --ov) {
        bool shrunk = true;
        while (shrunk) {
            const size_t before = c.size();
            for (size_t i = 0; i < c.size(); ++i) {
                if (c[i].empty()) continue;
                for (long j = (long)c.size() - 1; j >= 0; --j) {
                    if (c[i] != c[j]) {
                        std::string tail = c[i].substr(c[i].size() - ov, c[i].size());
                        std::string head = c[j].substr(0, ov);
                        if (tail == head) {
                            c[i].append(c[j].substr(ov, c[j].size()));
                            c[j].clear();
                        }
                    }
                }
            }
            for (long i = (long)c.size() - 1; i >= 0; --i)
                if (c[i].empty()) c.erase(c.begin() + i);
            shrunk = before != c.size();
        }
    }
    return c;
}

// lib/DeNovoAssembler.cpp:275-304 — flatten, sort+unique, then the
// (non-stable) std::sort by descending length; 100 % of the result is kept.
static StrVec finish_scaffolds(const std::vector<StrVec>& per_perm) {
    StrVec flat;
    for (const StrVec& v : per_perm) flat.insert(flat.end(), v.begin(), v.end());
    sort_unique(flat);
    std::sort(flat.begin(), flat.end(),
              [](const std::string& a, const std::string& b) { return a.length() > b.length(); });
    return flat;
}

// lib/DeNovoAssembler.cpp:215-305 — assemble_contigs(contig_matrix, dbg_kmer)
StrVec assemble_matrix(const std::vector<StrVec>& matrix, int k) {
    std::vector<StrVec> out(matrix.size());
    for (size_t r = 0; r < matrix.size(); ++r) out[r] = greedy_merge(matrix[r], k);
    return finish_scaffolds(out);
}

// lib/BreakageScorer.cpp:80-174 — assemble_contigs(velvet_contigs, dbg_kmer,
// seed): 20 000 internal shuffles, then the same merge and ordering.
StrVec assemble_velvet(const StrVec& contigs, int k, int seed, int rows) {
    return assemble_matrix(shuffle_matrix(contigs, seed, rows), k);
}

// ---------------------------------------------------------------------------
// A17 — calc_levenshtein (lib/DeNovoAssembler.cpp:41-55 NW,
// lib/BreakageScorer.cpp:41-55 HW).  edlib is not in the tree; the edit
// distance is mathematically unique, so a plain two-row DP restates it.
// NW: global.  HW: query against any infix of target (free target end gaps).
// edlib reports failure (→ the reference returns 0) for an empty query or
// target; mirrored here.
// ---------------------------------------------------------------------------
int levenshtein(const std::string& q, const std::string& t, bool infix) {
    if (q.empty() || t.empty()) return 0;
    const size_t n = q.size(), m = t.size();
    std::vector<int> prev(m + 1), cur(m + 1);
    for (size_t j = 0; j <= m; ++j) prev[j] = infix ? 0 : (int)j;
    for (size_t i = 1; i <= n; ++i) {
        cur[0] = (int)i;
        for (size_t j = 1; j <= m; ++j) {
            int sub = prev[j - 1] + (q[i - 1] != t[j - 1]);
            int del = prev[j] + 1, ins = cur[j - 1] + 1;
            cur[j] = std::min(sub, std::min(del, ins));
        }
        std::swap(prev, cur);
    }
    if (!infix) return prev[m];
    return *std::min_element(prev.begin(), prev.end());
}

// ---------------------------------------------------------------------------
// A9–A13 — calc_breakscore (lib/DeNovoAssembler.cpp:317-477; velvet deltas
// lib/BreakageScorer.cpp:186-353).
// ---------------------------------------------------------------------------
struct BreakScores {
    std::vector<int> sequence_len;
    std::vector<double> bp_score, norm_by_break_freqs, norm_by_len;
    std::vector<int> kmer_breaks, lev_dist;
    // own-assembler variant: path_freq in the map's iteration order, plus the
    // same numbers re-ordered to follow bp_kmer (oracle convenience view).
    std::vector<std::vector<double>> path_freq, path_freq_by_input;
    // velvet variant
    std::vector<int> startpos;
    std::vector<std::vector<double>> prob_dist;
};

BreakScores calc_breakscore(const StrVec& path, const StrVec& reads, const std::string& truth,
                            int kmer, const StrVec& bp_kmer, const std::vector<double>& bp_prob,
                            bool velvet, bool with_lev) {
    BreakScores R;
    const size_t P = path.size();
    // :325-328  table: kmer -> (prob, running count)
    std::unordered_map<std::string, std::pair<double, int>> tbl;
    for (size_t i = 0; i < bp_kmer.size(); ++i) tbl[bp_kmer[i]] = {bp_prob[i], 0};

    // BreakageScorer.cpp:200-215  rolling probability per path position
    if (velvet) {
        R.prob_dist.resize(P);
        for (size_t i = 0; i < P; ++i) {
            if ((long)path[i].size() < kmer) continue;  // reference underflows here
            std::vector<double>& d = R.prob_dist[i];
            d.resize(path[i].size() - kmer + 1);
            for (size_t pos = 0; pos + kmer <= path[i].size(); ++pos) {
                auto it = tbl.find(path[i].substr(pos, kmer));
                d[pos] = it == tbl.end() ? 0.0 : it->second.first;
            }
        }
    }
    // :334-337  unique reads with multiplicities
    std::unordered_map<std::string, int> uniq;
    for (const std::string& r : reads) uniq[r]++;

    R.sequence_len.resize(P);
    R.bp_score.assign(P, 0.0);
    R.norm_by_break_freqs.assign(P, 0.0);
    R.norm_by_len.assign(P, 0.0);
    R.kmer_breaks.assign(P, 0);
    R.lev_dist.assign(P, 0);
    if (velvet) R.startpos.assign(P, 0);
    else { R.path_freq.resize(P); R.path_freq_by_input.resize(P); }

    for (size_t i = 0; i < P; ++i) {
        int total = 0;
        for (const auto& kv : uniq) {
            // :360  first exact occurrence only
            size_t pos = path[i].find(kv.first);
            if (pos == std::string::npos) continue;
            // :366-381  window start and width
            int start = std::max(0, (int)pos - kmer / 2);
            int width = 8;
            if (start == 0) {
                if (pos == 1) width = 2;
                else if (pos == 2) width = 4;
                else if (pos == 3) width = 6;
            }
            // :386-390  (operator[] inserts a zero-probability entry for a key
            // that is not in the table; kept, it then only feeds `total`)
            tbl[path[i].substr(start, width)].second += kv.second;
            total += kv.second;
            // BreakageScorer.cpp:273-274
            if (velvet) R.startpos[i] = (int)truth.find(path[i]);
        }
        // :394-420  weighted sums in map order, counters reset on the way
        std::vector<double> freq;
        std::unordered_map<std::string, double> freq_of;
        if (!velvet) freq.reserve(tbl.size());
        for (auto& kv : tbl) {
            double prob = kv.second.first;
            double cnt = kv.second.second;
            if (!velvet) { freq.push_back(cnt / (double)total); freq_of[kv.first] = cnt / (double)total; }
            if (cnt != 0) {
                R.bp_score[i] += prob * cnt;
                R.norm_by_break_freqs[i] += prob * (cnt / (double)total);
                kv.second.second = 0;
            }
        }
        R.kmer_breaks[i] = total;
        R.sequence_len[i] = (int)path[i].length();
        R.norm_by_len[i] = R.bp_score[i] / R.sequence_len[i];  // :425
        if (!velvet) {
            R.path_freq[i] = std::move(freq);
            std::vector<double>& bi = R.path_freq_by_input[i];
            bi.resize(bp_kmer.size());
            for (size_t j = 0; j < bp_kmer.size(); ++j) bi[j] = freq_of[bp_kmer[j]];
        }
        // :463 / BreakageScorer.cpp:339
        if (with_lev) R.lev_dist[i] = levenshtein(path[i], truth, velvet);
    }
    return R;
}

// ---------------------------------------------------------------------------
// A14 — lib/DeNovoAssembler.R:135-168 (count_read_kmers): occurrences of every
// length-k window over all reads, reported against a key list, absent -> 0.
// ---------------------------------------------------------------------------
std::vector<int64_t> count_windows(const StrVec& reads, int k, const StrVec& keys) {
    std::unordered_map<std::string, int64_t> c;
    for (const std::string& r : reads)
        for (size_t p = 0; p + k <= r.size(); ++p) c[r.substr(p, k)]++;
    std::vector<int64_t> out(keys.size(), 0);
    for (size_t i = 0; i < keys.size(); ++i) {
        auto it = c.find(keys[i]);
        if (it != c.end()) out[i] = it->second;
    }
    return out;
}

// "k-mer counts" of the build: multiplicity of each
// distinct k-mer, keys sorted.
void distinct_counts(const StrVec& kmers, StrVec& keys, std::vector<int64_t>& counts) {
    std::unordered_map<std::string, int64_t> c;
    for (const std::string& s : kmers) c[s]++;
    keys.clear();
    for (const auto& kv : c) keys.push_back(kv.first);
    std::sort(keys.begin(), keys.end());
    counts.resize(keys.size());
    for (size_t i = 0; i < keys.size(); ++i) counts[i] = c[keys[i]];
}

// ---------------------------------------------------------------------------
// A15 — lib/GenerateReads.R:153-184 (get_prob_values): NaN -> minimum of its
// own table, then divide every value by the sum over all rows.  R's sum()
// accumulates doubles in long double; mirrored.
// `tables` = the per-k tables in file order (k = 2,4,6,8), concatenated.
// ---------------------------------------------------------------------------
void normalise_tables(std::vector<double>& prob, const std::vector<size_t>& table_sizes) {
    size_t off = 0;
    for (size_t n : table_sizes) {
        double lo = 0; bool have = false, any_nan = false;
        for (size_t i = off; i < off + n; ++i) {
            if (prob[i] != prob[i]) { any_nan = true; continue; }
            if (!have || prob[i] < lo) { lo = prob[i]; have = true; }
        }
        if (any_nan) for (size_t i = off; i < off + n; ++i) if (prob[i] != prob[i]) prob[i] = lo;
        off += n;
    }
    long double s = 0;
    for (double p : prob) s += p;
    const double ds = (double)s;
    for (double& p : prob) p /= ds;
}

}  // namespace orc

// ============================================================================
// C wrappers for ctypes.  Results come back as one malloc'd blob:
//   repeat { u64 tag; u64 nbytes; payload padded to 8 bytes }  ... tag 0 ends.
// String lists are '\n'-joined; numeric vectors are raw little-endian arrays;
// lists of vectors are (u64 count, u64 lengths[count], values...).
// ============================================================================
namespace {

struct Blob {
    std::vector<unsigned char> b;
    void raw(const void* p, size_t n) {
        const unsigned char* c = (const unsigned char*)p;
        b.insert(b.end(), c, c + n);
    }
    void section(uint64_t tag, const void* p, uint64_t n) {
        raw(&tag, 8); raw(&n, 8); raw(p, n);
        while (b.size() % 8) b.push_back(0);
    }
    void strs(uint64_t tag, const orc::StrVec& v) {
        std::string j;
        for (size_t i = 0; i < v.size(); ++i) { if (i) j.push_back('\n'); j += v[i]; }
        // an empty list and a list holding one empty string both join to "";
        // prefix the count to tell them apart
        std::string s = std::to_string(v.size()) + "\n" + j;
        section(tag, s.data(), s.size());
    }
    template <class T> void vec(uint64_t tag, const std::vector<T>& v) { section(tag, v.data(), v.size() * sizeof(T)); }
    void vecs(uint64_t tag, const std::vector<std::vector<double>>& vv) {
        std::vector<unsigned char> t;
        auto put = [&](const void* p, size_t n) { const unsigned char* c = (const unsigned char*)p; t.insert(t.end(), c, c + n); };
        uint64_t cnt = vv.size(); put(&cnt, 8);
        for (auto& v : vv) { uint64_t l = v.size(); put(&l, 8); }
        for (auto& v : vv) put(v.data(), v.size() * 8);
        section(tag, t.data(), t.size());
    }
    unsigned char* finish(uint64_t* nbytes) {
        uint64_t z = 0; raw(&z, 8); raw(&z, 8);
        unsigned char* out = (unsigned char*)std::malloc(b.size());
        std::memcpy(out, b.data(), b.size());
        *nbytes = b.size();
        return out;
    }
};

// strings arrive as one buffer + (n+1) offsets
orc::StrVec unpack(const char* data, const uint64_t* off, uint64_t n) {
    orc::StrVec v(n);
    for (uint64_t i = 0; i < n; ++i) v[i].assign(data + off[i], data + off[i + 1]);
    return v;
}

}  // namespace

extern "C" {

void orc_free(void* p) { std::free(p); }

// tags: 1 kmers
unsigned char* orc_kmers_from_reads(const char* rd, const uint64_t* roff, uint64_t nr, int k, uint64_t* nbytes) {
    Blob B; B.strs(1, orc::kmers_from_reads(unpack(rd, roff, nr), k));
    return B.finish(nbytes);
}

// tags: 1 contigs, 2 edge_prefix, 3 edge_suffix, 4 node, 5 node_in(i32), 6 node_out(i32), 7 branch,
//       8 shuffle matrix as u32 indices into contigs (rows*C), 9 distinct kmers, 10 counts(i64)
unsigned char* orc_get_contigs(const char* kd, const uint64_t* koff, uint64_t nk, int k, int seed, int rows,
                               uint64_t* nbytes) {
    orc::StrVec km = unpack(kd, koff, nk);
    orc::DbgDetail D;
    orc::StrVec contigs = orc::dbg_contigs(km, k, &D);
    Blob B;
    B.strs(1, contigs); B.strs(2, D.edge_prefix); B.strs(3, D.edge_suffix); B.strs(4, D.node);
    B.vec(5, D.node_in); B.vec(6, D.node_out); B.strs(7, D.branch);
    // the matrix itself is rows × contigs strings; ship it as indices
    std::vector<orc::StrVec> M = orc::shuffle_matrix(contigs, seed, rows);
    std::unordered_map<std::string, uint32_t> idx;
    for (uint32_t i = 0; i < contigs.size(); ++i) idx[contigs[i]] = i;
    std::vector<uint32_t> perm; perm.reserve((size_t)rows * contigs.size());
    for (auto& row : M) for (auto& s : row) perm.push_back(idx[s]);
    B.vec(8, perm);
    orc::StrVec keys; std::vector<int64_t> cnt;
    orc::distinct_counts(km, keys, cnt);
    B.strs(9, keys); B.vec(10, cnt);
    return B.finish(nbytes);
}

// timing entry for bench.py's cpu_baseline leg: the reference's get_contigs
// including its `rows` shuffled copies, result discarded.  Returns #contigs.
uint64_t orc_time_get_contigs(const char* kd, const uint64_t* koff, uint64_t nk, int k, int seed, int rows) {
    orc::StrVec km = unpack(kd, koff, nk);
    orc::StrVec contigs = orc::dbg_contigs(km, k, nullptr);
    std::vector<orc::StrVec> M = orc::shuffle_matrix(contigs, seed, rows);
    return contigs.size() + (M.empty() ? 0 : 0);
}

// timing entry for bench.py's cpu_baseline leg: what one GPU step does for one segment, the reference's way —
// k-mers of the reads (lib/DeNovoAssembler.R:109-130), contigs (lib/DeNovoAssembler.cpp:91-192, no shuffle), then
// calc_breakscore of those contigs against the reads (lib/DeNovoAssembler.cpp:325-426, no Levenshtein).
// Returns the number of k-mers processed; *checksum gets sum(kmer_breaks) so the work cannot be optimised away.
uint64_t orc_time_build_score(const char* rd, const uint64_t* roff, uint64_t nr, int k, int kmer, const char* kd,
                              const uint64_t* koff, uint64_t nk, const double* prob, uint64_t* checksum) {
    orc::StrVec reads = unpack(rd, roff, nr);
    orc::StrVec km = orc::kmers_from_reads(reads, k);
    orc::StrVec contigs = orc::dbg_contigs(km, k, nullptr);
    orc::StrVec bpk = unpack(kd, koff, nk);
    std::vector<double> bpp(prob, prob + nk);
    orc::BreakScores R = orc::calc_breakscore(contigs, reads, std::string(), kmer, bpk, bpp, false, false);
    uint64_t cs = 0;
    for (int v : R.kmer_breaks) cs += (uint64_t)v;
    *checksum = cs + contigs.size();
    return km.size();
}

// The same work with its results handed back (bench.py's self-check and cpu_baseline leg in one call): tags 1 contigs,
// 2 kmer_breaks(i32), 3 bp_score(f64), 4 [n_kmers](u64).  *seconds = time of the three stages alone
// (steady_clock around k-mers + contigs + scoring; unpacking of the arguments and packing of the results excluded).
unsigned char* orc_build_score(const char* rd, const uint64_t* roff, uint64_t nr, int k, int kmer, const char* kd,
                               const uint64_t* koff, uint64_t nk, const double* prob, double* seconds, uint64_t* nbytes) {
    orc::StrVec reads = unpack(rd, roff, nr);
    orc::StrVec bpk = unpack(kd, koff, nk);
    std::vector<double> bpp(prob, prob + nk);
    const auto t0 = std::chrono::steady_clock::now();
    orc::StrVec km = orc::kmers_from_reads(reads, k);
    orc::StrVec contigs = orc::dbg_contigs(km, k, nullptr);
    orc::BreakScores R = orc::calc_breakscore(contigs, reads, std::string(), kmer, bpk, bpp, false, false);
    *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    Blob B;
    B.strs(1, contigs);
    B.vec(2, R.kmer_breaks);
    B.vec(3, R.bp_score);
    std::vector<uint64_t> n = {(uint64_t)km.size()};
    B.vec(4, n);
    return B.finish(nbytes);
}

// tags: 1 scaffolds.  matrix given as `rows` permutations (u32 indices) of `contigs`.
unsigned char* orc_assemble_matrix(const char* cd, const uint64_t* coff, uint64_t nc, const uint32_t* perm,
                                   uint64_t rows, int k, uint64_t* nbytes, int* err) {
    *err = 0;
    orc::StrVec c = unpack(cd, coff, nc);
    std::vector<orc::StrVec> M(rows);
    for (uint64_t r = 0; r < rows; ++r) { M[r].resize(nc); for (uint64_t j = 0; j < nc; ++j) M[r][j] = c[perm[r * nc + j]]; }
    Blob B;
    try { B.strs(1, orc::assemble_matrix(M, k)); } catch (const std::out_of_range&) { *err = 1; }
    return B.finish(nbytes);
}

unsigned char* orc_assemble_velvet(const char* cd, const uint64_t* coff, uint64_t nc, int k, int seed, int rows,
                                   uint64_t* nbytes, int* err) {
    *err = 0;
    Blob B;
    try { B.strs(1, orc::assemble_velvet(unpack(cd, coff, nc), k, seed, rows)); } catch (const std::out_of_range&) { *err = 1; }
    return B.finish(nbytes);
}

// tags: 1 sequence_len(i32) 2 bp_score 3 norm_by_break_freqs 4 norm_by_len 5 kmer_breaks(i32) 6 lev(i32)
//       7 path_freq (map order) 8 path_freq_by_input 9 startpos(i32) 10 prob_dist
unsigned char* orc_calc_breakscore(const char* pd, const uint64_t* poff, uint64_t np_, const char* rd,
                                   const uint64_t* roff, uint64_t nr, const char* truth, uint64_t truth_len, int kmer,
                                   const char* kd, const uint64_t* koff, uint64_t nk, const double* prob, int velvet,
                                   int with_lev, int with_freq, uint64_t* nbytes) {
    orc::StrVec bpk = unpack(kd, koff, nk);
    std::vector<double> bpp(prob, prob + nk);
    orc::BreakScores R = orc::calc_breakscore(unpack(pd, poff, np_), unpack(rd, roff, nr),
                                              std::string(truth, truth_len), kmer, bpk, bpp, velvet != 0, with_lev != 0);
    Blob B;
    B.vec(1, R.sequence_len); B.vec(2, R.bp_score); B.vec(3, R.norm_by_break_freqs); B.vec(4, R.norm_by_len);
    B.vec(5, R.kmer_breaks); B.vec(6, R.lev_dist);
    if (!velvet && with_freq) { B.vecs(7, R.path_freq); B.vecs(8, R.path_freq_by_input); }
    if (velvet) { B.vec(9, R.startpos); B.vecs(10, R.prob_dist); }
    return B.finish(nbytes);
}

int orc_levenshtein(const char* q, uint64_t nq, const char* t, uint64_t nt, int infix) {
    return orc::levenshtein(std::string(q, nq), std::string(t, nt), infix != 0);
}

void orc_count_windows(const char* rd, const uint64_t* roff, uint64_t nr, int k, const char* kd, const uint64_t* koff,
                       uint64_t nk, int64_t* out) {
    std::vector<int64_t> c = orc::count_windows(unpack(rd, roff, nr), k, unpack(kd, koff, nk));
    std::memcpy(out, c.data(), c.size() * 8);
}

void orc_normalise_tables(double* prob, const uint64_t* sizes, uint64_t ntables) {
    uint64_t tot = 0;
    std::vector<size_t> sz(ntables);
    for (uint64_t i = 0; i < ntables; ++i) { sz[i] = sizes[i]; tot += sizes[i]; }
    std::vector<double> v(prob, prob + tot);
    orc::normalise_tables(v, sz);
    std::memcpy(prob, v.data(), tot * 8);
}

// ---------------------------------------------------------------------------
// F4 — the R post-processing of score_solutions() (lib/DeNovoAssembler.R:318-479).
// ---------------------------------------------------------------------------
// lib/GenerateReads.R:243-259: kmer_from_seq = probability of the kmer-long window starting at every genome position
// (match() against the kmer-long table; a window absent from the table gives NA — not with ACGT input).
void orc_kmer_from_seq(const char* genome, uint64_t len, int kmer, const char* kd, const uint64_t* koff, uint64_t nk,
                       const double* prob, double* out /* len - kmer + 1 */) {
    std::unordered_map<std::string, double> t;
    orc::StrVec keys = unpack(kd, koff, nk);
    for (uint64_t i = 0; i < nk; ++i) if ((int)keys[i].size() == kmer) t.emplace(keys[i], prob[i]);
    const std::string g(genome, len);
    for (uint64_t p = 0; p + kmer <= len; ++p) {
        auto it = t.find(g.substr(p, kmer));
        out[p] = it == t.end() ? std::nan("") : it->second;
    }
}

// lib/DeNovoAssembler.R:414-424: NAs dropped from x, then stats::ks.test(x, y, "two.sided")$statistic, restated from
// R's C-free R code for the two-sample case (stats:::ks.test.default, R 4.x):
//     w <- c(x, y); z <- cumsum(ifelse(order(w) <= n.x, 1 / n.x, -1 / n.y))
//     if (length(unique(w)) < (n.x + n.y)) z <- z[c(which(diff(sort(w)) != 0), n.x + n.y)]
//     STATISTIC <- max(abs(z))
// (y's NAs are dropped as well; an empty x or y is an error in R: NaN here.)  The p-value is not restated: parity unpinned.
double orc_ks_statistic(const double* x, uint64_t nx0, const double* y, uint64_t ny0) {
    std::vector<std::pair<double, int>> w;
    uint64_t nx = 0, ny = 0;
    for (uint64_t i = 0; i < nx0; ++i) if (!std::isnan(x[i])) { w.emplace_back(x[i], 0); ++nx; }
    for (uint64_t i = 0; i < ny0; ++i) if (!std::isnan(y[i])) { w.emplace_back(y[i], 1); ++ny; }
    if (nx == 0 || ny == 0) return std::nan("");
    std::stable_sort(w.begin(), w.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) { return a.first < b.first; });
    const double ux = 1.0 / (double)nx, uy = -1.0 / (double)ny;
    double z = 0, best = 0;
    for (size_t i = 0; i < w.size(); ++i) {
        z += w[i].second == 0 ? ux : uy;
        if (i + 1 == w.size() || w[i + 1].first != w[i].first) best = std::max(best, std::fabs(z));   // last of a run of ties
    }
    return best;
}

// lib/DeNovoAssembler.R:432-445: percentage of [1, seq_len] covered by the union of the inclusive ranges
// [start_i, start_i + len_i] (IRanges: start/end inclusive; `end = path_freq_startpos + sequence_len`).
double orc_coverage_percent(const int64_t* start, const int64_t* len, uint64_t n, int64_t seq_len) {
    if (seq_len <= 0) return 0.0;
    std::vector<char> cov((size_t)seq_len + 2, 0);
    for (uint64_t i = 0; i < n; ++i) {
        const int64_t a = std::max<int64_t>(1, start[i]), b = std::min<int64_t>(seq_len, start[i] + len[i]);
        for (int64_t p = a; p <= b; ++p) cov[(size_t)p] = 1;
    }
    int64_t unc = 0;
    for (int64_t p = 1; p <= seq_len; ++p) unc += !cov[(size_t)p];
    return (1.0 - (double)unc / (double)seq_len) * 100.0;
}

// ---------------------------------------------------------------------------
// F3 — the read simulator (lib/GenerateReads.R:235-313), restated with the build's own pinned random stream (R's
// sample() cannot be reproduced without R): n = ceil(coverage * L / read_len) draws (:302) with replacement over the
// L - kmer + 1 start positions, weight = probability of the kmer-long window starting there (:243-259, :303-308; nullptr:
// equal weights), draws whose read would run past the end dropped (:310-313), kept starts in draw order.
// Integer procedure (the HIP path does the same arithmetic, kernels_sim.hip): weight = llrint(prob * 2^52), running sums in
// u64, draw d of segment s = mix64(mix64(seed ^ (0xD1B54A32D192ED03 * (s + 1))) + d) with splitmix64's output function,
// r = high 64 bits of draw * total, start = first position whose running sum exceeds r.
// ---------------------------------------------------------------------------
static inline uint64_t sim_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
uint64_t orc_simulate_starts(const char* genome, uint64_t L, uint32_t seg_index, uint32_t read_len, double coverage, uint64_t seed,
                             int kmer, const char* kd, const uint64_t* koff, uint64_t nk, const double* prob, int weight_shift, uint32_t* out_starts) {
    if (L < (uint64_t)kmer) return 0;
    const uint64_t np_ = L - kmer + 1;
    std::vector<uint64_t> cum(np_);
    std::unordered_map<std::string, double> t;
    if (prob) {
        orc::StrVec keys = unpack(kd, koff, nk);
        for (uint64_t i = 0; i < nk; ++i) if ((int)keys[i].size() == kmer) t.emplace(keys[i], prob[i]);
    }
    const std::string g(genome, L);
    uint64_t run = 0;
    for (uint64_t p = 0; p < np_; ++p) {
        uint64_t w = 1;
        if (prob) w = (uint64_t)std::llrint(std::ldexp(t.at(g.substr(p, kmer)), weight_shift));
        run += w;
        cum[p] = run;
    }
    const uint64_t nd = (uint64_t)std::ceil(coverage * (double)L / (double)read_len);
    const uint64_t sseed = sim_mix64(seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(seg_index + 1)));
    uint64_t kept = 0;
    if (run == 0) return 0;
    for (uint64_t d = 0; d < nd; ++d) {
        const uint64_t x = sim_mix64(sseed + d);
        const uint64_t r = (uint64_t)(((unsigned __int128)x * run) >> 64);
        const uint64_t p = (uint64_t)(std::upper_bound(cum.begin(), cum.end(), r) - cum.begin());
        if (p + read_len <= L) out_starts[kept++] = (uint32_t)p;
    }
    return kept;
}

}  // extern "C"
