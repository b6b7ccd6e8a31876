// BreakageScorerHIP.cpp — Rcpp glue to source INSTEAD OF lib/BreakageScorer.cpp (scripts/00_Real_vs_rand_prob_velvet.R:18).
// Same caveats and build line as DeNovoAssemblerHIP.cpp: not compiled in this pipeline (no R here).
// [[Rcpp::plugins("cpp17")]]
#include <Rcpp.h>

#include <string>
#include <vector>

#include "gasm.h"

namespace {
gasm_ctx* the_ctx() {
    static gasm_ctx* ctx = nullptr;
    if (!ctx && gasm_ctx_create(0, &ctx) != GASM_OK) Rcpp::stop(gasm_last_error());
    return ctx;
}
void check(int status) { if (status != GASM_OK) Rcpp::stop(gasm_last_error()); }
struct Flat {
    std::string data;
    std::vector<uint64_t> off;
    explicit Flat(const std::vector<std::string>& v) : off(v.size() + 1, 0) {
        for (size_t i = 0; i < v.size(); ++i) { data += v[i]; off[i + 1] = data.size(); }
    }
};
}  // namespace

// replaces lib/BreakageScorer.cpp:80-174
// [[Rcpp::export]]
std::vector<std::string> assemble_contigs(const std::vector<std::string>& velvet_contigs, const int& dbg_kmer, const int& seed) {
    Flat f(velvet_contigs);
    gasm_strlist* s = nullptr;
    check(gasm_assemble_contigs_velvet(the_ctx(), f.data.data(), f.off.data(), velvet_contigs.size(), dbg_kmer, seed, 20000, &s));
    const uint64_t n = gasm_strlist_count(s);
    const char* d = gasm_strlist_data(s);
    const uint64_t* off = gasm_strlist_offsets(s);
    std::vector<std::string> out(n);
    for (uint64_t i = 0; i < n; ++i) out[i].assign(d + off[i], d + off[i + 1]);
    gasm_strlist_free(s);
    return out;
}

// replaces lib/BreakageScorer.cpp:186-353
// [[Rcpp::export]]
Rcpp::List calc_breakscore(const std::vector<std::string>& path, const std::vector<std::string>& sequencing_reads,
                           const std::string& true_solution, const int& kmer, const std::vector<std::string>& bp_kmer,
                           const std::vector<double>& bp_prob) {
    Flat p(path), r(sequencing_reads), t(bp_kmer);
    gasm_scores* s = nullptr;
    check(gasm_calc_breakscore(the_ctx(), p.data.data(), p.off.data(), path.size(), r.data.data(), r.off.data(),
                               sequencing_reads.size(), true_solution.data(), true_solution.size(), kmer, t.data.data(),
                               t.off.data(), bp_kmer.size(), bp_prob.data(), GASM_SCORE_VELVET, GASM_WANT_LEV, &s));
    const uint64_t n = gasm_scores_count(s);
    auto ivec = [&](const int32_t* a) { return std::vector<int>(a, a + n); };
    auto dvec = [&](const double* a) { return std::vector<double>(a, a + n); };
    const uint64_t* po = gasm_scores_prob_dist_offsets(s);
    const double* pd = gasm_scores_prob_dist(s);
    std::vector<std::vector<double>> dist(n);
    for (uint64_t i = 0; i < n; ++i) dist[i].assign(pd + po[i], pd + po[i + 1]);
    Rcpp::List out = Rcpp::List::create(
        Rcpp::Named("sequence") = path, Rcpp::Named("sequence_len") = ivec(gasm_scores_sequence_len(s)),
        Rcpp::Named("bp_score") = dvec(gasm_scores_bp_score(s)),
        Rcpp::Named("bp_score_norm_by_break_freqs") = dvec(gasm_scores_norm_by_break_freqs(s)),
        Rcpp::Named("bp_score_norm_by_len") = dvec(gasm_scores_norm_by_len(s)),
        Rcpp::Named("kmer_breaks") = ivec(gasm_scores_kmer_breaks(s)),
        Rcpp::Named("lev_dist_vs_true") = ivec(gasm_scores_lev_dist(s)),
        Rcpp::Named("path_prob_dist_startpos") = ivec(gasm_scores_startpos(s)),
        Rcpp::Named("path_prob_dist") = Rcpp::wrap(dist));
    gasm_scores_free(s);
    return out;
}
