// DeNovoAssemblerHIP.cpp — Rcpp glue: the file a maintainer of SahakyanLab/GenomeAssembler_dev would
// `Rcpp::sourceCpp()` INSTEAD OF lib/DeNovoAssembler.cpp (scripts/02_Real_vs_rand_prob_own.R:19).  It exports the same
// three R functions with the same argument names and return shapes and forwards to libgasm's C ABI (include/gasm.h).
//
// NOT COMPILED IN THIS PIPELINE: R and Rcpp are not installed in the build image or on the GPU box.  The same C-ABI
// calls are exercised by genomeassembler_dev_amd/api.py (ctypes) in tests/.
//
// Build (where R exists):   Sys.setenv(PKG_CPPFLAGS = "-I<repo>/include",
//                                      PKG_LIBS = "-L<repo>/genomeassembler_dev_amd -lgasm -Wl,-rpath,<repo>/genomeassembler_dev_amd")
//                           Rcpp::sourceCpp("integration/DeNovoAssemblerHIP.cpp")
// [[Rcpp::plugins("cpp17")]]
#include <Rcpp.h>

#include <algorithm>
#include <string>
#include <vector>

#include "gasm.h"

namespace {

gasm_ctx* the_ctx() {
    static gasm_ctx* ctx = nullptr;
    if (!ctx && gasm_ctx_create(0, &ctx) != GASM_OK) Rcpp::stop(gasm_last_error());
    return ctx;
}

void check(int status) {
    if (status != GASM_OK) Rcpp::stop(gasm_last_error());  // the reference raises R errors through BEGIN_RCPP/END_RCPP
}

struct Flat {
    std::string data;
    std::vector<uint64_t> off;
    explicit Flat(const std::vector<std::string>& v) : off(v.size() + 1, 0) {
        for (size_t i = 0; i < v.size(); ++i) { data += v[i]; off[i + 1] = data.size(); }
    }
};

std::vector<std::string> unflat(const char* d, const uint64_t* off, uint64_t n) {
    std::vector<std::string> v(n);
    for (uint64_t i = 0; i < n; ++i) v[i].assign(d + off[i], d + off[i + 1]);
    return v;
}

}  // namespace

// replaces lib/DeNovoAssembler.cpp:86-206
// [[Rcpp::export]]
std::vector<std::vector<std::string>> get_contigs(const std::vector<std::string>& read_kmers, const int& dbg_kmer,
                                                  const int& seed) {
    for (const std::string& s : read_kmers)
        if ((int)s.size() != dbg_kmer) Rcpp::stop("get_contigs: every read k-mer must be dbg_kmer characters long");
    Flat f(read_kmers);
    gasm_contigs* c = nullptr;
    check(gasm_get_contigs(the_ctx(), f.data.data(), read_kmers.size(), dbg_kmer, seed, 10000, &c));
    const uint64_t n = gasm_contigs_count(c), rows = gasm_contigs_rows(c);
    std::vector<std::string> contigs = unflat(gasm_contigs_data(c), gasm_contigs_offsets(c), n);
    const uint32_t* perm = gasm_contigs_perm(c);
    std::vector<std::vector<std::string>> m(rows, std::vector<std::string>(n));
    for (uint64_t r = 0; r < rows; ++r)
        for (uint64_t j = 0; j < n; ++j) m[r][j] = contigs[perm[r * n + j]];
    gasm_contigs_free(c);
    return m;
}

// get_kmers_from_reads (lib/DeNovoAssembler.R:109-130) + get_contigs in one call: the reads go down as they are and the
// k-mers are taken on the GPU (self$read_kmers need not exist; in DeNovoAssembler.R: get_contigs_from_reads(
// self$sequencing_reads$read_one, self$dbg_kmer, self$seed) in place of the two steps)
// [[Rcpp::export]]
std::vector<std::vector<std::string>> get_contigs_from_reads(const std::vector<std::string>& reads, const int& dbg_kmer, const int& seed) {
    Flat f(reads);
    gasm_contigs* c = nullptr;
    check(gasm_get_contigs_from_reads(the_ctx(), f.data.data(), f.off.data(), reads.size(), dbg_kmer, seed, 10000, &c));
    const uint64_t n = gasm_contigs_count(c), rows = gasm_contigs_rows(c);
    std::vector<std::string> contigs = unflat(gasm_contigs_data(c), gasm_contigs_offsets(c), n);
    const uint32_t* perm = gasm_contigs_perm(c);
    std::vector<std::vector<std::string>> m(rows, std::vector<std::string>(n));
    for (uint64_t r = 0; r < rows; ++r)
        for (uint64_t j = 0; j < n; ++j) m[r][j] = contigs[perm[r * n + j]];
    gasm_contigs_free(c);
    return m;
}

// replaces lib/DeNovoAssembler.cpp:215-305
// [[Rcpp::export]]
std::vector<std::string> assemble_contigs(const std::vector<std::vector<std::string>>& contig_matrix, const int& dbg_kmer) {
    // distinct strings + index matrix
    std::vector<std::string> uniq;
    for (const auto& row : contig_matrix) uniq.insert(uniq.end(), row.begin(), row.end());
    std::sort(uniq.begin(), uniq.end());
    uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
    const uint64_t rows = contig_matrix.size(), width = rows ? contig_matrix[0].size() : 0;
    std::vector<uint32_t> perm(rows * width);
    for (uint64_t r = 0; r < rows; ++r) {
        if (contig_matrix[r].size() != width) Rcpp::stop("assemble_contigs: rows of unequal length");
        for (uint64_t j = 0; j < width; ++j)
            perm[r * width + j] = (uint32_t)(std::lower_bound(uniq.begin(), uniq.end(), contig_matrix[r][j]) - uniq.begin());
    }
    Flat f(uniq);
    gasm_strlist* s = nullptr;
    check(gasm_assemble_contigs(the_ctx(), f.data.data(), f.off.data(), uniq.size(), perm.data(), rows, width, dbg_kmer, &s));
    std::vector<std::string> out = unflat(gasm_strlist_data(s), gasm_strlist_offsets(s), gasm_strlist_count(s));
    gasm_strlist_free(s);
    return out;
}

// replaces lib/DeNovoAssembler.cpp:317-477
// [[Rcpp::export]]
Rcpp::List calc_breakscore(const std::vector<std::string>& path, const std::vector<std::string>& sequencing_reads,
                           const std::string& true_solution, const int& kmer, const std::vector<std::string>& bp_kmer,
                           const std::vector<double>& bp_prob) {
    Flat p(path), r(sequencing_reads), t(bp_kmer);
    gasm_scores* s = nullptr;
    check(gasm_calc_breakscore(the_ctx(), p.data.data(), p.off.data(), path.size(), r.data.data(), r.off.data(),
                               sequencing_reads.size(), true_solution.data(), true_solution.size(), kmer, t.data.data(),
                               t.off.data(), bp_kmer.size(), bp_prob.data(), GASM_SCORE_OWN, GASM_WANT_LEV | GASM_WANT_FREQ, &s));
    const uint64_t n = gasm_scores_count(s), nt = bp_kmer.size();
    auto ivec = [&](const int32_t* a) { return std::vector<int>(a, a + n); };
    auto dvec = [&](const double* a) { return std::vector<double>(a, a + n); };
    std::vector<std::vector<double>> freq(n);
    const double* fq = gasm_scores_path_freq(s);
    for (uint64_t i = 0; i < n; ++i) freq[i].assign(fq + i * nt, fq + (i + 1) * nt);
    Rcpp::List out = Rcpp::List::create(
        Rcpp::Named("sequence") = path, Rcpp::Named("sequence_len") = ivec(gasm_scores_sequence_len(s)),
        Rcpp::Named("bp_score") = dvec(gasm_scores_bp_score(s)),
        Rcpp::Named("bp_score_norm_by_break_freqs") = dvec(gasm_scores_norm_by_break_freqs(s)),
        Rcpp::Named("bp_score_norm_by_len") = dvec(gasm_scores_norm_by_len(s)),
        Rcpp::Named("kmer_breaks") = ivec(gasm_scores_kmer_breaks(s)),
        Rcpp::Named("lev_dist_vs_true") = ivec(gasm_scores_lev_dist(s)), Rcpp::Named("path_freq") = Rcpp::wrap(freq));
    gasm_scores_free(s);
    return out;
}
