// san_driver.cpp — host algorithms of libgasm (host_algos.cpp, seqio.cpp) and the oracle, compiled together with
// -fsanitize=address,undefined by tests/test_sanitizers.py and run on randomised inputs: shuffle matrix, greedy merge in
// both forms (index form and string form, incl. the substr-out-of-range case), signatures, Myers edit distance in both
// modes against the oracle's DP, the sequence-file reader.  Exit code 0 = every comparison held and no sanitizer report.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "gasm_internal.h"

static thread_local std::string g_err;
void gasm_set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

extern "C" {
unsigned char* orc_assemble_matrix(const char* cd, const uint64_t* coff, uint64_t nc, const uint32_t* perm, uint64_t rows, int k, uint64_t* nbytes, int* err);
int orc_levenshtein(const char* q, uint64_t nq, const char* t, uint64_t nt, int infix);
void orc_free(void* p);
}

static std::string rnd(std::mt19937& g, size_t n) {
    std::string s(n, 'A');
    for (auto& c : s) c = "ACGT"[g() & 3];
    return s;
}

static std::vector<std::string> blob_strs(const unsigned char* raw) {
    // Blob layout (oracle): sections of {tag u64, n u64, payload padded to 8}; tag 1 = "<count>\n" + strings joined by '\n'
    uint64_t tag, n;
    memcpy(&tag, raw, 8); memcpy(&n, raw + 8, 8);
    std::string s(reinterpret_cast<const char*>(raw + 16), n);
    const size_t nl = s.find('\n');
    const size_t cnt = std::stoul(s.substr(0, nl));
    std::vector<std::string> v;
    if (!cnt) return v;
    size_t p = nl + 1;
    for (;;) {
        const size_t q = s.find('\n', p);
        if (q == std::string::npos) { v.push_back(s.substr(p)); break; }
        v.push_back(s.substr(p, q - p));
        p = q + 1;
    }
    return v;
}

int main(int argc, char** argv) {
    std::mt19937 g(12345);
    int checks = 0;
    for (int trial = 0; trial < 60; ++trial) {
        const int k = 3 + (int)(g() % 10);
        const std::string genome = rnd(g, 80 + g() % 400);
        std::vector<std::string> contigs;
        for (size_t a = 0; a + k < genome.size();) {
            const size_t len = k - 1 + g() % 40;
            contigs.push_back(genome.substr(a, std::min(len, genome.size() - a)));
            a += std::max<size_t>(1, len - (k - 1));
        }
        if (trial % 7 == 3) contigs.push_back(rnd(g, 1 + g() % (k - 1)));      // shorter than k-1: string form / range error
        std::sort(contigs.begin(), contigs.end());
        contigs.erase(std::unique(contigs.begin(), contigs.end()), contigs.end());
        const uint64_t n = contigs.size(), rows = 1 + g() % 120;
        std::vector<u32> perm;
        gasm_host::shuffle_perm(n, 7 + trial, rows, perm);
        std::vector<std::string> out;
        const int st = gasm_host::assemble(contigs, perm.data(), rows, n, k, out);
        std::string cat;
        std::vector<uint64_t> off(1, 0);
        for (auto& c : contigs) { cat += c; off.push_back(cat.size()); }
        uint64_t nb = 0;
        int err = 0;
        unsigned char* raw = orc_assemble_matrix(cat.data(), off.data(), n, perm.data(), rows, k, &nb, &err);
        if (err) {
            if (st != GASM_ERR_RANGE) { fprintf(stderr, "trial %d: oracle throws, host returns %d\n", trial, st); return 1; }
        } else {
            if (st != GASM_OK || out != blob_strs(raw)) { fprintf(stderr, "trial %d: scaffolds differ\n", trial); return 1; }
        }
        orc_free(raw);
        std::vector<std::string> sigs;
        (void)gasm_host::assemble_signatures(contigs, perm.data(), rows, n, k, sigs);
        ++checks;
    }
    for (int trial = 0; trial < 200; ++trial) {
        const std::string t = rnd(g, g() % 300), q = trial % 3 ? rnd(g, g() % 200) : t.substr(g() % (t.size() + 1));
        for (int infix = 0; infix < 2; ++infix) {
            const int a = gasm_host::levenshtein(q.data(), q.size(), t.data(), t.size(), infix != 0);
            const int b = orc_levenshtein(q.data(), q.size(), t.data(), t.size(), infix);
            if (a != b) { fprintf(stderr, "levenshtein(%zu, %zu, infix=%d): %d vs %d\n", q.size(), t.size(), infix, a, b); return 1; }
            ++checks;
        }
    }
    for (int i = 1; i < argc; ++i) {
        gasm_host::PackedReads pr;
        u64 kept = 0, dropped = 0;
        const int st = gasm_host::read_sequence_file(argv[i], false, pr, &kept, &dropped);
        printf("%s: status %d, %llu reads kept, %llu dropped, %llu bases\n", argv[i], st, (unsigned long long)kept, (unsigned long long)dropped,
               (unsigned long long)pr.total_bases);
        ++checks;
    }
    printf("sanitizer driver: %d checks ok\n", checks);
    return 0;
}
