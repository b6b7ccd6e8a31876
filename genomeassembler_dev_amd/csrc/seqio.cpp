// seqio.cpp — FASTQ / FASTA in (the north star's "FASTQ-in" surface).  The reference has no file
// reader on this path — it simulates reads in R and writes FASTA (lib/GenerateReads.R:405-433: one record per read,
// header "<chr>_<start>_<end>:0_<i>/1", sequence on one line); README.md:29-31 names kseq.h but nothing includes it.
// Accepted here: FASTQ with four-line records (what sequencers and the usual simulators write), FASTA with sequences on
// one or several lines, plain or gzip (zlib's gzread reads both), lower case folded to upper case.  A read holding a
// byte outside ACGT (N, IUPAC codes) cannot be packed in 2 bits: it is dropped and counted, or refused.
// The sequences are packed 2-bit on the fly (first base most significant, 32 bases per word, reads back to back): what
// goes over PCIe is a quarter of the text.
#include <zlib.h>

#include "gasm_internal.h"

namespace gasm_host {

struct LineReader {
    gzFile f = nullptr;
    std::vector<char> buf;
    size_t pos = 0, end = 0;
    bool failed = false;          // a read error or a truncated / corrupt gzip stream: the reads so far are NOT the file's reads
    std::string fail_text;
    bool open(const char* path) {
        f = gzopen(path, "rb");
        if (!f) return false;
        gzbuffer(f, 1 << 20);
        buf.resize(1 << 20);
        return true;
    }
    ~LineReader() { if (f) gzclose(f); }
    // next line without its terminator ("\n" or "\r\n"); false at the end of the file
    bool line(std::string& out) {
        out.clear();
        bool any = false;
        for (;;) {
            if (pos == end) {
                const int n = gzread(f, buf.data(), (unsigned)buf.size());
                if (n < 0 || (n == 0 && !gzeof(f))) {       // (n == 0 at a clean end of file; anything else is an error)
                    int code = 0;
                    const char* msg = gzerror(f, &code);
                    failed = true;
                    fail_text = msg && *msg ? msg : "read error";
                    break;
                }
                if (n == 0) {
                    // zlib reports a gzip stream that stops short of its trailer through gzerror (Z_BUF_ERROR), not through gzread
                    int code = 0;
                    const char* msg = gzerror(f, &code);
                    if (code != Z_OK && code != Z_STREAM_END) { failed = true; fail_text = msg && *msg ? msg : "truncated stream"; }
                    break;
                }
                pos = 0; end = (size_t)n;
            }
            any = true;
            const char* b = buf.data() + pos;
            const char* nl = static_cast<const char*>(memchr(b, '\n', end - pos));
            if (nl) {
                out.append(b, (size_t)(nl - b));
                pos = (size_t)(nl - buf.data()) + 1;
                if (!out.empty() && out.back() == '\r') out.pop_back();
                return true;
            }
            out.append(b, end - pos);
            pos = end;
        }
        if (any && !out.empty() && out.back() == '\r') out.pop_back();
        return any;
    }
};

static inline int base_code_host(unsigned char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return -1;
    }
}

void PackedReads::append(const std::string& seq) {
    for (unsigned char c : seq) {
        const u64 p = total_bases++;
        if ((p & 31) == 0) words.push_back(0);
        words.back() |= (u64)base_code_host(c) << (62 - 2 * (p & 31));
    }
    read_off.push_back(total_bases);
}

// one file -> its reads appended to `out`.  Returns GASM_OK / GASM_ERR_INVALID (unreadable, malformed) / GASM_ERR_NON_ACGT.
int read_sequence_file(const char* path, bool error_on_non_acgt, PackedReads& out, u64* n_kept, u64* n_dropped) {
    LineReader lr;
    if (!lr.open(path)) { gasm_set_error("cannot open %s", path); return GASM_ERR_INVALID; }
    *n_kept = 0;
    if (out.read_off.empty()) out.read_off.push_back(0);
    std::string ln, seq;
    auto take = [&](const std::string& s) -> int {
        for (unsigned char c : s)
            if (base_code_host(c) < 0) {
                if (error_on_non_acgt) { gasm_set_error("%s: a read holds a base outside ACGT", path); return GASM_ERR_NON_ACGT; }
                ++*n_dropped;
                return GASM_OK;
            }
        out.append(s);
        ++*n_kept;
        return GASM_OK;
    };
    auto eof_status = [&]() -> int {
        if (lr.failed) { gasm_set_error("%s: %s (truncated or corrupt file)", path, lr.fail_text.c_str()); return GASM_ERR_INVALID; }
        return GASM_OK;
    };
    if (!lr.line(ln)) return eof_status();                  // empty file: no reads
    while (ln.empty()) if (!lr.line(ln)) return eof_status();
    if (ln[0] == '@') {
        u64 rec = 0;
        for (;;) {
            if (ln.empty()) { if (!lr.line(ln)) break; continue; }           // blank lines between records
            if (ln[0] != '@') { gasm_set_error("%s: record %llu does not start with '@'", path, (unsigned long long)rec); return GASM_ERR_INVALID; }
            std::string plus, qual;
            if (!lr.line(seq) || !lr.line(plus) || plus.empty() || plus[0] != '+') {
                gasm_set_error("%s: record %llu has no '+' line (multi-line FASTQ is not supported)", path, (unsigned long long)rec);
                return GASM_ERR_INVALID;
            }
            lr.line(qual);
            while (!seq.empty() && (seq.back() == ' ' || seq.back() == '\t')) seq.pop_back();
            GCHK(take(seq));
            ++rec;
            if (!lr.line(ln)) break;
        }
    } else if (ln[0] == '>') {
        bool have = false;
        seq.clear();
        for (;;) {
            if (!ln.empty() && ln[0] == '>') {
                if (have) GCHK(take(seq));
                have = true;
                seq.clear();
            } else if (have) {
                size_t a = 0, b = ln.size();
                while (a < b && (ln[a] == ' ' || ln[a] == '\t')) ++a;
                while (b > a && (ln[b - 1] == ' ' || ln[b - 1] == '\t')) --b;
                seq.append(ln, a, b - a);
            }
            if (!lr.line(ln)) break;
        }
        if (have) GCHK(take(seq));
    } else {
        gasm_set_error("%s: neither FASTQ ('@') nor FASTA ('>')", path);
        return GASM_ERR_INVALID;
    }
    if (lr.failed) { gasm_set_error("%s: %s (truncated or corrupt file: its reads are not used)", path, lr.fail_text.c_str()); return GASM_ERR_INVALID; }
    return GASM_OK;
}

}  // namespace gasm_host
