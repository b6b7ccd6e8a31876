// ingest.hip — sequence files parsed and packed ON THE DEVICE (the "on GPU" half; the reference has no
// reader on this path: it writes FASTA itself, lib/GenerateReads.R:405-433, header "<chr>_<start>_<end>:0_<i>/1", sequence on
// one line).  The host keeps what only it can do — open the file, inflate gzip (zlib) — and hands the text to the GPU:
//   k_ing_count_nl / k_ing_line_starts   newlines counted per 4 KB chunk, scanned, every line's first byte written
//   k_ing_lines                          per line: trimmed span (CR, blanks), first byte, "holds a byte outside ACGT"
//   k_ing_fastq_check                    four-line records?  ('@' on lines 4r, '+' on lines 4r + 2, blank lines only at the end)
//   k_ing_fasta_records                  '>' lines numbered (scan): record of every sequence line, a record's bad flag
//   k_ing_piece_len / scans / k_ing_read_off   kept records, their base offsets: read_off
//   k_ing_pack                           a thread per output word gathers its 32 bases across line pieces, 2-bit, case folded
//   k_ing_append                         a file's reads appended to the batch's stream at any base offset
// The grammar is the host reader's (seqio.cpp), which stays the reference for anything irregular: a text the device path
// does not recognise as plain four-line FASTQ or as FASTA (blank lines between FASTQ records, a missing '+' line, neither
// format) goes to read_sequence_file, which also words the errors.  gasm_packed_parsed_on_device tells which one read a file.
#include <zlib.h>

#include <algorithm>

#include "pipeline.h"

namespace {

constexpr u32 CH = 4096;          // bytes per chunk of the newline count
constexpr u32 SB = 8192;          // elements per block of the device-wide scans

__device__ __forceinline__ bool is_blank(u8 c) { return c == ' ' || c == '\t'; }
__device__ __forceinline__ int code_of(u8 c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return -1;
    }
}

__global__ void __launch_bounds__(256) k_ing_count_nl(const u8* __restrict__ text, u64 n, u32* __restrict__ chunk_cnt) {
    __shared__ u32 s_tmp[8];
    const u64 base = (u64)blockIdx.x * CH + threadIdx.x * 16;
    u32 c = 0;
    for (u32 q = 0; q < 16; ++q) if (base + q < n && text[base + q] == '\n') ++c;
    u32 tot;
    (void)block_excl_scan<256>(c, s_tmp, &tot);
    if (threadIdx.x == 0) chunk_cnt[blockIdx.x] = tot;
}

// line_start[j + 1] = byte after the j-th newline (line_start[0] = 0 is the host's)
__global__ void __launch_bounds__(256) k_ing_line_starts(const u8* __restrict__ text, u64 n, const u64* __restrict__ chunk_base, u64* __restrict__ line_start) {
    __shared__ u32 s_tmp[8];
    const u64 base = (u64)blockIdx.x * CH + threadIdx.x * 16;
    u32 c = 0;
    for (u32 q = 0; q < 16; ++q) if (base + q < n && text[base + q] == '\n') ++c;
    u32 tot;
    u32 r = (u32)chunk_base[blockIdx.x] + block_excl_scan<256>(c, s_tmp, &tot);
    for (u32 q = 0; q < 16; ++q) if (base + q < n && text[base + q] == '\n') line_start[++r] = base + q + 1;
}

// Per line i (bytes [ls[i], ls[i + 1] - 1), the newline excluded): a trailing CR goes, then blanks at both ends (`both`) or at the
// end only (FASTQ sequence lines: a leading blank is a byte outside ACGT there, as in seqio.cpp).  first = first byte of the
// line without its CR (0 for an empty one); bad = a byte outside ACGT / acgt inside the trimmed span.
__global__ void __launch_bounds__(256) k_ing_lines(const u8* __restrict__ text, const u64* __restrict__ ls, u32 n_lines, int both, u64* __restrict__ l_a,
                                                   u32* __restrict__ l_len, u8* __restrict__ l_first, u8* __restrict__ l_bad) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_lines) return;
    u64 a = ls[i], b = ls[i + 1] - 1;
    if (b > a && text[b - 1] == '\r') --b;
    l_first[i] = b > a ? text[a] : 0;              // 0: an empty line (what the host reader calls blank)
    while (b > a && is_blank(text[b - 1])) --b;
    if (both) while (a < b && is_blank(text[a])) ++a;
    bool bad = false;
    for (u64 p = a; p < b; ++p) bad = bad || code_of(text[p]) < 0;
    l_a[i] = a;
    l_len[i] = (u32)(b - a);
    l_bad[i] = bad ? 1 : 0;
}

// info[0] = index of the last line that is not blank + 1 (atomicMax), info[1] |= 1 when a line breaks the four-line pattern
__global__ void __launch_bounds__(256) k_ing_last_nonblank(const u32* __restrict__ l_len, const u8* __restrict__ l_first, u32 n_lines, u32* __restrict__ info) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_lines && l_first[i] != 0) atomicMax(&info[0], i + 1);
}
__global__ void __launch_bounds__(256) k_ing_fastq_check(const u32* __restrict__ l_len, const u8* __restrict__ l_first, u32 n_eff, u32* __restrict__ info) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_eff) return;
    const u32 m = i & 3u;
    bool ok = true;
    if (m == 0) ok = l_first[i] == '@';
    else if (m == 2) ok = l_first[i] == '+';
    if (!ok) atomicOr(&info[1], 1u);
}

// ---- device-wide exclusive scan of u32 -> u64: block sums, their scan (one workgroup), block scans
__global__ void __launch_bounds__(1024) k_ing_block_sum(const u32* __restrict__ in, u64 n, u64* __restrict__ bsum) {
    __shared__ u64 s_w[16];
    const u64 i0 = (u64)blockIdx.x * SB + threadIdx.x * 8;
    u64 sum = 0;
    for (u32 q = 0; q < 8; ++q) if (i0 + q < n) sum += in[i0 + q];
    for (int d = 32; d; d >>= 1) sum += __shfl_down(sum, d, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) { u64 t = 0; for (u32 w = 0; w < 16; ++w) t += s_w[w]; bsum[blockIdx.x] = t; }
}
__global__ void __launch_bounds__(1024) k_ing_scan_sums(u64* __restrict__ bsum, u32 nb, u64* __restrict__ total) {
    __shared__ u64 s_w[16];
    const u32 ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
    u64 carry = 0;
    for (u32 base = 0; base < nb; base += 1024) {
        const u32 i = base + threadIdx.x;
        const u64 v = i < nb ? bsum[i] : 0;
        u64 inc = v;
        for (int d = 1; d < 64; d <<= 1) { const u64 o = __shfl_up(inc, d, 64); if ((int)ln >= d) inc += o; }
        if (ln == 63) s_w[wv] = inc;
        __syncthreads();
        u64 before = 0, tot = 0;
        for (u32 w = 0; w < 16; ++w) { const u64 t = s_w[w]; if (w < wv) before += t; tot += t; }
        __syncthreads();
        if (i < nb) bsum[i] = carry + before + inc - v;
        carry += tot;
    }
    if (threadIdx.x == 0) *total = carry;
}
__global__ void __launch_bounds__(1024) k_ing_block_scan(const u32* __restrict__ in, u64 n, const u64* __restrict__ bsum, u64* __restrict__ out) {
    __shared__ u64 s_w[16];
    const u32 ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u64 i0 = (u64)blockIdx.x * SB + threadIdx.x * 8;
    u32 v[8];
    u64 sum = 0;
    for (u32 q = 0; q < 8; ++q) { v[q] = i0 + q < n ? in[i0 + q] : 0u; sum += v[q]; }
    u64 inc = sum;
    for (int d = 1; d < 64; d <<= 1) { const u64 o = __shfl_up(inc, d, 64); if ((int)ln >= d) inc += o; }
    if (ln == 63) s_w[wv] = inc;
    __syncthreads();
    u64 before = 0;
    for (u32 w = 0; w < wv; ++w) before += s_w[w];
    u64 ex = bsum[blockIdx.x] + before + inc - sum;
    for (u32 q = 0; q < 8; ++q) { if (i0 + q < n) out[i0 + q] = ex; ex += v[q]; }
}

// FASTA: hdr[i] = 1 on '>' lines
__global__ void __launch_bounds__(256) k_ing_hdr_flags(const u8* __restrict__ l_first, u32 n_lines, u32* __restrict__ hdr) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_lines) hdr[i] = l_first[i] == '>' ? 1u : 0u;
}
// record of line i = (headers before or at i) - 1; sequence lines raise their record's bad flag; header lines note where
// their record starts (rec_line)
__global__ void __launch_bounds__(256) k_ing_fasta_records(const u32* __restrict__ hdr, const u64* __restrict__ hdr_ex, const u8* __restrict__ l_bad, u32 n_lines,
                                                           u32* __restrict__ rec_bad, u32* __restrict__ rec_line) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_lines) return;
    if (hdr[i]) { rec_line[hdr_ex[i]] = i; return; }
    if (hdr_ex[i] == 0) return;                         // before the first header: ignored
    if (l_bad[i]) atomicOr(&rec_bad[hdr_ex[i] - 1], 1u);
}
__global__ void __launch_bounds__(256) k_ing_fastq_records(const u8* __restrict__ l_bad, u32 n_rec, u32* __restrict__ rec_bad, u32* __restrict__ rec_line) {
    const u32 r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rec) return;
    rec_bad[r] = l_bad[4 * r + 1];
    rec_line[r] = 4 * r;
}
// bases line i contributes: its trimmed length if it is a sequence line of a kept record
__global__ void __launch_bounds__(256) k_ing_piece_len(int fastq, const u32* __restrict__ hdr, const u64* __restrict__ hdr_ex, const u32* __restrict__ l_len,
                                                       const u32* __restrict__ rec_bad, u32 n_lines, u32 n_eff, u32* __restrict__ plen, u32* __restrict__ keep, u32 n_rec) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_rec) keep[i] = rec_bad[i] ? 0u : 1u;
    if (i >= n_lines) return;
    u32 v = 0;
    if (fastq) { if (i < n_eff && (i & 3u) == 1u && !rec_bad[i >> 2]) v = l_len[i]; }
    else if (!hdr[i] && hdr_ex[i] > 0 && !rec_bad[hdr_ex[i] - 1]) v = l_len[i];
    plen[i] = v;
}
__global__ void __launch_bounds__(256) k_ing_read_off(const u32* __restrict__ keep, const u64* __restrict__ keep_ex, const u32* __restrict__ rec_line,
                                                      const u64* __restrict__ pbase, u32 n_rec, const u64* __restrict__ totals, u64* __restrict__ read_off) {
    const u32 r = blockIdx.x * 256 + threadIdx.x;
    if (r < n_rec && keep[r]) read_off[keep_ex[r]] = pbase[rec_line[r]];
    if (r == 0) read_off[totals[1]] = totals[0];        // totals: [0] bases, [1] kept records
}
// word w of the file's packed stream: bases 32 w .. 32 w + 31, gathered across the line pieces (pbase ascending, many zero-length)
__global__ void __launch_bounds__(256) k_ing_pack(const u8* __restrict__ text, const u64* __restrict__ l_a, const u32* __restrict__ plen, const u64* __restrict__ pbase,
                                                  u32 n_lines, const u64* __restrict__ totals, u64* __restrict__ words, u64 n_words_alloc) {
    const u64 total = totals[0], nw = (total + 31) / 32;
    for (u64 w = (u64)blockIdx.x * 256 + threadIdx.x; w < n_words_alloc; w += (u64)gridDim.x * 256) {
        u64 v = 0;
        if (w < nw) {
            const u64 b0 = w * 32;
            u32 lo = 0, hi = n_lines;               // last line with pbase <= b0
            while (hi - lo > 1) { const u32 m = (lo + hi) >> 1; if (pbase[m] <= b0) lo = m; else hi = m; }
            u32 i = lo;
            u64 off = b0 - pbase[i];
            for (u32 j = 0; j < 32 && b0 + j < total; ++j) {
                while (off >= plen[i]) { off -= plen[i]; ++i; }       // (b0 + j < total: a piece with room exists)
                v |= (u64)code_of(text[l_a[i] + off]) << (62 - 2 * j);
                ++off;
            }
        }
        words[w] = v;
    }
}
// dst stream |= the n_bases bases of src placed at base dst_off (src has padding words; dst is zero beyond what was appended)
__global__ void __launch_bounds__(256) k_ing_append(const u64* __restrict__ src, const u64* __restrict__ totals, u64* __restrict__ dst, u64 dst_off) {
    const u64 n_bases = totals[0];
    const u64 w0 = dst_off / 32, w1 = (dst_off + n_bases + 31) / 32;
    for (u64 w = w0 + (u64)blockIdx.x * 256 + threadIdx.x; w < w1; w += (u64)gridDim.x * 256) {
        const long long s = (long long)(w * 32) - (long long)dst_off;        // first source base of this word (negative in the first word)
        u64 v;
        if (s >= 0) {
            v = window32(src, (u64)s);
            const u64 left = n_bases - (u64)s;
            if (left < 32) v &= ~0ull << (64 - 2 * left);
        } else {
            const u32 skip = (u32)(-s);                                        // 1..31 bases of the word belong to what is there already
            v = window32(src, 0);
            if (n_bases < 32) v &= n_bases ? ~0ull << (64 - 2 * n_bases) : 0ull;
            v >>= 2 * skip;
        }
        dst[w] |= v;
    }
}

int scan_u32(gasm_ctx* ctx, const u32* in, u64 n, DBuf& bsum, u64* out, u64* total_dev) {
    const u32 nb = std::max<u32>(1, (u32)((n + SB - 1) / SB));
    GCHK(bsum.ensure((size_t)nb * 8 + 8));
    hipLaunchKernelGGL(k_ing_block_sum, dim3(nb), dim3(1024), 0, ctx->stream, in, n, bsum.as<u64>());
    hipLaunchKernelGGL(k_ing_scan_sums, dim3(1), dim3(1024), 0, ctx->stream, bsum.as<u64>(), nb, total_dev);
    hipLaunchKernelGGL(k_ing_block_scan, dim3(nb), dim3(1024), 0, ctx->stream, in, n, bsum.as<u64>(), out);
    HIPCHK(hipGetLastError());
    return GASM_OK;
}

// the whole (inflated) file; GASM_ERR_INVALID for unreadable / truncated / corrupt input (seqio.cpp's rules)
int slurp(const char* path, std::vector<char>& text) {
    gzFile f = gzopen(path, "rb");
    if (!f) { gasm_set_error("cannot open %s", path); return GASM_ERR_INVALID; }
    gzbuffer(f, 1 << 20);
    text.clear();
    std::vector<char> buf(1 << 22);
    int st = GASM_OK;
    for (;;) {
        const int n = gzread(f, buf.data(), (unsigned)buf.size());
        if (n > 0) { text.insert(text.end(), buf.data(), buf.data() + n); continue; }
        int code = 0;
        const char* msg = gzerror(f, &code);
        if (n < 0 || !gzeof(f) || (code != Z_OK && code != Z_STREAM_END)) {
            gasm_set_error("%s: %s (truncated or corrupt file: its reads are not used)", path, msg && *msg ? msg : "read error");
            st = GASM_ERR_INVALID;
        }
        break;
    }
    gzclose(f);
    return st;
}

struct Scratch {
    DBuf text, chunk_cnt, chunk_base, ls, l_a, l_len, l_first, l_bad, info, hdr, hdr_ex, rec_bad, rec_line, plen, pbase, keep, keep_ex, bsum, totals, words, read_off;
    void release() {
        for (DBuf* b : {&text, &chunk_cnt, &chunk_base, &ls, &l_a, &l_len, &l_first, &l_bad, &info, &hdr, &hdr_ex, &rec_bad, &rec_line, &plen, &pbase, &keep,
                        &keep_ex, &bsum, &totals, &words, &read_off})
            b->release();
    }
};

// One file's text -> its kept reads appended to (d_stream, read_off).  *handled = false: not a text the device path
// recognises (the caller takes the host reader, which also words the error).
int parse_text_device(gasm_ctx* ctx, const char* path, const std::vector<char>& text, bool error_on_non_acgt, Scratch& sc, DBuf& d_stream, u64 stream_cap_words,
                      std::vector<u64>& read_off, u64* n_kept, u64* n_dropped, bool* handled) {
    *handled = false;
    *n_kept = 0;
    const u64 n = text.size();
    size_t p0 = 0;
    while (p0 < n && (text[p0] == '\n' || text[p0] == '\r')) ++p0;          // (the host reader skips empty lines before the first record)
    if (p0 == n) { *handled = true; return GASM_OK; }                        // empty file: no reads
    const char first = text[p0];
    if (first != '@' && first != '>') return GASM_OK;
    const bool fastq = first == '@';
    GCHK(sc.text.ensure(n + 16));
    HIPCHK(hipMemcpyAsync(sc.text.p, text.data(), n, hipMemcpyHostToDevice, ctx->stream));
    const u8* d_text = sc.text.as<u8>();
    // ---- lines
    const u32 n_chunks = (u32)((n + CH - 1) / CH);
    GCHK(sc.chunk_cnt.ensure((size_t)n_chunks * 4 + 8));
    GCHK(sc.chunk_base.ensure((size_t)n_chunks * 8 + 16));
    GCHK(sc.totals.ensure(64));
    hipLaunchKernelGGL(k_ing_count_nl, dim3(n_chunks), dim3(256), 0, ctx->stream, d_text, n, sc.chunk_cnt.as<u32>());
    GCHK(scan_u32(ctx, sc.chunk_cnt.as<u32>(), n_chunks, sc.bsum, sc.chunk_base.as<u64>(), sc.totals.as<u64>()));
    u64 n_nl = 0;
    HIPCHK(hipMemcpyAsync(&n_nl, sc.totals.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (n_nl >= 0xFFFFFFF0ull) return GASM_OK;
    const u32 n_lines = (u32)n_nl + 1;                   // the last line may lack its newline (or be empty)
    GCHK(sc.ls.ensure(((size_t)n_lines + 2) * 8));
    const u64 zero = 0, end_mark = n + 1;
    HIPCHK(hipMemcpyAsync(sc.ls.p, &zero, 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(sc.ls.as<u64>() + n_lines, &end_mark, 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_ing_line_starts, dim3(n_chunks), dim3(256), 0, ctx->stream, d_text, n, sc.chunk_base.as<u64>(), sc.ls.as<u64>());
    GCHK(sc.l_a.ensure((size_t)n_lines * 8)); GCHK(sc.l_len.ensure((size_t)n_lines * 4)); GCHK(sc.l_first.ensure(n_lines)); GCHK(sc.l_bad.ensure(n_lines));
    const u32 lg = (n_lines + 255) / 256;
    hipLaunchKernelGGL(k_ing_lines, dim3(lg), dim3(256), 0, ctx->stream, d_text, sc.ls.as<u64>(), n_lines, fastq ? 0 : 1, sc.l_a.as<u64>(), sc.l_len.as<u32>(),
                       sc.l_first.as<u8>(), sc.l_bad.as<u8>());
    GCHK(sc.info.ensure(16));
    HIPCHK(hipMemsetAsync(sc.info.p, 0, 16, ctx->stream));
    u32 n_rec = 0, n_eff = n_lines;
    GCHK(sc.hdr.ensure((size_t)n_lines * 4 + 8)); GCHK(sc.hdr_ex.ensure((size_t)n_lines * 8 + 8));
    if (fastq) {
        hipLaunchKernelGGL(k_ing_last_nonblank, dim3(lg), dim3(256), 0, ctx->stream, sc.l_len.as<u32>(), sc.l_first.as<u8>(), n_lines, sc.info.as<u32>());
        u32 h[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(h, sc.info.p, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        n_eff = h[0];
        if ((n_eff & 3u) == 1u || (n_eff & 3u) == 2u) return GASM_OK;         // a record without its '+' line: the host reader's error
        hipLaunchKernelGGL(k_ing_fastq_check, dim3(std::max(1u, (n_eff + 255) / 256)), dim3(256), 0, ctx->stream, sc.l_len.as<u32>(), sc.l_first.as<u8>(), n_eff,
                           sc.info.as<u32>());
        HIPCHK(hipMemcpyAsync(h, sc.info.p, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (h[1]) return GASM_OK;                                              // blank lines between records, a line out of place: host reader
        n_rec = (n_eff + 3) / 4;
    } else {
        hipLaunchKernelGGL(k_ing_hdr_flags, dim3(lg), dim3(256), 0, ctx->stream, sc.l_first.as<u8>(), n_lines, sc.hdr.as<u32>());
        GCHK(scan_u32(ctx, sc.hdr.as<u32>(), n_lines, sc.bsum, sc.hdr_ex.as<u64>(), sc.totals.as<u64>()));
        u64 nh = 0;
        HIPCHK(hipMemcpyAsync(&nh, sc.totals.p, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        n_rec = (u32)nh;
    }
    GCHK(sc.rec_bad.ensure((size_t)n_rec * 4 + 8)); GCHK(sc.rec_line.ensure((size_t)n_rec * 4 + 8));
    GCHK(sc.keep.ensure((size_t)n_rec * 4 + 8)); GCHK(sc.keep_ex.ensure((size_t)n_rec * 8 + 8));
    GCHK(sc.plen.ensure((size_t)n_lines * 4 + 8)); GCHK(sc.pbase.ensure((size_t)n_lines * 8 + 8));
    HIPCHK(hipMemsetAsync(sc.rec_bad.p, 0, (size_t)n_rec * 4 + 8, ctx->stream));
    const u32 rg = std::max(1u, (n_rec + 255) / 256);
    if (fastq) hipLaunchKernelGGL(k_ing_fastq_records, dim3(rg), dim3(256), 0, ctx->stream, sc.l_bad.as<u8>(), n_rec, sc.rec_bad.as<u32>(), sc.rec_line.as<u32>());
    else hipLaunchKernelGGL(k_ing_fasta_records, dim3(lg), dim3(256), 0, ctx->stream, sc.hdr.as<u32>(), sc.hdr_ex.as<u64>(), sc.l_bad.as<u8>(), n_lines,
                            sc.rec_bad.as<u32>(), sc.rec_line.as<u32>());
    hipLaunchKernelGGL(k_ing_piece_len, dim3(std::max(lg, rg)), dim3(256), 0, ctx->stream, fastq ? 1 : 0, sc.hdr.as<u32>(), sc.hdr_ex.as<u64>(), sc.l_len.as<u32>(),
                       sc.rec_bad.as<u32>(), n_lines, n_eff, sc.plen.as<u32>(), sc.keep.as<u32>(), n_rec);
    u64* const d_tot = sc.totals.as<u64>();             // [0] bases, [1] kept records
    GCHK(scan_u32(ctx, sc.plen.as<u32>(), n_lines, sc.bsum, sc.pbase.as<u64>(), d_tot));
    GCHK(scan_u32(ctx, sc.keep.as<u32>(), n_rec, sc.bsum, sc.keep_ex.as<u64>(), d_tot + 1));
    u64 tot[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(tot, d_tot, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const u64 bases = tot[0], kept = tot[1];
    if (kept < n_rec && error_on_non_acgt) { gasm_set_error("%s: a read holds a base outside ACGT", path); *handled = true; return GASM_ERR_NON_ACGT; }
    // ---- read offsets and the packed bases of this file, then into the batch's stream
    GCHK(sc.read_off.ensure((kept + 2) * 8));
    hipLaunchKernelGGL(k_ing_read_off, dim3(rg), dim3(256), 0, ctx->stream, sc.keep.as<u32>(), sc.keep_ex.as<u64>(), sc.rec_line.as<u32>(), sc.pbase.as<u64>(), n_rec,
                       d_tot, sc.read_off.as<u64>());
    const u64 nw = (bases + 31) / 32 + 4;
    GCHK(sc.words.ensure(nw * 8));
    hipLaunchKernelGGL(k_ing_pack, dim3((u32)std::min<u64>((nw + 255) / 256, (u64)ctx->n_cu * 16)), dim3(256), 0, ctx->stream, d_text, sc.l_a.as<u64>(),
                       sc.plen.as<u32>(), sc.pbase.as<u64>(), n_lines, d_tot, sc.words.as<u64>(), nw);
    const u64 dst_off = read_off.back();
    if ((dst_off + bases + 31) / 32 + 4 > stream_cap_words) { gasm_set_error("internal: packed stream too small"); return GASM_ERR_STATE; }
    if (bases) hipLaunchKernelGGL(k_ing_append, dim3((u32)std::min<u64>((nw + 255) / 256, (u64)ctx->n_cu * 16)), dim3(256), 0, ctx->stream, sc.words.as<u64>(), d_tot,
                                  d_stream.as<u64>(), dst_off);
    HIPCHK(hipGetLastError());
    std::vector<u64> ro(kept + 1);
    HIPCHK(hipMemcpyAsync(ro.data(), sc.read_off.p, (kept + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (u64 r = 1; r <= kept; ++r) read_off.push_back(dst_off + ro[r]);
    *n_kept = kept;
    *n_dropped += n_rec - kept;
    *handled = true;
    return GASM_OK;
}

}  // namespace

// One file per segment, parsed and packed on the device.  d_words: the batch's 2-bit stream (+ 4 zero padding words), read_off /
// seg: host directories.  on_device[f]: the device path read file f (else the host reader: an irregular text).
int ingest_files_device(gasm_ctx* ctx, const char* const* paths, u32 n_files, bool error_on_non_acgt, DBuf& d_words, std::vector<u64>& read_off,
                        std::vector<u64>& seg, u64* dropped, std::vector<u8>& on_device) {
    HIPCHK(hipSetDevice(ctx->device));
    GasmRange range("gasm:ingest (device record scan + 2-bit pack)");
    std::vector<std::vector<char>> texts(n_files);
    u64 cap_bases = 0;
    for (u32 f = 0; f < n_files; ++f) {
        if (!paths[f]) { gasm_set_error("paths[%u] is null", f); return GASM_ERR_INVALID; }
        GCHK(slurp(paths[f], texts[f]));
        cap_bases += texts[f].size();
    }
    const u64 cap_words = cap_bases / 32 + 8;
    GCHK(d_words.ensure(cap_words * 8));
    HIPCHK(hipMemsetAsync(d_words.p, 0, cap_words * 8, ctx->stream));
    read_off.assign(1, 0);
    seg.assign((size_t)n_files + 1, 0);
    on_device.assign(n_files, 0);
    *dropped = 0;
    Scratch sc;
    struct Rel { Scratch& s; ~Rel() { s.release(); } } rel{sc};
    for (u32 f = 0; f < n_files; ++f) {
        u64 kept = 0;
        bool handled = false;
        GCHK(parse_text_device(ctx, paths[f], texts[f], error_on_non_acgt, sc, d_words, cap_words, read_off, &kept, dropped, &handled));
        if (!handled) {
            // the host reader (its grammar is the definition; it also words the error of a malformed file), then the same append
            gasm_host::PackedReads pr;
            u64 hk = 0;
            GCHK(gasm_host::read_sequence_file(paths[f], error_on_non_acgt, pr, &hk, dropped));
            const u64 nwh = pr.words.size();
            GCHK(sc.words.ensure((nwh + 4) * 8));
            HIPCHK(hipMemsetAsync(sc.words.p, 0, (nwh + 4) * 8, ctx->stream));
            if (nwh) HIPCHK(hipMemcpyAsync(sc.words.p, pr.words.data(), nwh * 8, hipMemcpyHostToDevice, ctx->stream));
            GCHK(sc.totals.ensure(64));
            HIPCHK(hipMemcpyAsync(sc.totals.p, &pr.total_bases, 8, hipMemcpyHostToDevice, ctx->stream));
            const u64 dst_off = read_off.back();
            if (pr.total_bases) hipLaunchKernelGGL(k_ing_append, dim3((u32)std::min<u64>((nwh + 255) / 256 + 1, (u64)ctx->n_cu * 16)), dim3(256), 0, ctx->stream,
                                                   sc.words.as<u64>(), sc.totals.as<u64>(), d_words.as<u64>(), dst_off);
            HIPCHK(hipStreamSynchronize(ctx->stream));
            for (size_t r = 1; r < pr.read_off.size(); ++r) read_off.push_back(dst_off + pr.read_off[r]);
            kept = hk;
        } else {
            on_device[f] = 1;
        }
        seg[f + 1] = seg[f] + kept;
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return GASM_OK;
}
