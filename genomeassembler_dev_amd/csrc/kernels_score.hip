// kernels_score.hip — breakage scoring kernels (gfx950).
//
// The reference scores a path by, for every unique read, std::string::find(read) in the path (first occurrence
// only), turning the hit position into an octamer window and accumulating count x probability
// (lib/DeNovoAssembler.cpp:346-426).  Same results, different route:
//   k_read_insert / k_path_scan / k_first_to_poscnt
//                  the reads indexed by their first w bases (w = min(32, shortest read)); one thread per path position
//                  looks its window up, compares in full and keeps the first occurrence per (path, read); the first
//                  occurrences become the position counters (lib/DeNovoAssembler.cpp:346-392: first find() per path)
//   k_path_reduce  per path: sum over positions of count x prob(window(position)) in a fixed order (:394-426)
// Reads are not de-duplicated first: the reference does that (:334-337) only to save work, a read occurring c times
// adds c to the same counter either way.
// Counting per *position* instead of per table row makes the sums integer-exact up to the final FP64 reduction,
// which runs in a fixed lane/tree order: results are bit-reproducible run to run (the reference's own order is its
// hash map's iteration order, hence the 1e-9 tolerance of the parity tests).
#include "device_utils.h"
#include "kernels.h"

// Direct-address probability table over all ACGT strings of length 1..8: index = (4^L - 4)/3 + value.
#define GASM_DIRECT_ROWS 87380
__device__ __forceinline__ u32 direct_base(u32 L) { return ((1u << (2 * L)) - 4u) / 3u; }

// Window of the break at hit position j (lib/DeNovoAssembler.cpp:366-386).  Returns false when the window is empty
// (only for an empty path); *idx is the direct-table index of the (possibly end-truncated) window.
__device__ __forceinline__ bool break_window(const u64* __restrict__ words, u64 pbase, u32 plen, u32 j, int kmer, u32* idx) {
    const int st = (int)j - kmer / 2;
    const u32 start = st > 0 ? (u32)st : 0u;
    u32 width = 8;
    if (start == 0) {
        if (j == 1) width = 2;
        else if (j == 2) width = 4;
        else if (j == 3) width = 6;
    }
    if (start >= plen) return false;
    const u32 avail = plen - start;
    const u32 wd = width < avail ? width : avail;
    *idx = direct_base(wd) + (u32)kmer_at(words, pbase + start, (int)wd);
    return true;
}

// ---- API path (arbitrary paths, e.g. the scaffolds of assemble_contigs): index the READS, scan the PATHS ----------
// lib/DeNovoAssembler.cpp:346-392 tests every read against every path and takes the first occurrence.  Scaffolds repeat
// the same genome over and over (thousands of paths, each most of the segment), so an index over path positions has
// chains as long as the number of paths; the reads of a segment, on the other hand, are few and mostly start at
// different places.  So: a small open-addressing table of the reads keyed by their first w bases (k_read_insert); one
// thread per path position looks its window up, verifies every read that starts like it, and keeps the smallest
// position per (path, read) with an atomicMin in a dense table (k_path_scan); the first occurrences then become the
// position counters the reductions work from (k_first_to_poscnt).
__global__ void __launch_bounds__(GASM_WG) k_read_insert(ReadSet rs, SeedTable st, int w) {
    const u32 seg = blockIdx.y;
    const u64 r = rs.seg_read_off[seg] + (u64)blockIdx.x * GASM_WG + threadIdx.x;
    if (r >= rs.seg_read_off[seg + 1]) return;
    u64 p0; u32 len;
    read_span(rs, r, &p0, &len);
    if (len < (u32)w || len == 0) return;
    const u64 seed = kmer_at(rs.words, p0, w);
    const u64 tb = st.tbl_off[seg];
    const u32 mask = (u32)(st.tbl_off[seg + 1] - tb) - 1;
    u32 h = hash64(seed) & mask;
    while (true) {
        const u32 old = atomicCAS(&st.gpos[tb + h], GASM_NONE32, (u32)r);
        if (old == GASM_NONE32) { st.seed[tb + h] = seed; break; }
        h = (h + 1) & mask;
    }
}

// `len` bases of two packed streams equal?  160 bases per round: up to six words from either stream as three 16-byte loads
// (independent; a 150-base read is one round of six requests — word by word and 128 bases per round it was twenty, and the
// scorer is bound by the requests it sends), aligned in registers.  Both streams end in four padding words; the third pair
// is only touched when more than 96 bases remain, i.e. when the fourth word still holds bases.
typedef unsigned long long gasm_u64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gasm_u64x2 load_pair(const u64* __restrict__ p) {
    gasm_u64x2 v;
    __builtin_memcpy(&v, p, 16);                  // (8-byte aligned: one global_load_dwordx4)
    return v;
}
__device__ __forceinline__ bool bases_equal(const u64* __restrict__ aw, u64 a0, const u64* __restrict__ bw, u64 b0, u32 len) {
    bool same = true;
    for (u32 o = 0; o < len && same; o += 160) {
        const u64 ra = (a0 + o) >> 5, ca = (b0 + o) >> 5;
        const u32 sr = (u32)((a0 + o) & 31) << 1, sc = (u32)((b0 + o) & 31) << 1;
        const u32 nb = len - o;
        u64 rw[6], cw[6];
        const gasm_u64x2 r0 = load_pair(aw + ra), r1 = load_pair(aw + ra + 2), c0 = load_pair(bw + ca), c1 = load_pair(bw + ca + 2);
        gasm_u64x2 r2 = {0, 0}, c2 = {0, 0};
        if (nb > 96) { r2 = load_pair(aw + ra + 4); c2 = load_pair(bw + ca + 4); }
        rw[0] = r0.x; rw[1] = r0.y; rw[2] = r1.x; rw[3] = r1.y; rw[4] = r2.x; rw[5] = r2.y;
        cw[0] = c0.x; cw[1] = c0.y; cw[2] = c1.x; cw[3] = c1.y; cw[4] = c2.x; cw[5] = c2.y;
#pragma unroll
        for (u32 j = 0; j < 5; ++j) {
            if (32 * j >= nb) break;
            const u32 left = nb - 32 * j, nbase = left < 32 ? left : 32;
            const u64 x = funnel64(rw[j], rw[j + 1], sr), y = funnel64(cw[j], cw[j + 1], sc);
            same = same && ((x ^ y) >> (64 - 2 * nbase)) == 0;
        }
    }
    return same;
}

// first[first_off[seg] + (path - first path of seg) * reads of seg + (read - first read of seg)] = smallest global base
// position at which the read occurs in the path (GASM_NONE32: nowhere)
// A launch covers all segments (seg0 = 0, path_hi = 0: blockIdx.y = segment, rows at first_off[segment]) or one slice
// [path_lo, path_hi) of the paths of segment seg0 (rows from 0): the host slices when paths x reads would not fit.
__global__ void __launch_bounds__(GASM_WG) k_path_scan(ReadSet rs, PathSet ps, SeedTable st, const u64* __restrict__ seg_base_off, int w,
                                                       const u64* __restrict__ first_off, u32* __restrict__ first, u32 seg0, u32 path_lo,
                                                       u32 path_hi) {
    __shared__ u32 s_c0;
    const u32 seg = seg0 + blockIdx.y;
    const bool slice = path_hi != 0;
    const u64 lo = slice ? ps.p_off[path_lo] : seg_base_off[seg], hi = slice ? ps.p_off[path_hi] : seg_base_off[seg + 1];
    const u64 g0 = lo + (u64)blockIdx.x * GASM_WG;
    if (g0 >= hi) return;
    const u32 pfirst = slice ? path_lo : ps.seg_path_off[seg], plast = slice ? path_hi : ps.seg_path_off[seg + 1];
    if (threadIdx.x == 0) s_c0 = pfirst + upper_seg<u64>(ps.p_off + pfirst, plast - pfirst, g0);   // path of the block's first base
    __syncthreads();
    const u64 g = g0 + threadIdx.x;
    if (g >= hi) return;
    u32 c = s_c0;
    while (g >= ps.p_off[c + 1]) ++c;             // a block spans a path boundary now and then (empty paths: several)
    const u64 pend = ps.p_off[c + 1];
    if (g + (u64)w > pend) return;
    const u64 seed = kmer_at(ps.words, g, w);
    const u64 tb = st.tbl_off[seg];
    const u32 mask = (u32)(st.tbl_off[seg + 1] - tb) - 1;
    const u64 rfirst = rs.seg_read_off[seg], nreads = rs.seg_read_off[seg + 1] - rfirst;
    u32* const row = first + (slice ? 0ull : first_off[seg]) + (u64)(c - pfirst) * nreads;
    for (u32 h = hash64(seed) & mask;; h = (h + 1) & mask) {
        const u32 r = st.gpos[tb + h];
        if (r == GASM_NONE32) break;
        if (st.seed[tb + h] != seed) continue;
        u64 p0; u32 len;
        read_span(rs, r, &p0, &len);
        if (g + len > pend) continue;
        if (!bases_equal(rs.words, p0, ps.words, g, len)) continue;
        atomicMin(&row[r - rfirst], (u32)g);
    }
}

__global__ void __launch_bounds__(GASM_WG) k_first_to_poscnt(const u32* __restrict__ first, u64 n, u32* __restrict__ poscnt) {
    const u64 i = (u64)blockIdx.x * GASM_WG + threadIdx.x;
    if (i >= n) return;
    const u32 g = first[i];
    if (g != GASM_NONE32) atomicAdd(&poscnt[g], 1u);   // kmer_breaks of the path = sum of its position counters (k_path_reduce)
}

// Batch path: the paths are the contigs of the same build, so the index already exists.  A read of length >= k starts
// with a k-mer; that k-mer is one distinct edge (bucket, bin, a one-or-two-key search); list ranking left (head,
// distance) on every edge, i.e. the contig and the offset of the k-mer inside it.  Every k-mer lies on at most one
// contig, once, so this is the only place the read can occur (lib/DeNovoAssembler.cpp:360).
// Round 3: the rest of the read is NOT compared there any more.  The graph was built from these very reads, so every k-mer of
// the read is an edge and consecutive k-mers are consecutive edges; inside a contig every node between two edges has exactly one
// out-edge (that is what ends a contig: lib/DeNovoAssembler.cpp:161-189), so a walk that enters the contig at edge d is forced
// along it — if the read fits between d and the contig's end it IS the contig's text there, and if it does not fit it crosses
// a branching node and occurs in no contig.  The comparison was six of the scorer's ~twelve scattered requests per read
// (k_score_reads_graph 0.064 -> see DESIGN.md §4); `verify` != 0 (GASM_SCORE_VERIFY=1) still makes it, and a mismatch then raises
// bit 2 of the build's flags — the argument above checked against the data.
// Returns the global base position of the hit, or ~0 when the read matches no contig.
template <class K>
__device__ __forceinline__ u64 graph_match(const ReadSet& rs, const GraphView& gv, const u64* __restrict__ link,
                                           const u32* __restrict__ e_cid, const PathSet& ps, u32 seg, u64 r, u32* path, u64* pbeg, u64* pend,
                                           int verify, u32* __restrict__ verify_flag) {
    u64 p0; u32 len;
    read_span(rs, r, &p0, &len);
    if (len < (u32)gv.k) return ~0ull;
    const K key = kmer_key_at<K>(rs.words, p0, gv.k);
    u32 hi;
    const u32 e = graph_lower_bound<K>(gv, seg, key, &hi);
    if (e >= hi || !keq(reinterpret_cast<const K*>(gv.dk_key)[e], key)) return ~0ull;
    const u64 l = link[e];
    const u32 a = (u32)(l >> 32);
    if (a == GASM_NONE32 || !(l & GASM_LINK_DONE)) return ~0ull;   // on an isolated cycle: part of no contig
    const u32 c = e_cid[a];
    *path = c;
    const gasm_u64x2 po = load_pair(ps.p_off + c);              // the contig's first base and its end
    *pbeg = po.x; *pend = po.y;
    const u64 g = po.x + ((u32)l & 0x7FFFFFFFu);
    if (g + len > po.y) return ~0ull;
    if (verify) {
        // (128 bases per round, aligned in registers: word-by-word with an early exit was a chain of dependent round trips)
        if (!bases_equal(rs.words, p0, ps.words, g, len)) { atomicOr(verify_flag, 1u); return ~0ull; }
    }
    return g;
}

// Batch scoring without position counters.  bp_score of a path = sum over the reads that occur in it of
// prob(window(hit position)) (lib/DeNovoAssembler.cpp:389-413 sums count x prob per table row: the same terms grouped
// differently).  Each read adds 1 to its path's count and round(prob * 2^fx_shift) to its path's 64-bit fixed-point sum:
// integer additions commute, so the result does not depend on the order the reads arrive in (bit-reproducible without a
// fixed reduction tree).  A workgroup keeps the accumulators of its segment's paths in LDS (ds_add_u32 / ds_add_u64) and
// flushes the non-zero ones with global atomics at the end; segments with more paths than fit go to global atomics
// directly.  A workgroup = one slice of one segment's reads (seg_chunk, device_utils.h).
template <class K>
__global__ void __launch_bounds__(GASM_WG) k_score_reads_graph(ReadSet rs, GraphView gv, const u64* __restrict__ link,
                                                               const u32* __restrict__ e_cid, PathSet ps,
                                                               const long long* __restrict__ dfix, int kmer, u32 reads_per_wg, u32 chunks,
                                                               u32 lds_paths, u32* __restrict__ cnt, unsigned long long* __restrict__ sum, int verify,
                                                               u32* __restrict__ verify_flag) {
    // accumulators of the segment's paths: `lds_paths` (<= GASM_SCORE_PATH_CAP) of each, sized by the launch — a fixed
    // 72 KB would leave two workgroups per CU, and this kernel is a chain of dependent gathers that lives on occupancy
    extern __shared__ unsigned long long s_sum[];
    u32* s_cnt = reinterpret_cast<u32*>(s_sum + lds_paths);
    u32 seg, chunk;
    if (!seg_chunk(rs.n_segments, chunks, &seg, &chunk)) return;      // a segment's graph and contigs stay in one XCD's L2
    const u64 r0 = rs.seg_read_off[seg] + (u64)chunk * reads_per_wg;
    const u64 rseg_end = rs.seg_read_off[seg + 1];
    if (r0 >= rseg_end) return;
    const u64 r1 = r0 + reads_per_wg < rseg_end ? r0 + reads_per_wg : rseg_end;
    const u32 pfirst = ps.seg_path_off[seg], np = ps.seg_path_off[seg + 1] - pfirst;
    const bool in_lds = np <= lds_paths;
    if (in_lds) for (u32 i = threadIdx.x; i < np; i += GASM_WG) { s_sum[i] = 0; s_cnt[i] = 0; }
    __syncthreads();
    for (u64 r = r0 + threadIdx.x; r < r1; r += GASM_WG) {
        u32 c = 0;
        u64 pb = 0, pe = 0;
        const u64 g = graph_match<K>(rs, gv, link, e_cid, ps, seg, r, &c, &pb, &pe, verify, verify_flag);
        if (g == ~0ull) continue;
        const u32 plen = (u32)(pe - pb);
        u32 idx;
        long long fx = 0;
        if (break_window(ps.words, pb, plen, (u32)(g - pb), kmer, &idx)) fx = dfix[idx];
        if (in_lds) { atomicAdd(&s_cnt[c - pfirst], 1u); atomicAdd(&s_sum[c - pfirst], (unsigned long long)fx); }
        else { atomicAdd(&cnt[c], 1u); atomicAdd(&sum[c], (unsigned long long)fx); }
    }
    __syncthreads();
    if (in_lds) for (u32 i = threadIdx.x; i < np; i += GASM_WG) {
        const u32 c = s_cnt[i];
        if (c) { atomicAdd(&cnt[pfirst + i], c); atomicAdd(&sum[pfirst + i], s_sum[i]); }
    }
}

template __global__ void k_score_reads_graph<u64>(ReadSet, GraphView, const u64*, const u32*, PathSet, const long long*, int, u32, u32, u32, u32*,
                                                  unsigned long long*, int, u32*);
template __global__ void k_score_reads_graph<K128>(ReadSet, GraphView, const u64*, const u32*, PathSet, const long long*, int, u32, u32, u32, u32*,
                                                   unsigned long long*, int, u32*);

// The batch scorer's accumulators, cleared for the paths there are (*n_paths_p lives on the device: the contigs of a build
// the host has not waited for; a memset would have to cover the upper bound).
__global__ void __launch_bounds__(GASM_WG) k_score_zero(u32* __restrict__ cnt, unsigned long long* __restrict__ sum, const u32* __restrict__ n_paths_p) {
    const u32 n = *n_paths_p;
    for (u32 p = blockIdx.x * GASM_WG + threadIdx.x; p < n; p += gridDim.x * GASM_WG) { cnt[p] = 0; sum[p] = 0; }
}

// Fixed-point sums -> the reference's per-path numbers.  `seg_empty`: empty reads of the path's segment, each a hit at
// position 0 of every path (std::string::find("") == 0).  Grid-stride over *n_paths_p paths.
__global__ void __launch_bounds__(GASM_WG) k_score_finish(PathSet ps, const u32* __restrict__ cnt, const unsigned long long* __restrict__ sum,
                                                          const long long* __restrict__ dfix, const u64* __restrict__ seg_empty,
                                                          int kmer, double inv_scale, double* __restrict__ bp_score,
                                                          double* __restrict__ norm_freq, double* __restrict__ norm_len,
                                                          int32_t* __restrict__ kmer_breaks, int32_t* __restrict__ seq_len,
                                                          const u32* __restrict__ n_paths_p) {
    const u32 n_paths = *n_paths_p;
    for (u32 p = blockIdx.x * GASM_WG + threadIdx.x; p < n_paths; p += gridDim.x * GASM_WG) {
        const u64 pb = ps.p_off[p];
        const u32 len = (u32)(ps.p_off[p + 1] - pb);
        u32 tot = cnt[p];
        long long fs = (long long)sum[p];
        if (seg_empty) {
            const u32 seg = upper_seg<u32>(ps.seg_path_off, ps.n_segments, p);
            const u32 e = (u32)seg_empty[seg];
            if (e) {
                u32 idx;
                tot += e;
                if (break_window(ps.words, pb, len, 0, kmer, &idx)) fs += (long long)e * dfix[idx];
            }
        }
        const double s1 = (double)fs * inv_scale;
        bp_score[p] = s1;
        norm_freq[p] = tot ? s1 / (double)tot : 0.0;
        norm_len[p] = s1 / (double)(int32_t)len;
        kmer_breaks[p] = (int32_t)tot;
        seq_len[p] = (int32_t)len;
    }
}

__device__ __forceinline__ u32 wave_sum_u32(u32 v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__device__ __forceinline__ double wave_sum_fixed(double v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// One workgroup per path.  First the integer total of the path's position counters (= kmer_breaks; `extra` holds what
// empty reads add to empty paths), then thread t sums positions t, t+256, ... in order, the 64 partials of a wave are
// combined by a fixed butterfly and the four wave sums are added in wave order: a fixed summation order, so scores are
// bit-reproducible.
__global__ void __launch_bounds__(GASM_WG) k_path_reduce(PathSet ps, const u32* __restrict__ poscnt, const u32* __restrict__ extra,
                                                         const double* __restrict__ dprob, int kmer, double* __restrict__ bp_score,
                                                         double* __restrict__ norm_freq, double* __restrict__ norm_len,
                                                         int32_t* __restrict__ kmer_breaks, int32_t* __restrict__ seq_len, u32 n_paths) {
    __shared__ u32 s_t[4];
    __shared__ double s_a[4], s_b[4];
    const u32 p = blockIdx.x;
    if (p >= n_paths) return;
    const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u64 pb = ps.p_off[p];
    const u32 len = (u32)(ps.p_off[p + 1] - pb);
    u32 tsum = 0;
    for (u32 j = threadIdx.x; j < len; j += GASM_WG) tsum += poscnt[pb + j];
    tsum = wave_sum_u32(tsum);
    if (lane == 0) s_t[wv] = tsum;
    __syncthreads();
    const u32 tot = s_t[0] + s_t[1] + s_t[2] + s_t[3] + extra[p];
    const double dtot = (double)tot;
    double s1 = 0.0, s2 = 0.0;
    for (u32 j = threadIdx.x; j < len; j += GASM_WG) {
        const u32 c = poscnt[pb + j];
        if (c) {
            u32 idx;
            if (break_window(ps.words, pb, len, j, kmer, &idx)) {
                const double pr = dprob[idx];
                s1 += pr * (double)c;
                s2 += pr * ((double)c / dtot);
            }
        }
    }
    s1 = wave_sum_fixed(s1);
    s2 = wave_sum_fixed(s2);
    if (lane == 0) { s_a[wv] = s1; s_b[wv] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double a = ((s_a[0] + s_a[1]) + s_a[2]) + s_a[3];
        const double b = ((s_b[0] + s_b[1]) + s_b[2]) + s_b[3];
        bp_score[p] = a;
        norm_freq[p] = b;
        norm_len[p] = a / (double)(int32_t)len;
        kmer_breaks[p] = (int32_t)tot;
        seq_len[p] = (int32_t)len;
    }
}

// Dense per-row break counts (for the reference's path_freq output): freq_cnt[p * n_table + row] += count.
__global__ void __launch_bounds__(GASM_WG) k_path_freq(PathSet ps, const u32* __restrict__ poscnt, const u32* __restrict__ total,
                                                       const int32_t* __restrict__ drow, int kmer, u32 n_table,
                                                       u32* __restrict__ freq_cnt, u32 n_paths) {
    const u32 p = blockIdx.x * (GASM_WG / 64) + (threadIdx.x >> 6);
    if (p >= n_paths) return;
    (void)total;
    const u32 lane = threadIdx.x & 63;
    const u64 pb = ps.p_off[p];
    const u32 len = (u32)(ps.p_off[p + 1] - pb);
    for (u32 j = lane; j < len; j += 64) {
        const u32 c = poscnt[pb + j];
        if (c) {
            u32 idx;
            if (break_window(ps.words, pb, len, j, kmer, &idx)) {
                const int32_t row = drow[idx];
                if (row >= 0) atomicAdd(&freq_cnt[(u64)p * n_table + row], c);
            }
        }
    }
}

// lib/BreakageScorer.cpp:200-215: probability of the kmer-long window at every path position.
__global__ void __launch_bounds__(GASM_WG) k_prob_dist(PathSet ps, const double* __restrict__ dprob, int kmer,
                                                       const u64* __restrict__ pd_off, double* __restrict__ out, u32 n_paths) {
    const u32 p = blockIdx.x * (GASM_WG / 64) + (threadIdx.x >> 6);
    if (p >= n_paths) return;
    const u32 lane = threadIdx.x & 63;
    const u64 pb = ps.p_off[p];
    const u32 len = (u32)(ps.p_off[p + 1] - pb);
    const u32 n = (u32)(pd_off[p + 1] - pd_off[p]);
    (void)len;
    for (u32 j = lane; j < n; j += 64) {
        double v = 0.0;
        if (kmer >= 1 && kmer <= 8) v = dprob[direct_base((u32)kmer) + (u32)kmer_at(ps.words, pb + j, kmer)];
        out[pd_off[p] + j] = v;
    }
}

// ================================================================================================================
// Levenshtein distance of every path against one target (lib/DeNovoAssembler.cpp:41-55 global,
// lib/BreakageScorer.cpp:41-55 infix; the reference calls edlib, path = query, true solution = target).
//
// Myers' bit-vector algorithm in 64-row blocks (the same recurrence as gasm_host::levenshtein), ONE WAVE PER PATH:
// lane b owns block b of a band of 64 blocks (4096 path bases) and walks the target columns one step behind lane b-1,
// so the horizontal delta leaving block b-1 at a column reaches lane b through a DPP wave shift in the next step, and
// so does the target base — no LDS, no barriers.  Paths longer than 4096 bases take several bands; the deltas leaving a
// band's last block are parked in a per-wave global array (one byte per column, written and read 64 columns at a time
// through v_readlane and a lane select) and enter the next band's lane 0.  Only the lane with the path's last block keeps
// the score.  Work is O(|path| * |target| / 64) integer instructions; the waves of a launch are independent.
// ================================================================================================================
__device__ __forceinline__ u32 compress_even32(u64 x) {          // bits 0, 2, 4, ... 62 of x, packed
    x &= 0x5555555555555555ull;
    x = (x | (x >> 1)) & 0x3333333333333333ull;
    x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
    x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
    return (u32)x;
}
// 32 bases (first base most significant) -> the plane of their high code bits and of their low code bits, base i at bit i
__device__ __forceinline__ void code_planes32(u64 w, u32* hi, u32* lo) {
    *lo = __brev(compress_even32(w));
    *hi = __brev(compress_even32(w >> 1));
}
__device__ __forceinline__ int wave_shr1(int v, int lane0) {      // lane b gets lane b-1's v, lane 0 gets lane0
    return __builtin_amdgcn_update_dpp(lane0, v, 0x138, 0xf, 0xf, false);
}

// Round 3: the column step costs about half the instructions of round 2's (117 -> ~60, ISA count) —
//   * a lane's four match masks (one per target base, the valid rows folded in) are made once per band and picked with two
//     bit-selects, instead of rebuilding the comparison from two code planes at every column;
//   * the horizontal delta travels as two bits (+1 / -1) packed with the target base in one register: one wave shift per step;
//   * the chunks of 64 columns in which every lane is inside its band (all but the first and the last one or two of ~780)
//     run a body without any "is this lane active" select; the others keep the predicated body;
//   * the delta leaving the band's last block goes from lane 63 to "its" lane with a readlane / writelane pair, stored once
//     per 64 columns; the score is a 32-bit add under a per-lane mask.
// lane LANE of `old` := sval (v_writelane_b32; this clang has no builtin for it, and the instruction takes one SGPR only: the
// lane select is an immediate, so the chunk body that uses it is unrolled).  The nops cover the wait states the ISA asks for
// after the instruction that wrote the SGPR.
template <int LANE>
__device__ __forceinline__ int lev_writelane(int old, int sval) {
    __asm__ volatile("s_nop 1\n\tv_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(sval), "n"(LANE));
    return old;
}

struct LevLane {
    u64 P[4];          // rows of this block that match target base c, valid rows only
    u64 Pv, Mv;
    u32 tb;            // bit of the block's last row (63, or less in the path's last block)
    int smask;         // -1 in the lane that holds the path's last row (and only in the last band)
};

template <bool PRED>
__device__ __forceinline__ void lev_step(LevLane& L, u32& pk, u32 inj, bool act, int& score) {
    // pk: this lane's (target base | pos << 2 | neg << 3) of the column it has just finished; it becomes the next lane's input
    const u32 x = (u32)__builtin_amdgcn_update_dpp((int)inj, (int)pk, 0x138, 0xf, 0xf, false);      // wave_shr:1, lane 0 keeps inj
    // the target base picks one of the four match masks: two levels of bit-select on 32-bit halves (v_bfe_i32 + v_bfi_b32; written
    // with 64-bit masks the compiler built them with carry chains)
    const u32 m0 = (u32)__builtin_amdgcn_sbfe((int)x, 0, 1), m1 = (u32)__builtin_amdgcn_sbfe((int)x, 1, 1);      // all ones / zero
    const u64 pos = (x >> 2) & 1u, neg = (x >> 3) & 1u;
    auto sel = [](u32 m, u32 a, u32 b) { return (a & m) | (b & ~m); };
    const u32 e_lo = sel(m1, sel(m0, (u32)L.P[3], (u32)L.P[2]), sel(m0, (u32)L.P[1], (u32)L.P[0]));
    const u32 e_hi = sel(m1, sel(m0, (u32)(L.P[3] >> 32), (u32)(L.P[2] >> 32)), sel(m0, (u32)(L.P[1] >> 32), (u32)(L.P[0] >> 32)));
    const u64 eq0 = (u64)e_lo | ((u64)e_hi << 32);
    const u64 xv = eq0 | L.Mv;
    const u64 eq = eq0 | neg;
    const u64 xh = (((eq & L.Pv) + L.Pv) ^ L.Pv) | eq;
    const u64 ph0 = L.Mv | ~(xh | L.Pv), mh0 = L.Pv & xh;
    const u32 hp = (u32)(ph0 >> L.tb) & 1u, hn = (u32)(mh0 >> L.tb) & 1u;
    const u64 ph = (ph0 << 1) | pos, mh = (mh0 << 1) | neg;
    const u64 nPv = mh | ~(xv | ph), nMv = ph & xv;
    const u32 npk = (x & 3u) | (hp << 2) | (hn << 3);
    if (PRED) {
        L.Pv = act ? nPv : L.Pv;
        L.Mv = act ? nMv : L.Mv;
        pk = act ? npk : ((x & 3u) | (pk & 12u));      // the base moves on; the delta of a lane outside its band stays what it was
        score += act ? (((int)hp - (int)hn) & L.smask) : 0;
    } else {
        L.Pv = nPv; L.Mv = nMv; pk = npk;
        score += ((int)hp - (int)hn) & L.smask;
    }
}

// 64 columns in which every lane is inside its band: no "is this lane active" select anywhere
template <bool INFIX, bool LAST_BAND>
__device__ __forceinline__ void lev_chunk(LevLane& L, u32& pk, u32 inj_all, int& score, int& best, int& cout, u8* __restrict__ carry, u32 s0, u32 ln) {
    static_for<64>([&](auto T) {
        constexpr u32 t = T;
        const u32 inj = (u32)__builtin_amdgcn_readlane((int)inj_all, (int)t);
        lev_step<false>(L, pk, inj, true, score);
        if (INFIX) best = score < best ? score : best;
        if (!LAST_BAND) {
            const int h63 = __builtin_amdgcn_readlane((int)((pk >> 2) & 3u), 63);       // pos | neg << 1 of column s0 + t - 63
            cout = lev_writelane<(int)((t + 1) & 63u)>(cout, h63);
            if (t == 62) {
                // columns s0 - 64 .. s0 - 1 are complete: bits -> the byte code (delta + 1) the next band reads
                const u32 c2 = (u32)cout;
                carry[s0 - 64 + ln] = (u8)(1u + (c2 & 1u) - ((c2 >> 1) & 1u));
            }
        }
    });
}

template <bool INFIX>
__device__ __forceinline__ void lev_path(const PathSet& ps, u32 p, const u64* __restrict__ twords, u32 nt, u8* __restrict__ carry, int32_t* __restrict__ out) {
    const u32 ln = threadIdx.x & 63;
    const u64 pb = ps.p_off[p];
    const u32 nq = (u32)(ps.p_off[p + 1] - pb);
    if (nq == 0 || nt == 0) { if (ln == 0) out[p] = 0; return; }     // edlib reports an error, the reference returns 0
    const u32 nblk = (nq + 63) / 64, nbands = (nblk + 63) / 64;
    const u32 last_lane = (nblk - 1) & 63;
    int score = (int)nq, best = (int)nq;
    for (u32 band = 0; band < nbands; ++band) {
        const u32 blk = band * 64 + ln;
        const bool mine = blk < nblk, last_band = band + 1 == nbands;
        LevLane L;
        {
            u64 H = 0, Lo = 0, valid = 0;
            if (mine) {
                const u32 rows = min(64u, nq - blk * 64);
                u32 h0, l0, h1 = 0, l1 = 0;
                code_planes32(window32(ps.words, pb + (u64)blk * 64), &h0, &l0);
                if (rows > 32) code_planes32(window32(ps.words, pb + (u64)blk * 64 + 32), &h1, &l1);
                H = (u64)h0 | ((u64)h1 << 32);
                Lo = (u64)l0 | ((u64)l1 << 32);
                valid = rows == 64 ? ~0ull : ((1ull << rows) - 1);
            }
            L.P[0] = ~H & ~Lo & valid; L.P[1] = ~H & Lo & valid; L.P[2] = H & ~Lo & valid; L.P[3] = H & Lo & valid;
            L.Pv = ~0ull; L.Mv = 0;
            L.tb = blk + 1 == nblk ? ((nq - 1) & 63) : 63u;
            L.smask = (last_band && ln == last_lane) ? -1 : 0;
        }
        // delta entering block 0 of the first band: +1 (global) / 0 (infix, lib/BreakageScorer.cpp: HW mode)
        const u32 hin0 = INFIX ? 0u : 4u;
        u32 pk = 0;               // (a lane before its band hands on "no delta": never consumed)
        int cout = 0;             // deltas leaving the band (+1 coded), the lane of column c is c & 63
        for (u32 s0 = 0; s0 < nt + 63; s0 += 64) {
            const u32 jl = s0 + ln;
            const u32 tch = jl < nt ? (u32)(twords[jl >> 5] >> (62 - 2 * (jl & 31))) & 3u : 0u;
            u32 hbits = hin0;
            if (band) { const u32 cin = jl < nt ? (u32)carry[jl] : 1u; hbits = (cin == 2u ? 4u : 0u) | (cin == 0u ? 8u : 0u); }
            const u32 inj_all = tch | hbits;
            const u32 steps = min(64u, nt + 63 - s0);
            if (s0 >= 64 && s0 + 64 <= nt) {
                // ---- every lane is inside its band for all 64 columns of this chunk: no selects
                if (last_band) lev_chunk<INFIX, true>(L, pk, inj_all, score, best, cout, carry, s0, ln);
                else lev_chunk<INFIX, false>(L, pk, inj_all, score, best, cout, carry, s0, ln);
            } else {
                for (u32 t = 0; t < steps; ++t) {
                    const u32 s = s0 + t;
                    const u32 inj = (u32)__builtin_amdgcn_readlane((int)inj_all, (int)t);
                    const bool act = mine && s >= ln && s - ln < nt;
                    lev_step<true>(L, pk, inj, act, score);
                    if (INFIX) best = score < best ? score : best;
                    if (!last_band) {
                        const int h63 = __builtin_amdgcn_readlane((int)((pk >> 2) & 3u), 63);
                        cout = ln == ((s - 63) & 63u) ? h63 : cout;
                        if (s >= 63 && (((s - 63) & 63u) == 63u || s - 62 == nt)) {                  // uniform: once per 64 columns, and at the end
                            const u32 jj = ((s - 63) & ~63u) + ln;
                            const u32 c2 = (u32)cout;
                            if (jj < nt) carry[jj] = (u8)(1u + (c2 & 1u) - ((c2 >> 1) & 1u));
                        }
                    }
                }
            }
        }
    }
    const int sc = __shfl(INFIX ? best : score, (int)last_lane, 64);
    if (ln == 0) out[p] = (int32_t)sc;
}

__global__ void __launch_bounds__(GASM_WG) k_levenshtein(PathSet ps, u32 n_paths, const u64* __restrict__ twords, u32 nt, int infix,
                                                         u8* __restrict__ carry_ws, u64 carry_stride, int32_t* __restrict__ out) {
    const u32 wave = (blockIdx.x * GASM_WG + threadIdx.x) >> 6, n_waves = gridDim.x * (GASM_WG / 64);
    u8* const carry = carry_ws + (u64)wave * carry_stride;         // nt + 64 bytes of this wave
    for (u32 p = wave; p < n_paths; p += n_waves) {
        if (infix) lev_path<true>(ps, p, twords, nt, carry, out);
        else lev_path<false>(ps, p, twords, nt, carry, out);
    }
}

// ----------------------------------------------------------------------------------------------------------------
// Round 3, second form (k_levenshtein2; GASM_LEV_V=1 keeps the one above): the same systolic band, ~40 instead of ~60
// instructions per column step —
//   * the match mask of a step does not wait for the wave shift: lane b's column at step s is s - b, so the lane keeps
//     its own 64-base window of the target in registers (loaded once per 64 columns) and fetches the mask of that base
//     from LDS (the lane's four masks, written once per band; layout [base][lane]: conflict-free 64-bit reads) — a
//     bit-field extract, an address add and a ds_read_b64 where there were two sign extractions and six bit-selects;
//   * the horizontal delta travels as the producer's own registers: bit 31 of the high words of ph0 and mh0 IS the delta
//     leaving a full block, so the consumer takes both through an or with a wave-shifted operand and shifts them in with
//     v_alignbit — no packing, no unpacking (lane 0's input is or-ed in: a register that is zero in every other lane);
//   * the global distance is read off the LAST column: D[m][n] = n + (vertical +1s) - (vertical -1s), two popcounts per
//     lane and band — nothing per step (the infix minimum over the columns still follows the last row step by step);
//   * the deltas leaving the band collect in two shift registers per lane (v_alignbit again); lane 63 stores 16 bytes
//     per 64 columns and the next band's lane 0 reads them back as bit planes.
// ----------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 lev2_orn(u32 y, u32 m) {          // m | ~y in one instruction: (y & m) | (~y & -1)
    u32 r;
    __asm__("v_bfi_b32 %0, %1, %2, -1" : "=v"(r) : "v"(y), "v"(m));
    return r;
}

struct Lev2 {
    u32 pv_lo, pv_hi, mv_lo, mv_hi;      // vertical deltas of the lane's 64 rows
    u32 oh_p, oh_n;                      // high words of ph0 / mh0 of the last step: bit 31 = the delta leaving a full block
    u32 cp, cn;                          // the last 32 deltas that left this lane (+1s, -1s), newest at bit 0
    // infix, last band: the row of the path's last base (bit tbs of the low or the high word of the lane that holds it)
    u32 tbs; bool tb_hi; int smask, score, best;
};

// one column step of a lane.  eq64: the lane's match mask for this column's base; wp / wn: lane 0's incoming delta at
// bit 31 (zero in the other lanes).  PRED: lanes outside their band keep their state.
template <bool PRED, bool CARRY, bool FOLLOW>
__device__ __forceinline__ void lev2_step(Lev2& L, u64 eq64, u32 wp, u32 wn, bool act) {
    const u32 eq_lo = (u32)eq64, eq_hi = (u32)(eq64 >> 32);
    const u32 xp = (u32)__builtin_amdgcn_update_dpp(0, (int)L.oh_p, 0x138, 0xf, 0xf, true) | wp;      // wave_shr:1, lane 0 gets 0
    const u32 xn = (u32)__builtin_amdgcn_update_dpp(0, (int)L.oh_n, 0x138, 0xf, 0xf, true) | wn;
    const u32 eqn_lo = eq_lo | (xn >> 31);
    const u32 xv_lo = eq_lo | L.mv_lo, xv_hi = eq_hi | L.mv_hi;
    const u64 pv = (u64)L.pv_lo | ((u64)L.pv_hi << 32);
    const u64 sum = ((u64)(eqn_lo & L.pv_lo) | ((u64)(eq_hi & L.pv_hi) << 32)) + pv;
    const u32 xh_lo = ((u32)sum ^ L.pv_lo) | eqn_lo, xh_hi = ((u32)(sum >> 32) ^ L.pv_hi) | eq_hi;
    const u32 ph0_lo = lev2_orn(xh_lo | L.pv_lo, L.mv_lo), ph0_hi = lev2_orn(xh_hi | L.pv_hi, L.mv_hi);
    const u32 mh0_lo = L.pv_lo & xh_lo, mh0_hi = L.pv_hi & xh_hi;
    const u32 ph_lo = __builtin_amdgcn_alignbit(ph0_lo, xp, 31), ph_hi = __builtin_amdgcn_alignbit(ph0_hi, ph0_lo, 31);
    const u32 mh_lo = __builtin_amdgcn_alignbit(mh0_lo, xn, 31), mh_hi = __builtin_amdgcn_alignbit(mh0_hi, mh0_lo, 31);
    const u32 npv_lo = lev2_orn(xv_lo | ph_lo, mh_lo), npv_hi = lev2_orn(xv_hi | ph_hi, mh_hi);
    const u32 nmv_lo = ph_lo & xv_lo, nmv_hi = ph_hi & xv_hi;
    if (PRED) {
        L.pv_lo = act ? npv_lo : L.pv_lo; L.pv_hi = act ? npv_hi : L.pv_hi;
        L.mv_lo = act ? nmv_lo : L.mv_lo; L.mv_hi = act ? nmv_hi : L.mv_hi;
    } else {
        L.pv_lo = npv_lo; L.pv_hi = npv_hi; L.mv_lo = nmv_lo; L.mv_hi = nmv_hi;
    }
    L.oh_p = ph0_hi; L.oh_n = mh0_hi;         // (of a lane outside its band: never consumed by a lane inside its own)
    if (CARRY) { L.cp = __builtin_amdgcn_alignbit(L.cp, ph0_hi, 31); L.cn = __builtin_amdgcn_alignbit(L.cn, mh0_hi, 31); }
    if (FOLLOW) {
        const u32 hp = __builtin_amdgcn_ubfe(L.tb_hi ? ph0_hi : ph0_lo, L.tbs, 1), hn = __builtin_amdgcn_ubfe(L.tb_hi ? mh0_hi : mh0_lo, L.tbs, 1);
        const int d = ((int)hp - (int)hn) & L.smask;
        L.score += (!PRED || act) ? d : 0;
        L.best = L.score < L.best ? L.score : L.best;
    }
}

template <bool INFIX>
// rows: nq bases from base qb of qwords (a lane owns 64 of them), columns: nt bases from base tb0 of twords
__device__ __forceinline__ void lev2_path(const u64* __restrict__ qwords, u64 qb, u32 nq, const u64* __restrict__ twords, u64 tb0, u32 nt,
                                          uint4* __restrict__ carry, u64* __restrict__ lmask, int32_t* __restrict__ out_p) {
    const u32 ln = threadIdx.x & 63;
    const u64 pb = qb;
    const u32 nblk = (nq + 63) / 64, nbands = (nblk + 63) / 64;
    const u32 last_lane = (nblk - 1) & 63, q_last = (nt - 1) >> 6;
    int vsum = 0;                      // global: vertical deltas of the last column, summed over this lane's blocks
    Lev2 L;
    L.score = L.best = (int)nq;        // infix: D[m][0] = m
    for (u32 band = 0; band < nbands; ++band) {
        const u32 blk = band * 64 + ln;
        const bool mine = blk < nblk, last_band = band + 1 == nbands;
        u64 valid = 0;
        {
            u64 H = 0, Lo = 0;
            if (mine) {
                const u32 rows = min(64u, nq - blk * 64);
                u32 h0, l0, h1 = 0, l1 = 0;
                code_planes32(window32(qwords, pb + (u64)blk * 64), &h0, &l0);
                if (rows > 32) code_planes32(window32(qwords, pb + (u64)blk * 64 + 32), &h1, &l1);
                H = (u64)h0 | ((u64)h1 << 32);
                Lo = (u64)l0 | ((u64)l1 << 32);
                valid = rows == 64 ? ~0ull : ((1ull << rows) - 1);
            }
            // (the wave's own 2 KB: a lane reads back only what it wrote itself — no barrier)
            lmask[0 * 64 + ln] = ~H & ~Lo & valid; lmask[1 * 64 + ln] = ~H & Lo & valid;
            lmask[2 * 64 + ln] = H & ~Lo & valid;  lmask[3 * 64 + ln] = H & Lo & valid;
        }
        L.pv_lo = L.pv_hi = 0xFFFFFFFFu; L.mv_lo = L.mv_hi = 0; L.oh_p = L.oh_n = 0; L.cp = L.cn = 0;
        const u32 tb = blk + 1 == nblk ? ((nq - 1) & 63) : 63u;
        L.tbs = tb & 31; L.tb_hi = tb >= 32;
        L.smask = (INFIX && last_band && ln == last_lane) ? -1 : 0;
        u32 snap_p = 0, snap_n = 0;
        // bands with a successor: the columns run until the group of 64 that holds the last one is complete in lane 63
        const u32 s_end = last_band ? nt + 63 : 64 * q_last + 127;
        for (u32 s0 = 0; s0 < s_end; s0 += 64) {
            // lane 0's incoming deltas of columns s0 .. s0 + 63, first column at bit 31 of word 0: +1 everywhere (global) /
            // nothing (infix) in the first band, else what the band before left
            u32 wp0 = 0, wp1 = 0, wn0 = 0, wn1 = 0;
            if (band == 0) { if (!INFIX && ln == 0) wp0 = wp1 = 0xFFFFFFFFu; }
            else if (ln == 0 && (s0 >> 6) <= q_last) { const uint4 c = carry[s0 >> 6]; wp0 = c.x; wp1 = c.y; wn0 = c.z; wn1 = c.w; }
            if (s0 >= 64 && s0 + 64 <= nt) {
                // ---- every lane is inside its band for all 64 columns of this chunk; the lane's columns are s0 - ln .. s0 - ln + 63
                const u64 W0 = window32(twords, tb0 + (u64)(s0 - ln)), W1 = window32(twords, tb0 + (u64)(s0 - ln) + 32);
                const u32 w[4] = {(u32)(W0 >> 32), (u32)W0, (u32)(W1 >> 32), (u32)W1};
                if (last_band) {
                    static_for<64>([&](auto T) {
                        constexpr u32 t = T;
                        const u32 base = __builtin_amdgcn_ubfe(w[t >> 4], 30 - 2 * (t & 15), 2);
                        lev2_step<false, false, INFIX>(L, lmask[base * 64 + ln], (t < 32 ? wp0 : wp1) << (t & 31), (t < 32 ? wn0 : wn1) << (t & 31), true);
                    });
                } else {
                    static_for<64>([&](auto T) {
                        constexpr u32 t = T;
                        const u32 base = __builtin_amdgcn_ubfe(w[t >> 4], 30 - 2 * (t & 15), 2);
                        lev2_step<false, true, false>(L, lmask[base * 64 + ln], (t < 32 ? wp0 : wp1) << (t & 31), (t < 32 ? wn0 : wn1) << (t & 31), true);
                        if (t == 30) { snap_p = L.cp; snap_n = L.cn; }
                        if (t == 62 && ln == 63) carry[(s0 >> 6) - 1] = make_uint4(snap_p, L.cp, snap_n, L.cn);      // columns s0 - 64 .. s0 - 1
                    });
                }
            } else {
                const u32 steps = min(64u, s_end - s0);
                for (u32 t = 0; t < steps; ++t) {
                    const u32 j = s0 + t - ln;            // (wraps for s < ln: not active)
                    const bool act = mine && j < nt;
                    const u64 tj = tb0 + j;
                    const u32 base = act ? (u32)(twords[tj >> 5] >> (62 - 2 * (u32)(tj & 31))) & 3u : 0u;
                    lev2_step<true, true, INFIX>(L, lmask[base * 64 + ln], (t < 32 ? wp0 : wp1) << (t & 31), (t < 32 ? wn0 : wn1) << (t & 31), act);
                    if (t == 30) { snap_p = L.cp; snap_n = L.cn; }
                    if (!last_band && t == 62 && s0 >= 64 && ln == 63) carry[(s0 >> 6) - 1] = make_uint4(snap_p, L.cp, snap_n, L.cn);
                }
            }
        }
        if (mine) vsum += __popcll((((u64)L.pv_hi << 32) | L.pv_lo) & valid) - __popcll((((u64)L.mv_hi << 32) | L.mv_lo) & valid);
    }
    int sc;
    if (INFIX) sc = __shfl(L.best, (int)last_lane, 64);
    else {
        // D[m][n] = D[0][n] + the vertical deltas of column n
        sc = vsum;
        for (int o = 32; o; o >>= 1) sc += __shfl_xor(sc, o, 64);
        sc += (int)nt;
    }
    if (ln == 0) *out_p = (int32_t)sc;
}

#ifndef GASM_LEV2_EU
#define GASM_LEV2_EU 6
#endif
__global__ void __launch_bounds__(GASM_WG) __attribute__((amdgpu_waves_per_eu(GASM_LEV2_EU, GASM_LEV2_EU))) k_levenshtein2(PathSet ps, u32 n_paths, const u64* __restrict__ twords, u32 nt, int infix,
                                                          uint4* __restrict__ carry_ws, u64 carry_stride, int32_t* __restrict__ out) {
    __shared__ u64 s_mask[GASM_WG / 64][4 * 64];
    const u32 wave = (blockIdx.x * GASM_WG + threadIdx.x) >> 6, n_waves = gridDim.x * (GASM_WG / 64);
    uint4* const carry = carry_ws + (u64)wave * carry_stride;         // (nt - 1) / 64 + 1 groups of this wave
    u64* const lmask = s_mask[threadIdx.x >> 6];
    for (u32 p = wave; p < n_paths; p += n_waves) {
        const u64 pb = ps.p_off[p];
        const u32 nq = (u32)(ps.p_off[p + 1] - pb);
        if (nq == 0 || nt == 0) { if ((threadIdx.x & 63) == 0) out[p] = 0; continue; }     // edlib reports an error, the reference returns 0
        if (infix) { lev2_path<true>(ps.words, pb, nq, twords, 0, nt, carry, lmask, out + p); continue; }
        // the global distance is symmetric: whichever of the two strings makes the fuller bands gives the rows (a band is 64
        // lanes x 64 rows whatever is left for the last one; 17 kb against 50 kb: 5 bands x 50 k columns or 13 x 17 k: 11 % fewer steps)
        const u64 cost_q = (u64)((((nq + 63) >> 6) + 63) >> 6) * ((u64)nt + 63), cost_t = (u64)((((nt + 63) >> 6) + 63) >> 6) * ((u64)nq + 63);
        if (cost_t < cost_q) lev2_path<false>(twords, 0, nt, ps.words, pb, nq, carry, lmask, out + p);
        else lev2_path<false>(ps.words, pb, nq, twords, 0, nt, carry, lmask, out + p);
    }
}

// ================================================================================================================
// F4 — the R post-processing of score_solutions() (lib/DeNovoAssembler.R:414-445): the two-sample Kolmogorov-Smirnov
// statistic of a path's path_freq vector against the genome's per-position window probabilities (kmer_from_seq,
// lib/GenerateReads.R:243-259), and the percentage of the genome covered by the solutions.
//
// Both samples of the KS test take their values from table rows: x_i = count_i / total for each of the n_table rows of
// the path (most of them 0), y_j = prob(row of the genome window at j).  So neither vector is built: the genome side is a
// histogram over rows, put into ascending-probability order and prefix-summed once per call (host, 70 k entries), and a
// path's side is its per-row counts.  D = sup_t |Fx(t) - Fy(t)| is attained at a sample value; every distinct value of
// either sample is evaluated with integer counts and one division per side (exact rationals in double; R accumulates
// 1/n.x and -1/n.y in a cumsum, hence 1e-9 in the parity test, not bit equality).
// ================================================================================================================
// rows of the genome's kmer-long windows: hist[row] += 1 (windows absent from the table count nowhere: R's NA, dropped)
__global__ void __launch_bounds__(GASM_WG) k_ks_genome_hist(const u64* __restrict__ gwords, u64 glen, int kmer, const int32_t* __restrict__ drow,
                                                            u32* __restrict__ hist) {
    if (kmer < 1 || kmer > 8 || glen < (u64)kmer) return;
    const u64 n = glen - kmer + 1;
    for (u64 p = (u64)blockIdx.x * GASM_WG + threadIdx.x; p < n; p += (u64)gridDim.x * GASM_WG) {
        const int32_t row = drow[direct_base((u32)kmer) + (u32)kmer_at(gwords, p, kmer)];
        if (row >= 0) atomicAdd(&hist[row], 1u);
    }
}

// #{r : pv[r] <= t} over the ascending probabilities
__device__ __forceinline__ u32 ks_rank_le(const double* __restrict__ pv, u32 n, double t) {
    u32 lo = 0, hi = n;
    while (lo < hi) { const u32 m = (lo + hi) >> 1; if (pv[m] <= t) lo = m + 1; else hi = m; }
    return lo;
}

// One workgroup per path (grid-stride over the paths).  scratch: two rows of n_table u32 per workgroup.
//   pv[r]    the table's probabilities, ascending;  cumy[r] = genome windows with probability <= pv[r] (inclusive prefix
//   sums in that order);  ny = cumy[n_table - 1]
__global__ void __launch_bounds__(GASM_WG) k_path_ks(PathSet ps, const u32* __restrict__ poscnt, const int32_t* __restrict__ drow, int kmer,
                                                     u32 n_table, const double* __restrict__ pv, const u32* __restrict__ cumy,
                                                     u32* __restrict__ scratch, double* __restrict__ out, u32 n_paths, const u32* __restrict__ list) {
    __shared__ u32 s_n[2];
    __shared__ double s_best[GASM_WG / 64];
    u32* const cnt = scratch + (u64)blockIdx.x * 2 * n_table;       // per-row counts of the path
    u32* const srt = cnt + n_table;                                  // its non-zero counts, ascending
    const u32 ny = n_table ? cumy[n_table - 1] : 0u;
    for (u32 pi = blockIdx.x; pi < n_paths; pi += gridDim.x) {
        const u32 p = list ? list[pi] : pi;                          // (list: the paths k_path_ks2 left to this kernel)
        const u64 pb = ps.p_off[p];
        const u32 len = (u32)(ps.p_off[p + 1] - pb);
        for (u32 i = threadIdx.x; i < n_table; i += GASM_WG) cnt[i] = 0;
        if (threadIdx.x == 0) { s_n[0] = 0; s_n[1] = 0; }
        __syncthreads();
        u32 tot_local = 0;
        for (u32 j = threadIdx.x; j < len; j += GASM_WG) {
            const u32 c = poscnt[pb + j];
            if (!c) continue;
            tot_local += c;
            u32 idx;
            if (break_window(ps.words, pb, len, j, kmer, &idx)) {
                const int32_t row = drow[idx];
                if (row >= 0) atomicAdd(&cnt[row], c);
            }
        }
        atomicAdd(&s_n[1], tot_local);
        __syncthreads();
        const u32 total = s_n[1];
        // ---- the non-zero counts, compacted and sorted (bitonic, in the workgroup's scratch: L2-resident)
        for (u32 i = threadIdx.x; i < n_table; i += GASM_WG) {
            const u32 c = cnt[i];
            if (c) srt[atomicAdd(&s_n[0], 1u)] = c;
        }
        __syncthreads();
        const u32 nnz = s_n[0];
        u32 p2 = 1;
        while (p2 < nnz) p2 <<= 1;
        for (u32 i = nnz + threadIdx.x; i < p2; i += GASM_WG) srt[i] = 0xFFFFFFFFu;
        __syncthreads();
        for (u32 kk = 2; kk <= p2; kk <<= 1) {
            for (u32 jj = kk >> 1; jj > 0; jj >>= 1) {
                for (u32 t = threadIdx.x; t < (p2 >> 1); t += GASM_WG) {
                    const u32 lo = ((t & ~(jj - 1)) << 1) | (t & (jj - 1)), hi = lo | jj;
                    const u32 a = srt[lo], b = srt[hi];
                    const bool up = (lo & kk) == 0;
                    if ((a > b) == up) { srt[lo] = b; srt[hi] = a; }
                }
                __syncthreads();
            }
        }
        double best = 0.0;
        if (total && ny && n_table) {
            const double dnx = (double)n_table, dny = (double)ny, dtot = (double)total;
            const u32 nzero = n_table - nnz;
            // values of the path's sample: 0 (if any row is empty) and every distinct count / total
            if (threadIdx.x == 0 && nzero) {
                const u32 ry = ks_rank_le(pv, n_table, 0.0);
                best = fabs((double)nzero / dnx - (ry ? (double)cumy[ry - 1] : 0.0) / dny);
            }
            for (u32 i = threadIdx.x; i < nnz; i += GASM_WG) {
                if (i + 1 < nnz && srt[i + 1] == srt[i]) continue;         // last of a run of ties
                const double t = (double)srt[i] / dtot;
                const u32 ry = ks_rank_le(pv, n_table, t);
                const double d = fabs((double)(nzero + i + 1) / dnx - (ry ? (double)cumy[ry - 1] : 0.0) / dny);
                best = d > best ? d : best;
            }
            // values of the genome's sample: every distinct probability
            for (u32 r = threadIdx.x; r < n_table; r += GASM_WG) {
                if (r + 1 < n_table && pv[r + 1] == pv[r]) continue;
                const double t = pv[r];
                u32 lo = 0, hi = nnz;                                      // #{non-zero x <= t}, x = count / total in double
                while (lo < hi) { const u32 m = (lo + hi) >> 1; if ((double)srt[m] / dtot <= t) lo = m + 1; else hi = m; }
                const u32 cx = (t >= 0.0 ? nzero : 0u) + lo;
                const double d = fabs((double)cx / dnx - (double)cumy[r] / dny);
                best = d > best ? d : best;
            }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { const double o = __shfl_xor(best, d, 64); best = o > best ? o : best; }
        if ((threadIdx.x & 63) == 0) s_best[threadIdx.x >> 6] = best;
        __syncthreads();
        if (threadIdx.x == 0) {
            double b = s_best[0];
            for (int w = 1; w < GASM_WG / 64; ++w) b = s_best[w] > b ? s_best[w] : b;
            out[p] = (total && ny && n_table) ? b : __builtin_nan("");      // nothing matched: path_freq is all NaN, R has no x
        }
        __syncthreads();
    }
}

// #{r : pv[r] < t}
__device__ __forceinline__ u32 ks_rank_lt(const double* __restrict__ pv, u32 n, double t) {
    u32 lo = 0, hi = n;
    while (lo < hi) { const u32 m = (lo + hi) >> 1; if (pv[m] < t) lo = m + 1; else hi = m; }
    return lo;
}

// Round 3: the same statistic without a pass over the table per path.  k_path_ks zeroes, compacts and walks all n_table
// (~70 k) rows for every path and sorts the non-zero counts; here
//   * the rows a path touches are listed as they are first touched (the atomic add that finds a zero) and reset from
//     that list, so the per-workgroup counts stay zero between paths;
//   * the counts are small integers: a histogram over their VALUES (KS_BINS bins in LDS) replaces the sort — its prefix
//     sums are the ranks, its non-empty bins the distinct sample values (a path with a count >= KS_BINS is left to
//     k_path_ks: flags[p] = 1);
//   * of the genome's sample values only those next to a value of the path's sample can attain the supremum: between two
//     consecutive values of the path's sample Fx is constant and Fy is monotone, so |Fx - Fy| peaks at the first and the
//     last genome value of the interval (run_end[r]: the last index of the run of equal probabilities that holds r).
// Same candidates' maximum as k_path_ks, hence the same double.
#define KS_BINS 1024
__global__ void __launch_bounds__(GASM_WG) k_path_ks2(PathSet ps, const u32* __restrict__ poscnt, const int32_t* __restrict__ drow, int kmer,
                                                      u32 n_table, const double* __restrict__ pv, const u32* __restrict__ cumy, const u32* __restrict__ run_end,
                                                      u32* __restrict__ scratch, double* __restrict__ out, u32* __restrict__ flags, u32 n_paths, u32 bins) {
    __shared__ u32 s_hist[KS_BINS];          // bin c: rows with count c; then the inclusive prefix (count | non-empty << 20)
    __shared__ u32 s_dv[KS_BINS + 1], s_dc[KS_BINS + 1];     // the distinct values (counts) ascending and the ranks behind them
    __shared__ u32 s_n[4];
    __shared__ u32 s_w[GASM_WG / 64];
    __shared__ double s_best[GASM_WG / 64];
    u32* const cnt = scratch + (u64)blockIdx.x * 2 * n_table;       // per-row counts of the path: zero between paths
    u32* const touched = cnt + n_table;
    const u32 ny = n_table ? cumy[n_table - 1] : 0u;
    const u32 ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (u32 p = blockIdx.x; p < n_paths; p += gridDim.x) {
        const u64 pb = ps.p_off[p];
        const u32 len = (u32)(ps.p_off[p + 1] - pb);
        for (u32 i = threadIdx.x; i < KS_BINS; i += GASM_WG) s_hist[i] = 0;
        if (threadIdx.x < 4) s_n[threadIdx.x] = 0;
        __syncthreads();
        u32 tot_local = 0;
        for (u32 j = threadIdx.x; j < len; j += GASM_WG) {
            const u32 c = poscnt[pb + j];
            if (!c) continue;
            tot_local += c;
            u32 idx;
            if (break_window(ps.words, pb, len, j, kmer, &idx)) {
                const int32_t row = drow[idx];
                if (row >= 0 && atomicAdd(&cnt[row], c) == 0) touched[atomicAdd(&s_n[0], 1u)] = (u32)row;
            }
        }
        atomicAdd(&s_n[1], tot_local);
        __syncthreads();
        const u32 nnz = s_n[0], total = s_n[1];
        for (u32 i = threadIdx.x; i < nnz; i += GASM_WG) {
            const u32 row = touched[i], c = cnt[row];
            cnt[row] = 0;
            if (c < bins) atomicAdd(&s_hist[c], 1u);
            else s_n[2] = 1;
        }
        __syncthreads();
        if (s_n[2]) {                                  // a count beyond the histogram: the general kernel takes this path
            if (threadIdx.x == 0) { flags[p] = 1; out[p] = __builtin_nan(""); }
            __syncthreads();
            continue;
        }
        // ---- inclusive prefix over the bins, four per thread: rows with a count <= c, and how many bins up to c are non-empty
        {
            u32 v[KS_BINS / GASM_WG], run = 0;
#pragma unroll
            for (u32 q = 0; q < KS_BINS / GASM_WG; ++q) { const u32 h = s_hist[threadIdx.x * (KS_BINS / GASM_WG) + q]; run += h | (h ? 1u << 20 : 0u); v[q] = run; }
            const u32 inc = wave_incl_scan(run);
            if (ln == 63) s_w[wv] = inc;
            __syncthreads();
            u32 before = inc - run;
            for (u32 w = 0; w < wv; ++w) before += s_w[w];
#pragma unroll
            for (u32 q = 0; q < KS_BINS / GASM_WG; ++q) {
                const u32 c = threadIdx.x * (KS_BINS / GASM_WG) + q, incl = before + v[q];
                if (s_hist[c]) { const u32 k = (incl >> 20) - 1; s_dv[k] = c; s_dc[k] = incl & 0xFFFFFu; }
            }
            if (threadIdx.x == GASM_WG - 1) s_n[3] = (before + run) >> 20;        // distinct non-zero values
        }
        __syncthreads();
        const u32 nd = s_n[3];
        double best = 0.0;
        if (total && ny && n_table) {
            const double dnx = (double)n_table, dny = (double)ny, dtot = (double)total;
            const u32 nzero = n_table - nnz;
            auto fy = [&](u32 r) { return (double)cumy[r] / dny; };
            auto upd = [&](double d) { best = d > best ? d : best; };
            // item -1: the sample value 0 (rows the path never touches), item k >= 0: count s_dv[k]
            for (u32 it = threadIdx.x; it < nd + 1; it += GASM_WG) {
                const bool zero_item = it == 0;
                if (zero_item && !nzero) {
                    // no value 0: the genome's values below the path's smallest value see Fx = 0
                    const u32 hi = nd ? ks_rank_lt(pv, n_table, (double)s_dv[0] / dtot) : n_table;
                    if (hi) { upd(fy(run_end[0])); upd(fy(hi - 1)); }
                    continue;
                }
                const double v = zero_item ? 0.0 : (double)s_dv[it - 1] / dtot;
                const double fx = (double)(nzero + (zero_item ? 0u : s_dc[it - 1])) / dnx;
                const u32 ry = ks_rank_le(pv, n_table, v);
                upd(fabs(fx - (ry ? (double)cumy[ry - 1] : 0.0) / dny));                   // at the path's own value
                const u32 lo = ks_rank_lt(pv, n_table, v), hi = it < nd ? ks_rank_lt(pv, n_table, (double)s_dv[it] / dtot) : n_table;
                if (lo < hi) { upd(fabs(fx - fy(run_end[lo]))); upd(fabs(fx - fy(hi - 1))); }    // the genome's values in [v, next value)
                if (zero_item && lo) { upd(fy(run_end[0])); upd(fy(lo - 1)); }                   // (probabilities below 0: Fx = 0 there)
            }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { const double o = __shfl_xor(best, d, 64); best = o > best ? o : best; }
        if (ln == 0) s_best[wv] = best;
        __syncthreads();
        if (threadIdx.x == 0) {
            double b = s_best[0];
            for (int w = 1; w < GASM_WG / 64; ++w) b = s_best[w] > b ? s_best[w] : b;
            out[p] = (total && ny && n_table) ? b : __builtin_nan("");      // nothing matched: path_freq is all NaN, R has no x
        }
        __syncthreads();
    }
}

// Coverage: diff[a] += 1, diff[b + 1] -= 1 for the inclusive ranges [a, b] clipped to [1, seq_len]; a scan and a count of
// the positive positions follow (k_cover_count: one workgroup, running carry).
__global__ void __launch_bounds__(GASM_WG) k_cover_mark(const long long* __restrict__ start, const long long* __restrict__ len, u64 n, long long seq_len,
                                                        int* __restrict__ diff) {
    const u64 i = (u64)blockIdx.x * GASM_WG + threadIdx.x;
    if (i >= n) return;
    long long a = start[i], b = start[i] + len[i];
    if (a < 1) a = 1;
    if (b > seq_len) b = seq_len;
    if (a > b) return;
    atomicAdd(&diff[a], 1);
    atomicAdd(&diff[b + 1], -1);
}
__global__ void __launch_bounds__(1024) k_cover_count(const int* __restrict__ diff, long long seq_len, unsigned long long* __restrict__ covered) {
    __shared__ long long s_w[16];
    __shared__ unsigned long long s_c[16];
    const u32 ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
    long long carry = 0;
    unsigned long long cov = 0;
    for (long long base = 1; base <= seq_len; base += 1024) {
        const long long i = base + threadIdx.x;
        const int v = i <= seq_len ? diff[i] : 0;
        long long inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const long long o = __shfl_up(inc, d, 64); if ((int)ln >= d) inc += o; }
        if (ln == 63) s_w[wv] = inc;
        __syncthreads();
        long long before = 0, tot = 0;
        for (u32 w = 0; w < 16; ++w) { if (w < wv) before += s_w[w]; tot += s_w[w]; }
        __syncthreads();
        if (i <= seq_len && carry + before + inc > 0) ++cov;
        carry += tot;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) cov += __shfl_xor(cov, d, 64);
    if (ln == 0) s_c[wv] = cov;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long t = 0; for (u32 w = 0; w < 16; ++w) t += s_c[w]; *covered = t; }
}
