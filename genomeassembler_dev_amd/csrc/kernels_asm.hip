// kernels_asm.hip — scaffolds of assemble_contigs kept on the device (lib/DeNovoAssembler.cpp:228-304,
// lib/BreakageScorer.cpp:85-172).  A scaffold is a chain of whole contigs glued with overlaps (host_algos.cpp: the greedy
// merge runs on contig indices); here the chains become 2-bit sequences in HBM, are ordered and de-duplicated as strings
// (the reference's sort + unique on std::string: plain lexicographic, a proper prefix first), and stay there for
// calc_breakscore — the 3.8e8 characters of a 50 kb experiment never exist as host strings unless somebody fetches them.
//   k_chain_expand    chains -> one packed base stream (scaffold after scaffold, no gaps): a thread per output word
//   k_str_bitonic     one compare-exchange step of a bitonic sort of scaffold indices, comparing packed sequences
//   k_str_adjacent_eq equal neighbours in the sorted order (std::unique)
//   k_unpack_ascii    2-bit -> text, for a fetch
#include "device_utils.h"
#include "kernels.h"

__device__ __forceinline__ u32 base_at(const u64* __restrict__ w, u64 p) { return (u32)(w[p >> 5] >> (62 - 2 * (p & 31))) & 3u; }

// Scaffold m = elements [sig_off[m], sig_off[m+1]); element e contributes contig elem_contig[e] without its first
// elem_skip[e] bases, starting at base elem_pos[e] of the scaffold; scaffold m occupies bases [out_off[m], out_off[m+1]).
__global__ void __launch_bounds__(GASM_WG) k_chain_expand(const u64* __restrict__ cwords, const u64* __restrict__ c_off, const u64* __restrict__ sig_off,
                                                          const u32* __restrict__ elem_contig, const u32* __restrict__ elem_skip,
                                                          const u64* __restrict__ elem_pos, const u64* __restrict__ out_off, u32 n_scaffolds,
                                                          u64* __restrict__ out, u64 n_words) {
    const u64 total = out_off[n_scaffolds];
    for (u64 w = (u64)blockIdx.x * GASM_WG + threadIdx.x; w < n_words; w += (u64)gridDim.x * GASM_WG) {
        u64 g = w << 5, v = 0;
        if (g < total) {
            u32 m = upper_seg<u64>(out_off, n_scaffolds, g);              // scaffold of the word's first base
            while (out_off[m + 1] <= g) ++m;                               // (empty scaffolds share an offset)
            u64 e0 = sig_off[m], e1 = sig_off[m + 1];
            u64 rel = g - out_off[m];
            u64 e = e0 + upper_seg<u64>(elem_pos + e0, (u32)(e1 - e0), rel);
            for (u32 b = 0; b < 32 && g < total; ++b, ++g) {
                while (g >= out_off[m + 1]) { ++m; e0 = sig_off[m]; e1 = sig_off[m + 1]; e = e0; }
                rel = g - out_off[m];
                while (e + 1 < e1 && elem_pos[e + 1] <= rel) ++e;
                const u64 src = c_off[elem_contig[e]] + elem_skip[e] + (rel - elem_pos[e]);
                v |= (u64)base_at(cwords, src) << (62 - 2 * b);
            }
        }
        out[w] = v;
    }
}

// sequence a < sequence b (lexicographic over bases, a proper prefix first); *eq = equal
__device__ __forceinline__ bool seq_less(const u64* __restrict__ words, u64 oa, u64 la, u64 ob, u64 lb, bool* eq) {
    const u64 n = la < lb ? la : lb;
    for (u64 t = 0; t < n; t += 32) {
        u64 va = window32(words, oa + t), vb = window32(words, ob + t);
        const u64 left = n - t;
        if (left < 32) { const u64 mk = ~0ull << (64 - 2 * left); va &= mk; vb &= mk; }
        if (va != vb) { *eq = false; return va < vb; }
    }
    *eq = la == lb;
    return la < lb;
}

// Scaffolds are chains of whole contigs (signature = contig, (overlap, contig)*): two of them that begin with the same
// elements are the same string up to where the first differing element starts — so the comparison walks the signatures (a
// handful of 32-bit words) and only then the bases, from that point on.  Scaffolds of one experiment share thousands of
// leading bases (52 contigs, 22 053 scaffolds): the plain word-by-word comparison spent its time re-reading them, 120 passes
// of the bitonic network long (14.5 ms -> see DESIGN.md §6).  sig = nullptr: plain strings.
__device__ __forceinline__ u64 chain_common(const ChainSigs& cs, u32 a, u32 b, u64 la, u64 lb) {
    if (!cs.sig_off) return 0;
    const u64 sa = cs.sig_off[a], sb = cs.sig_off[b];
    const u32 na = (u32)(cs.sig_off[a + 1] - sa), nb = (u32)(cs.sig_off[b + 1] - sb), n = na < nb ? na : nb;
    u32 e = 0;
    while (e < n && cs.elem_contig[sa + e] == cs.elem_contig[sb + e] && cs.elem_skip[sa + e] == cs.elem_skip[sb + e]) ++e;
    return e < n ? cs.elem_pos[sa + e] : (la < lb ? la : lb);
}
__device__ __forceinline__ bool chain_less(const u64* __restrict__ words, const u64* __restrict__ off, const ChainSigs& cs, u32 a, u32 b, bool* eq) {
    const u64 oa = off[a], la = off[a + 1] - oa, ob = off[b], lb = off[b + 1] - ob;
    const u64 P = chain_common(cs, a, b, la, lb);
    return seq_less(words, oa + P, la - P, ob + P, lb - P, eq);
}

__global__ void __launch_bounds__(GASM_WG) k_str_bitonic(const u64* __restrict__ words, const u64* __restrict__ off, u32* __restrict__ idx, u32 n_pow2,
                                                         u32 kk, u32 j, ChainSigs cs) {
    const u32 t = blockIdx.x * GASM_WG + threadIdx.x;
    if (t >= (n_pow2 >> 1)) return;
    const u32 lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
    const u32 a = idx[lo], b = idx[hi];
    bool swap;
    if (a == GASM_NONE32 || b == GASM_NONE32) swap = (a == GASM_NONE32 && b != GASM_NONE32);      // padding sorts last
    else {
        bool eq;
        const bool b_less = chain_less(words, off, cs, b, a, &eq);
        swap = b_less || (eq && b < a);            // equal strings: by index (any fixed order will do: they are merged)
    }
    const bool up = (lo & kk) == 0;
    if (swap == up) { idx[lo] = b; idx[hi] = a; }
}

// the steps j0, j0 / 2, ... 1 of stage kk (j0 <= GASM_WG): every workgroup owns 2 * GASM_WG consecutive positions, keeps their
// indices in LDS and runs the steps with a barrier in between
__global__ void __launch_bounds__(GASM_WG) k_str_bitonic_block(const u64* __restrict__ words, const u64* __restrict__ off, u32* __restrict__ idx, u32 n_pow2,
                                                               u32 kk, u32 j0, ChainSigs cs) {
    __shared__ u32 s_idx[2 * GASM_WG];
    const u32 base = blockIdx.x * 2 * GASM_WG;
    for (u32 q = threadIdx.x; q < 2 * GASM_WG; q += GASM_WG) s_idx[q] = base + q < n_pow2 ? idx[base + q] : GASM_NONE32;
    __syncthreads();
    for (u32 j = j0; j > 0; j >>= 1) {
        const u32 t = threadIdx.x;
        const u32 lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
        const u32 a = s_idx[lo], b = s_idx[hi];
        bool swap;
        if (a == GASM_NONE32 || b == GASM_NONE32) swap = (a == GASM_NONE32 && b != GASM_NONE32);
        else {
            bool eq;
            const bool b_less = chain_less(words, off, cs, b, a, &eq);
            swap = b_less || (eq && b < a);
        }
        const bool up = ((base + lo) & kk) == 0;
        if (swap == up) { s_idx[lo] = b; s_idx[hi] = a; }
        __syncthreads();
    }
    for (u32 q = threadIdx.x; q < 2 * GASM_WG; q += GASM_WG) if (base + q < n_pow2) idx[base + q] = s_idx[q];
}

__global__ void __launch_bounds__(GASM_WG) k_str_adjacent_eq(const u64* __restrict__ words, const u64* __restrict__ off, const u32* __restrict__ idx, u32 n,
                                                             u8* __restrict__ same_as_prev, ChainSigs cs) {
    const u32 i = blockIdx.x * GASM_WG + threadIdx.x;
    if (i >= n) return;
    bool eq = false;
    if (i) { const u32 a = idx[i - 1], b = idx[i]; (void)chain_less(words, off, cs, a, b, &eq); }
    same_as_prev[i] = eq ? 1 : 0;
}

__global__ void __launch_bounds__(GASM_WG) k_unpack_ascii(const u64* __restrict__ words, u64 nbases, u8* __restrict__ out) {
    for (u64 p = (u64)blockIdx.x * GASM_WG + threadIdx.x; p < nbases; p += (u64)gridDim.x * GASM_WG) out[p] = "ACGT"[base_at(words, p)];
}

// ================================================================================================================
// The greedy merge itself (lib/DeNovoAssembler.cpp:233-266; lib/BreakageScorer.cpp:96-149), on contig indices as in
// host_algos.cpp (merge_indices): a merged string is a chain of whole contigs, and "suffix(c_i, ov) == prefix(c_j, ov)" is a
// statement about the chain's LAST contig and the other chain's FIRST contig only, so the tests of all (overlap, contig,
// contig) triples are made once (k_asm_match) and a permutation's merge is the reference's loop over small integers.
// One wave per permutation: the outer loops (overlap, repeat-until-stable, i) run as the reference writes them; the inner
// scan "for j from the back, the first chain whose head matches my tail" is 64 candidates per step with a ballot; the
// chain table lives in LDS.  Two chains of equal length whose tail/head match need the reference's full-string test
// `c[i] != c[j]`: made on the device too (chains_equal walks both chains through the packed contigs) — until round 3 such a
// permutation went back to the host routine, and with 52 contigs two of equal length meet in two rows out of three.
// ================================================================================================================
__global__ void __launch_bounds__(GASM_WG) k_asm_match(const u64* __restrict__ cwords, const u64* __restrict__ c_off, u32 n, int k,
                                                       u8* __restrict__ match, u8* __restrict__ row_any, u8* __restrict__ level_any) {
    const u64 total = (u64)(k - 1) * n * n;
    for (u64 t = (u64)blockIdx.x * GASM_WG + threadIdx.x; t < total; t += (u64)gridDim.x * GASM_WG) {
        const u32 b = (u32)(t % n), a = (u32)((t / n) % n);
        const int ov = 1 + (int)(t / ((u64)n * n));
        const u64 pa = c_off[a + 1] - ov, pb = c_off[b];         // (every contig has at least k-1 >= ov bases: host precondition)
        bool eq = true;
        for (int o = 0; o < ov && eq; o += 32) {
            u64 va = window32(cwords, pa + o), vb = window32(cwords, pb + o);
            const int left = ov - o;
            if (left < 32) { const u64 mk = ~0ull << (64 - 2 * left); va &= mk; vb &= mk; }
            eq = va == vb;
        }
        match[((u64)ov * n + a) * n + b] = eq ? 1 : 0;
        if (eq && a != b) { row_any[(u64)ov * n + a] = 1; level_any[ov] = 1; }
    }
}

// Are the strings of two chains of equal length the same?  (lib/DeNovoAssembler.cpp:238: `contigs[i] != contigs[j]` — at equal
// length only the full comparison tells.)  A chain is its head contig followed by every next contig minus the overlap; both
// are walked in pieces of up to 32 bases.  Every lane runs the same loop (uniform control flow, uniform loads); chains of
// distinct contigs differ within the first word almost always.
__device__ __forceinline__ bool chains_equal(const u64* __restrict__ cwords, const u64* __restrict__ c_off, const u32* next, const u8* ovl, u32 ha, u32 hb, u32 len) {
    u32 ca = ha, cb = hb;
    u64 pa = c_off[ca], ea = c_off[ca + 1], pb = c_off[cb], eb = c_off[cb + 1];
    u32 left = len;
    while (left) {
        while (pa == ea) { const u32 o = ovl[ca]; ca = next[ca]; if (ca == GASM_NONE32) return true; pa = c_off[ca] + o; ea = c_off[ca + 1]; }
        while (pb == eb) { const u32 o = ovl[cb]; cb = next[cb]; if (cb == GASM_NONE32) return true; pb = c_off[cb] + o; eb = c_off[cb + 1]; }
        const u32 m = (u32)min(min(ea - pa, eb - pb), (u64)min(32u, left));
        const u64 mk = m == 32 ? ~0ull : ~0ull << (64 - 2 * m);
        if ((window32(cwords, pa) & mk) != (window32(cwords, pb) & mk)) return false;
        pa += m; pb += m; left -= m;
    }
    return true;
}

__global__ void __launch_bounds__(64) k_asm_merge(const u32* __restrict__ perm, u32 rows, u32 n, int k, const u32* __restrict__ clen,
                                                  const u8* __restrict__ match, const u8* __restrict__ row_any, const u8* __restrict__ level_any,
                                                  const u64* __restrict__ cwords, const u64* __restrict__ c_off,
                                                  u32* out_next, u8* out_ov, u32* __restrict__ out_heads,
                                                  u32* __restrict__ out_nchains, u8* __restrict__ need_host) {
    extern __shared__ u32 sm[];
    u32* head = sm;
    u32* tail = sm + n;
    u32* len = sm + 2 * n;
    u8* emp = reinterpret_cast<u8*>(sm + 3 * n);
    const u32 lane = threadIdx.x;
    for (u32 r = blockIdx.x; r < rows; r += gridDim.x) {
        const u64 rb = (u64)r * n;
        for (u32 p = lane; p < n; p += 64) {
            const u32 c = perm[rb + p];
            head[p] = c; tail[p] = c; len[p] = clen[c]; emp[p] = 0;
            out_next[rb + c] = GASM_NONE32; out_ov[rb + c] = 0;
        }
        __syncthreads();
        u32 m = n;
        bool bad = false;
        for (int ov = k - 1; ov > 0 && !bad; --ov) {
            if (!level_any[ov]) continue;            // nothing can merge at this overlap: the reference's pass changes nothing
            const u8* M = match + (u64)ov * n * n;
            const u8* RA = row_any + (u64)ov * n;
            bool shrunk = true;
            while (shrunk && !bad) {
                const u32 before = m;
                for (u32 i = 0; i < m && !bad; ++i) {
                    if (emp[i]) continue;
                    u32 ti = tail[i];
                    if (!RA[ti]) continue;
                    int jtop = (int)m - 1;
                    while (jtop >= 0) {
                        const int jj = jtop - (int)lane;
                        const bool ok = jj >= 0 && jj != (int)i && !emp[jj] && M[(u64)ti * n + head[jj]];
                        const unsigned long long mask = __ballot(ok);
                        if (!mask) { jtop -= 64; continue; }
                        const int j = jtop - (__ffsll((long long)mask) - 1);      // the highest matching position
                        // c[i] == c[j] is possible at equal length only: the reference's full-string test, on the chains as they
                        // stand (round 2 handed such a permutation back to the host: 68 % of the rows of a 50 kb experiment)
                        if (len[i] == len[j] && chains_equal(cwords, c_off, out_next + rb, out_ov + rb, head[i], head[j], len[i])) { jtop = j - 1; continue; }
                        __syncthreads();                                          // (everybody has read the old state)
                        if (lane == 0) {
                            out_next[rb + ti] = head[j];
                            out_ov[rb + ti] = (u8)ov;
                            tail[i] = tail[j];
                            len[i] += len[j] - (u32)ov;
                            emp[j] = 1;
                        }
                        __syncthreads();
                        ti = tail[i];
                        if (!RA[ti]) break;           // the new tail matches nothing: the rest of the scan is idle
                        jtop = j - 1;
                    }
                }
                // order-preserving compaction of the chains that are left
                u32 w = 0;
                for (u32 base = 0; base < m; base += 64) {
                    const u32 p = base + lane;
                    const bool keep = p < m && !emp[p];
                    const u32 h = keep ? head[p] : 0, t = keep ? tail[p] : 0, l = keep ? len[p] : 0;
                    const unsigned long long mask = __ballot(keep);
                    const u32 pos = w + (u32)__popcll(mask & ((1ull << lane) - 1ull));
                    __syncthreads();
                    if (keep) { head[pos] = h; tail[pos] = t; len[pos] = l; }
                    w += (u32)__popcll(mask);
                    __syncthreads();
                }
                for (u32 p = lane; p < w; p += 64) emp[p] = 0;
                __syncthreads();
                m = w;
                shrunk = before != m;
            }
        }
        for (u32 p = lane; p < m; p += 64) out_heads[rb + p] = head[p];
        if (lane == 0) { out_nchains[r] = m; need_host[r] = bad ? 1 : 0; }
        __syncthreads();
    }
}

// ================================================================================================================
// Row A16 — breakage-score-guided traversal (BASELINE configs[4]'s "combined" mode).  The reference does not have it
// (README.md:83); the specification is this project's, restated on the CPU by oracle/guided_oracle.py:
// contigs end at branching nodes; the guided traversal chains them through those nodes, steered by the breakage score:
//   score of a contig = its fixed-point breakage sum / its length, compared as an exact rational (cross-multiplied in 128
//   bits: no rounding, so the CPU restatement makes the same choices);
//   seeds in order of descending score (ties: smaller contig index = lexicographically smaller contig); a seed is
//   extended to the right — among the unused contigs whose first k-1 bases are the path's last k-1 bases the best-scoring
//   one, again and again — then to the left the same way; every contig ends up in exactly one guided scaffold.
// One wave per segment; candidates are examined 64 at a time, the best found by a butterfly reduction.
// ================================================================================================================
struct GBest { unsigned long long fs; u32 len; u32 idx; };
__device__ __forceinline__ bool gbest_better(const GBest& a, const GBest& b) {          // a before b?
    if (b.idx == GASM_NONE32) return a.idx != GASM_NONE32;
    if (a.idx == GASM_NONE32) return false;
    // a.fs / a.len > b.fs / b.len  <=>  a.fs * b.len > b.fs * a.len, in 128 bits
    const unsigned __int128 pa = (unsigned __int128)a.fs * b.len, pb = (unsigned __int128)b.fs * a.len;
    if (pa != pb) return pa > pb;
    return a.idx < b.idx;
}
// b = c if c is better — field by field with selects, NOT `if (gbest_better(c, b)) b = c;`: for that form hipcc (ROCm 7.2,
// gfx950, -O3) generated code in k_guided_chain's candidate loop that updated b.fs and b.idx but left b.len at the length of the
// lane's FIRST candidate (ISA: `v_mov_b32 v13, v16` on every comparing lane after `; implicit-def: $vgpr13`), so ratios were taken
// with the wrong denominator and the traversal chose wrong seeds (found by tools/soak.py seed 91; tools/micro/guided_repro.hip is
// the stand-alone reproducer; an emulation of exactly that defect reproduces the old kernel's output on the kept case).
__device__ __forceinline__ void gbest_take(GBest& b, const GBest& c) {
    const bool take = gbest_better(c, b);
    b.fs = take ? c.fs : b.fs;
    b.len = take ? c.len : b.len;
    b.idx = take ? c.idx : b.idx;
}
__device__ __forceinline__ GBest gbest_wave(GBest v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        GBest o;
        o.fs = __shfl_xor(v.fs, d, 64); o.len = __shfl_xor(v.len, d, 64); o.idx = __shfl_xor(v.idx, d, 64);
        gbest_take(v, o);
    }
    return v;
}
// the k1 bases at p and at q equal?  (k1 <= 62)
__device__ __forceinline__ bool bases_eq_short(const u64* __restrict__ w, u64 p, u64 q, int k1) {
    for (int o = 0; o < k1; o += 32) {
        u64 a = window32(w, p + o), b = window32(w, q + o);
        const int left = k1 - o;
        if (left < 32) { const u64 mk = ~0ull << (64 - 2 * left); a &= mk; b &= mk; }
        if (a != b) return false;
    }
    return true;
}

__global__ void __launch_bounds__(64) k_guided_chain(PathSet ps, const unsigned long long* __restrict__ fx, int k, u32* __restrict__ g_next,
                                                     u32* __restrict__ g_prev, u32* __restrict__ dbg_order) {
    extern __shared__ u8 s_used[];
    const u32 seg = blockIdx.x, lane = threadIdx.x;
    const u32 c0 = ps.seg_path_off[seg], n = ps.seg_path_off[seg + 1] - c0;
    for (u32 i = lane; i < n; i += 64) { s_used[i] = 0; g_next[c0 + i] = GASM_NONE32; g_prev[c0 + i] = GASM_NONE32; }
    __syncthreads();
    const int k1 = k - 1;
    auto best_of = [&](int mode, u32 cur) {      // mode 0: any unused; 1: unused that can follow cur; 2: unused that cur can follow
        GBest b{0ull, 1u, GASM_NONE32};
        const u64 cb = mode ? ps.p_off[c0 + cur] : 0, ce = mode ? ps.p_off[c0 + cur + 1] : 0;
        for (u32 j = lane; j < n; j += 64) {
            if (s_used[j]) continue;
            const u64 jb = ps.p_off[c0 + j], je = ps.p_off[c0 + j + 1];
            if (mode == 1 && !bases_eq_short(ps.words, ce - k1, jb, k1)) continue;
            if (mode == 2 && !bases_eq_short(ps.words, je - k1, cb, k1)) continue;
            const GBest c{fx[c0 + j], (u32)(je - jb), j};
            gbest_take(b, c);
        }
        return gbest_wave(b);
    };
    for (;;) {
        const GBest seed = best_of(0, 0);
        if (seed.idx == GASM_NONE32) break;
        if (lane == 0) { s_used[seed.idx] = 1; if (dbg_order) dbg_order[c0 + seed.idx] = (atomicAdd(&dbg_order[ps.seg_path_off[ps.n_segments]], 1u) << 2); }
        __syncthreads();
        for (int dir = 1; dir <= 2; ++dir) {
            u32 cur = seed.idx;
            for (;;) {
                const GBest nx = best_of(dir, cur);
                if (nx.idx == GASM_NONE32) break;
                if (lane == 0) {
                    s_used[nx.idx] = 1;
                    if (dbg_order) dbg_order[c0 + nx.idx] = (atomicAdd(&dbg_order[ps.seg_path_off[ps.n_segments]], 1u) << 2) | (u32)dir;
                    if (dir == 1) { g_next[c0 + cur] = c0 + nx.idx; g_prev[c0 + nx.idx] = c0 + cur; }
                    else { g_next[c0 + nx.idx] = c0 + cur; g_prev[c0 + cur] = c0 + nx.idx; }
                }
                __syncthreads();
                cur = nx.idx;
            }
        }
    }
}
