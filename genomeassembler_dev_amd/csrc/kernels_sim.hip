// kernels_sim.hip — the read simulator (lib/GenerateReads.R:235-313): per segment
// n = ceil(coverage * L / read_len) start positions, drawn with replacement with the weight of position p = the breakage
// probability of the kmer-long window starting at p (:243-259, :302-308), starts whose read would run past the end of the
// genome dropped (:310-313), reads = substrings of the genome, forward strand, no errors (:375-379).
// R's sample() / set.seed stream cannot be reproduced without R.  What is pinned instead (oracle/gasm_oracle.cpp restates
// it line by line, parity bit-exact): integer weights round(prob * 2^52), their running sums in 64 bits, draw d of segment
// s = splitmix64 of a counter, scaled to [0, total) by a 64 x 64 -> high 64 multiply, start = first position whose running
// sum exceeds the draw.  Everything stays on the device: genome -> weights -> scan -> draws -> kept starts -> packed reads.
#include "device_utils.h"
#include "kernels.h"

#define GASM_DIRECT_BASE(L) (((1u << (2 * (L))) - 4u) / 3u)

__device__ __forceinline__ u64 sim_mix64(u64 z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// weight of every start position of every segment: w[woff[s] + p], p in [0, L_s - kmer]; fixw = table in fixed point
// (direct-address over the kmer-long windows) or nullptr = every position weighs 1
__global__ void __launch_bounds__(GASM_WG) k_sim_weights(const u64* __restrict__ gwords, const u64* __restrict__ gbase, const u64* __restrict__ woff,
                                                         u32 n_segments, int kmer, const long long* __restrict__ fixw, u64* __restrict__ w) {
    const u32 s = blockIdx.y;
    const u64 n = woff[s + 1] - woff[s];
    for (u64 p = (u64)blockIdx.x * GASM_WG + threadIdx.x; p < n; p += (u64)gridDim.x * GASM_WG) {
        u64 v = 1;
        if (fixw) v = (u64)fixw[GASM_DIRECT_BASE((u32)kmer) + (u32)kmer_at(gwords, gbase[s] + p, kmer)];
        w[woff[s] + p] = v;
    }
    (void)n_segments;
}

// inclusive scan of every segment's slice [off[s], off[s+1]) in place; one workgroup of 1024 per segment, running carry
template <class T>
__global__ void __launch_bounds__(1024) k_seg_scan_incl(T* __restrict__ a, const u64* __restrict__ off) {
    __shared__ T s_w[16];
    const u32 s = blockIdx.x, ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u64 lo = off[s], hi = off[s + 1];
    T carry = 0;
    for (u64 base = lo; base < hi; base += 1024) {
        const u64 i = base + threadIdx.x;
        const T v = i < hi ? a[i] : (T)0;
        T inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const T o = __shfl_up(inc, d, 64); if ((int)ln >= d) inc += o; }
        if (ln == 63) s_w[wv] = inc;
        __syncthreads();
        T before = 0, tot = 0;
#pragma unroll
        for (u32 q = 0; q < 16; ++q) { const T t = s_w[q]; if (q < wv) before += t; tot += t; }
        __syncthreads();
        if (i < hi) a[i] = carry + before + inc;
        carry += tot;
    }
}
template __global__ void k_seg_scan_incl<u64>(u64*, const u64*);
template __global__ void k_seg_scan_incl<u32>(u32*, const u64*);

// draw d of segment s -> start position; keep[.] = 1 when the read fits
__global__ void __launch_bounds__(GASM_WG) k_sim_draw(const u64* __restrict__ cum, const u64* __restrict__ woff, const u64* __restrict__ doff,
                                                      const u64* __restrict__ glen, u64 seed, u32 read_len, u32* __restrict__ start,
                                                      u32* __restrict__ keep) {
    const u32 s = blockIdx.y;
    const u64 nd = doff[s + 1] - doff[s], np = woff[s + 1] - woff[s];
    const u64* c = cum + woff[s];
    const u64 total = np ? c[np - 1] : 0;
    const u64 sseed = sim_mix64(seed ^ (0xD1B54A32D192ED03ull * (u64)(s + 1)));
    for (u64 d = (u64)blockIdx.x * GASM_WG + threadIdx.x; d < nd; d += (u64)gridDim.x * GASM_WG) {
        u32 st = 0, k = 0;
        if (total) {
            const u64 r = __umul64hi(sim_mix64(sseed + d), total);          // uniform in [0, total)
            u64 lo = 0, hi = np;                                            // first p with c[p] > r
            while (lo < hi) { const u64 m = (lo + hi) >> 1; if (c[m] > r) hi = m; else lo = m + 1; }
            st = (u32)lo;
            k = lo + read_len <= glen[s] ? 1u : 0u;
        }
        start[doff[s] + d] = st;
        keep[doff[s] + d] = k;
    }
}

// kept draws, in draw order, to their final place: rank[.] = inclusive scan of keep inside the segment
__global__ void __launch_bounds__(GASM_WG) k_sim_compact(const u32* __restrict__ start, const u32* __restrict__ keep, const u32* __restrict__ rank,
                                                         const u64* __restrict__ doff, const u64* __restrict__ seg_read_off, u32* __restrict__ kept_start) {
    const u32 s = blockIdx.y;
    const u64 nd = doff[s + 1] - doff[s];
    for (u64 d = (u64)blockIdx.x * GASM_WG + threadIdx.x; d < nd; d += (u64)gridDim.x * GASM_WG) {
        const u64 i = doff[s] + d;
        if (keep[i]) kept_start[seg_read_off[s] + rank[i] - 1] = start[i];
    }
}

// read r of segment s = genome[start, start + read_len), written at base r * read_len of the packed read stream (zeroed
// before): 32 bases at a time, each piece OR-ed into the one or two words it falls into
__global__ void __launch_bounds__(GASM_WG) k_sim_extract(const u64* __restrict__ gwords, const u64* __restrict__ gbase, const u64* __restrict__ seg_read_off,
                                                         const u32* __restrict__ kept_start, u32 read_len, unsigned long long* __restrict__ out) {
    const u32 s = blockIdx.y;
    const u64 r0 = seg_read_off[s], r1 = seg_read_off[s + 1];
    const u32 chunks = (read_len + 31) / 32;
    for (u64 t = (u64)blockIdx.x * GASM_WG + threadIdx.x; t < (r1 - r0) * chunks; t += (u64)gridDim.x * GASM_WG) {
        const u64 r = r0 + t / chunks;
        const u32 j = (u32)(t % chunks);
        const u32 nb = min(32u, read_len - 32 * j);
        u64 v = window32(gwords, gbase[s] + kept_start[r] + 32ull * j);
        if (nb < 32) v &= ~0ull << (64 - 2 * nb);
        const u64 d0 = r * read_len + 32ull * j;          // destination base
        const u32 sh = (u32)(d0 & 31) << 1;
        atomicOr(&out[d0 >> 5], v >> sh);
        if (sh) atomicOr(&out[(d0 >> 5) + 1], v << (64 - sh));
    }
}
