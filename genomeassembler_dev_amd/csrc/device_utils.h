// device_utils.h — device-side helpers shared by the kernels (gfx950, wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint16_t u16;
typedef uint8_t u8;

#define GASM_WG 256                    // threads per workgroup of the streaming kernels (4 waves)
#define GASM_EMPTY64 0xFFFFFFFFFFFFFFFFull
#define GASM_NONE32 0xFFFFFFFFu
#define GASM_LINK_DONE 0x80000000ull   // link word: ancestor << 32 | done << 31 | distance
#define GASM_LINK_TAG 0x40000000ull    // (between k_rank_rulers and k_link_jump only) ancestor = a ruler BEHIND this edge, distance = how far behind

// ----------------------------------------------------------------------------------------------------------------
// Packed base streams.  Base j of a stream lives in 64-bit word j>>5 at bit 62-2*(j&31) (first base most
// significant), A=0 C=1 G=2 T=3, so a window read as an integer compares like the string.  Every stream is followed
// by two zero padding words so a window may read one word past its last base.
// ----------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 window32(const u64* __restrict__ w, u64 p) {
    const u64 i = p >> 5;
    const u32 s = (u32)(p & 31) << 1;
    const u64 hi = w[i];
    const u64 lo = w[i + 1];
    // (lo >> 64) is undefined; split the shift so s == 0 yields hi
    return (hi << s) | ((lo >> 1) >> (63 - s));
}

// the k bases starting at p as a right-aligned 2k-bit integer, k <= 32
__device__ __forceinline__ u64 kmer_at(const u64* __restrict__ w, u64 p, int k) { return window32(w, p) >> (64 - 2 * k); }

__device__ __forceinline__ u32 base_code(u8 c) { return ((c >> 1) & 3u) ^ ((c >> 2) & 1u); }  // A0 C1 G2 T3
__device__ __forceinline__ bool base_ok(u8 c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

// f(integral_constant<u32, 0>) ... f(integral_constant<u32, N-1>): a loop whose index is a compile-time constant
template <class F, u32... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<u32, Is...>) { (f(std::integral_constant<u32, Is>{}), ...); }
template <u32 N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<u32, N>{}); }

__device__ __forceinline__ u32 hash64(u64 x) { return (u32)((x * 0x9E3779B97F4A7C15ull) >> 32); }

// ----------------------------------------------------------------------------------------------------------------
// Block-wide exclusive scan of one u32 per thread, GASM_WG (=256) threads = 4 waves.  s_tmp: >= 5 u32 of LDS.
// Returns the exclusive prefix; *total gets the block sum.  Contains two barriers.
// ----------------------------------------------------------------------------------------------------------------
// Inclusive scan across the 64 lanes in the VALU's DPP network (row shifts inside rows of 16, then row broadcasts):
// no LDS round trips, unlike __shfl_up (ds_bpermute).  Lanes without a source add 0.
__device__ __forceinline__ u32 wave_incl_scan(u32 v) {
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return v;
}
// the value of lane 63 (e.g. the total after wave_incl_scan), uniform
__device__ __forceinline__ u32 wave_last(u32 v) { return (u32)__builtin_amdgcn_readlane((int)v, 63); }

template <int NT>
__device__ __forceinline__ u32 block_excl_scan(u32 v, u32* s_tmp, u32* total) {
    constexpr int NW = NT / 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u32 inc = wave_incl_scan(v);
    if (lane == 63) s_tmp[wv] = inc;
    __syncthreads();
    u32 base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const u32 t = s_tmp[i];
        if (i < wv) base += t;
        tot += t;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// The same without the trailing barrier: for callers whose next write to s_tmp lies behind another barrier anyway.
template <int NT>
__device__ __forceinline__ u32 block_excl_scan_open(u32 v, u32* s_tmp, u32* total) {
    constexpr int NW = NT / 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u32 inc = wave_incl_scan(v);
    if (lane == 63) s_tmp[wv] = inc;
    __syncthreads();
    u32 base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const u32 t = s_tmp[i];
        if (i < wv) base += t;
        tot += t;
    }
    *total = tot;
    return base + inc - v;
}

// Segment-major grids, XCD-aware.  Workgroups are dealt round-robin to the 8 XCDs, each with its own L2 (observed
// behaviour, a speed assumption only), so the C workgroups of a segment get linear ids that are congruent mod 8:
// everything the segment's kernels gather from — its slice of the sorted k-mers, its links, its contigs — then lives
// in one L2 instead of being pulled into all eight.  Launch with grid = seg_grid(C, S) (host); false = padding workgroup.
__device__ __forceinline__ bool seg_chunk(u32 n_segments, u32 chunks, u32* seg, u32* chunk) {
    const u32 b = blockIdx.x, q = b >> 3;
    *chunk = q % chunks;
    *seg = (q / chunks) * 8u + (b & 7u);
    return *seg < n_segments;
}

// The edges of a segment, `chunks` workgroups per segment.  The launch takes `chunks` from an estimate of the largest
// segment (the host does not wait for the real sizes); a segment with more edges than chunks * GASM_WG is covered by
// further rounds of the same workgroups.  f(seg, lo, hi, i) for every edge i in [lo, hi) of the workgroup's segment.
template <class F>
__device__ __forceinline__ void for_seg_edges(const u32* __restrict__ dstart, u32 nb, u32 n_segments, u32 chunks, F&& f) {
    u32 seg, chunk;
    if (!seg_chunk(n_segments, chunks, &seg, &chunk)) return;
    const u32 lo = dstart[seg * nb], hi = dstart[(seg + 1) * nb];
    for (u32 base = chunk * GASM_WG; base < hi - lo; base += chunks * GASM_WG) {
        const u32 i = lo + base + threadIdx.x;
        if (i < hi) f(seg, lo, hi, i);
    }
}

// The same, GASM_EDGE_ILP edges per thread and round handed over together: these kernels are one or two dependent gathers
// per edge and live on loads in flight — with one edge per thread the 2048 threads of a CU are all the parallelism there is.
// f(seg, lo, hi, i[ILP], ok[ILP]); launch with chunks = ceil(edges / (GASM_WG * GASM_EDGE_ILP)).
#define GASM_EDGE_ILP 4
template <class F>
__device__ __forceinline__ void for_seg_edge_groups(const u32* __restrict__ dstart, u32 nb, u32 n_segments, u32 chunks, F&& f) {
    u32 seg, chunk;
    if (!seg_chunk(n_segments, chunks, &seg, &chunk)) return;
    const u32 lo = dstart[seg * nb], hi = dstart[(seg + 1) * nb];
    for (u32 base = chunk * (GASM_WG * GASM_EDGE_ILP); base < hi - lo; base += chunks * (GASM_WG * GASM_EDGE_ILP)) {
        u32 i[GASM_EDGE_ILP];
        bool ok[GASM_EDGE_ILP];
#pragma unroll
        for (int q = 0; q < GASM_EDGE_ILP; ++q) { i[q] = lo + base + q * GASM_WG + threadIdx.x; ok[q] = i[q] < hi; }
        f(seg, lo, hi, i, ok);
    }
}

template <class T>
__device__ __forceinline__ u32 lower_bound_dev(const T* __restrict__ a, u32 lo, u32 hi, T t) {
    while (lo < hi) {
        const u32 m = (lo + hi) >> 1;
        if (a[m] < t) lo = m + 1;
        else hi = m;
    }
    return lo;
}

// largest i in [0, n) with a[i] <= t, for a[0] <= t
template <class T>
__device__ __forceinline__ u32 upper_seg(const T* __restrict__ a, u32 n, T t) {
    u32 lo = 0, hi = n;  // invariant a[lo] <= t < a[hi] (a[n] = +inf)
    while (hi - lo > 1) {
        const u32 m = (lo + hi) >> 1;
        if (a[m] <= t) lo = m;
        else hi = m;
    }
    return lo;
}
