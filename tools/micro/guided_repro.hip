// k_guided_chain exactly as round 2 had it (comparator spelled with __umul64hi), on the kept case: which seeds does it pick?
// Reproducer of a wrong code generation (hipcc of ROCm 7.2, gfx950, -O3): in the candidate loop `if (gbest_better(c, b)) b = c;`
// updates b.fs and b.idx but NOT b.len for lanes whose b already held a candidate (ISA: `v_mov_b32 v13, v16` for every comparing
// lane after `; implicit-def: $vgpr13`), so a lane's best carries the length of its FIRST candidate and the ratios are wrong.
//   hipcc -O3 --offload-arch=gfx950 guided_repro.hip                  -> seeds 299 106 72 ...   (wrong)
//   hipcc ... -DSELECT_UPDATE   (field-by-field selects)               -> seeds 57 299 106 ...   (the specification)
//   hipcc ... -DNEWCMP          (unsigned __int128 comparator)         -> seeds 57 299 106 ...
#include "../../genomeassembler_dev_amd/csrc/device_utils.h"
#include <cstdio>
#include <vector>
struct PathSet { const u64* words; const u64* p_off; const u32* seg_path_off; u32 n_segments; };
struct GBest { unsigned long long fs; u32 len; u32 idx; };
__device__ __forceinline__ bool gbest_better(const GBest& a, const GBest& b) {
    if (b.idx == GASM_NONE32) return a.idx != GASM_NONE32;
    if (a.idx == GASM_NONE32) return false;
#ifdef NEWCMP
    const unsigned __int128 pa = (unsigned __int128)a.fs * b.len, pb = (unsigned __int128)b.fs * a.len;
    if (pa != pb) return pa > pb;
    return a.idx < b.idx;
#else
    const unsigned long long al = a.fs * (unsigned long long)b.len, ah = __umul64hi(a.fs, (unsigned long long)b.len);
    const unsigned long long bl = b.fs * (unsigned long long)a.len, bh = __umul64hi(b.fs, (unsigned long long)a.len);
    if (ah != bh) return ah > bh;
    if (al != bl) return al > bl;
    return a.idx < b.idx;
#endif
}
__device__ __forceinline__ GBest gbest_wave(GBest v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        GBest o;
        o.fs = __shfl_xor(v.fs, d, 64); o.len = __shfl_xor(v.len, d, 64); o.idx = __shfl_xor(v.idx, d, 64);
        if (gbest_better(o, v)) v = o;
    }
    return v;
}
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
                                                     u32* __restrict__ g_prev, u32* __restrict__ seeds) {
    extern __shared__ u8 s_used[];
    const u32 seg = blockIdx.x, lane = threadIdx.x;
    const u32 c0 = ps.seg_path_off[seg], n = ps.seg_path_off[seg + 1] - c0;
    for (u32 i = lane; i < n; i += 64) { s_used[i] = 0; g_next[c0 + i] = GASM_NONE32; g_prev[c0 + i] = GASM_NONE32; }
    __syncthreads();
    const int k1 = k - 1;
    u32 ns = 0;
    auto best_of = [&](int mode, u32 cur) {
        GBest b{0ull, 1u, GASM_NONE32};
        const u64 cb = mode ? ps.p_off[c0 + cur] : 0, ce = mode ? ps.p_off[c0 + cur + 1] : 0;
        for (u32 j = lane; j < n; j += 64) {
            if (s_used[j]) continue;
            const u64 jb = ps.p_off[c0 + j], je = ps.p_off[c0 + j + 1];
            if (mode == 1 && !bases_eq_short(ps.words, ce - k1, jb, k1)) continue;
            if (mode == 2 && !bases_eq_short(ps.words, je - k1, cb, k1)) continue;
            const GBest c{fx[c0 + j], (u32)(je - jb), j};
#ifdef SELECT_UPDATE
            const bool take = gbest_better(c, b);
            b.fs = take ? c.fs : b.fs; b.len = take ? c.len : b.len; b.idx = take ? c.idx : b.idx;
#else
            if (gbest_better(c, b)) b = c;
#endif
        }
        return gbest_wave(b);
    };
    for (;;) {
        const GBest seed = best_of(0, 0);
        if (seed.idx == GASM_NONE32) break;
        if (lane == 0) { s_used[seed.idx] = 1; if (ns < 8) seeds[ns] = seed.idx; }
        ++ns;
        __syncthreads();
        for (int dir = 1; dir <= 2; ++dir) {
            u32 cur = seed.idx;
            for (;;) {
                const GBest nx = best_of(dir, cur);
                if (nx.idx == GASM_NONE32) break;
                if (lane == 0) {
                    s_used[nx.idx] = 1;
                    if (dir == 1) { g_next[c0 + cur] = c0 + nx.idx; g_prev[c0 + nx.idx] = c0 + cur; }
                    else { g_next[c0 + nx.idx] = c0 + cur; g_prev[c0 + cur] = c0 + nx.idx; }
                }
                __syncthreads();
                cur = nx.idx;
            }
        }
    }
}
int main() {
    FILE* f = fopen("tools/micro/case91b.bin", "rb");
    u32 n, nw; fread(&n, 4, 1, f); fread(&nw, 4, 1, f);
    std::vector<u64> fx(n), off(n + 1), w(nw);
    fread(fx.data(), 8, n, f); fread(off.data(), 8, n + 1, f); fread(w.data(), 8, nw, f); fclose(f);
    u64 *dfx, *doff, *dw; u32 *dseg, *dn, *dp, *ds;
    u32 seg[2] = {0, n}, hs[8] = {0};
    hipMalloc(&dfx, n * 8); hipMalloc(&doff, (n + 1) * 8); hipMalloc(&dw, nw * 8); hipMalloc(&dseg, 8); hipMalloc(&dn, n * 4); hipMalloc(&dp, n * 4); hipMalloc(&ds, 32);
    hipMemcpy(dfx, fx.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(doff, off.data(), (n + 1) * 8, hipMemcpyHostToDevice);
    hipMemcpy(dw, w.data(), nw * 8, hipMemcpyHostToDevice); hipMemcpy(dseg, seg, 8, hipMemcpyHostToDevice); hipMemset(ds, 0, 32);
    PathSet ps{dw, doff, dseg, 1};
    hipLaunchKernelGGL(k_guided_chain, dim3(1), dim3(64), n + 16, 0, ps, (const unsigned long long*)dfx, 21, dn, dp, ds);
    hipMemcpy(hs, ds, 32, hipMemcpyDeviceToHost);
    printf("seeds: %u %u %u %u %u %u  (the specification: 57 299 106 ...; the library's kernel of round 2 started with 299)\n", hs[0], hs[1], hs[2], hs[3], hs[4], hs[5]);
    return 0;
}
