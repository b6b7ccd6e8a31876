// k_guided_chain's first seed choice on the kept case, old comparator (__umul64hi) and new (unsigned __int128)  (diagnostic)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct GBest { unsigned long long fs; unsigned len; unsigned idx; };
#define NONE 0xFFFFFFFFu
template <int NEW> __device__ __forceinline__ bool gbest_better(const GBest& a, const GBest& b) {
    if (b.idx == NONE) return a.idx != NONE;
    if (a.idx == NONE) return false;
    if (NEW) {
        const unsigned __int128 pa = (unsigned __int128)a.fs * b.len, pb = (unsigned __int128)b.fs * a.len;
        if (pa != pb) return pa > pb;
        return a.idx < b.idx;
    }
    const unsigned long long al = a.fs * (unsigned long long)b.len, ah = __umul64hi(a.fs, (unsigned long long)b.len);
    const unsigned long long bl = b.fs * (unsigned long long)a.len, bh = __umul64hi(b.fs, (unsigned long long)a.len);
    if (ah != bh) return ah > bh;
    if (al != bl) return al > bl;
    return a.idx < b.idx;
}
template <int NEW> __device__ __forceinline__ GBest gbest_wave(GBest v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        GBest o;
        o.fs = __shfl_xor(v.fs, d, 64); o.len = __shfl_xor(v.len, d, 64); o.idx = __shfl_xor(v.idx, d, 64);
        if (gbest_better<NEW>(o, v)) v = o;
    }
    return v;
}
template <int NEW> __global__ void __launch_bounds__(64) k(const unsigned long long* __restrict__ fx, const unsigned* __restrict__ len, unsigned n, unsigned* out) {
    extern __shared__ unsigned char s_used[];
    const unsigned lane = threadIdx.x;
    for (unsigned i = lane; i < n; i += 64) s_used[i] = 0;
    __syncthreads();
    for (int round = 0; round < 4; ++round) {
        GBest b{0ull, 1u, NONE};
        for (unsigned j = lane; j < n; j += 64) {
            if (s_used[j]) continue;
            const GBest c{fx[j], len[j], j};
            if (gbest_better<NEW>(c, b)) b = c;
        }
        const GBest w = gbest_wave<NEW>(b);
        if (lane == 0) { out[round] = w.idx; s_used[w.idx] = 1; }
        __syncthreads();
    }
}
int main() {
    FILE* f = fopen("tools/micro/case91.bin", "rb");
    unsigned n; fread(&n, 4, 1, f);
    std::vector<unsigned long long> fx(n); std::vector<unsigned> len(n);
    fread(fx.data(), 8, n, f); fread(len.data(), 4, n, f); fclose(f);
    unsigned long long* dfx; unsigned *dl, *o; unsigned ho[8];
    hipMalloc(&dfx, n * 8); hipMalloc(&dl, n * 4); hipMalloc(&o, 32);
    hipMemcpy(dfx, fx.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dl, len.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), n + 16, 0, dfx, dl, n, o);
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), n + 16, 0, dfx, dl, n, o + 4);
    hipMemcpy(ho, o, 32, hipMemcpyDeviceToHost);
    printf("old: %u %u %u %u   new: %u %u %u %u   (want 57 299 106 56)\n", ho[0], ho[1], ho[2], ho[3], ho[4], ho[5], ho[6], ho[7]);
    return 0;
}
