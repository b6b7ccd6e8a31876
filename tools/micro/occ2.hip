// Occupancy census vs register budget on gfx950 (static LDS 28,704 B, 256-thread blocks spinning 20 us).
#include <hip/hip_runtime.h>
#include <cstdio>
#define KERNEL(NAME, CLOBBER)                                                        \
    __global__ void __launch_bounds__(256) NAME(unsigned* out) {                     \
        __shared__ unsigned s[28704 / 4];                                            \
        s[threadIdx.x] = threadIdx.x;                                                \
        __syncthreads();                                                             \
        CLOBBER;                                                                     \
        const unsigned long long t0 = wall_clock64();                                \
        while (wall_clock64() - t0 < 2000) {}                                        \
        if (s[threadIdx.x] == 0xFFFFFFFF) out[0] = 1;                                \
    }
KERNEL(k_plain, )
KERNEL(k_v63, asm volatile("" ::: "v63"))
KERNEL(k_v71, asm volatile("" ::: "v71"))
KERNEL(k_v80, asm volatile("" ::: "v80"))
KERNEL(k_v67_a12, asm volatile("" ::: "v67", "a12"))
KERNEL(k_v127, asm volatile("" ::: "v127"))
KERNEL(k_s95, asm volatile("" ::: "s95"))
KERNEL(k_s101, asm volatile("" ::: "s101"))
KERNEL(k_v67_a12_s95, asm volatile("" ::: "v67", "a12", "s95"))
template <class F> float timeit(F f) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); for (int i = 0; i < 3; ++i) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / 3;
}
#define T(K) { int n = 0; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, K, 256, 0); float us = 1e3f * timeit([&] { hipLaunchKernelGGL(K, dim3(2560), dim3(256), 0, 0, d); }); printf("%-16s API %d blocks/CU, 2560 blocks x 20 us took %.1f us -> %d rounds\n", #K, n, us, (int)(us / 20.0f + 0.3f)); }
int main() {
    unsigned* d; (void)hipMalloc(&d, 64);
    T(k_plain) T(k_v63) T(k_v71) T(k_v80) T(k_v67_a12) T(k_v127) T(k_s95) T(k_s101) T(k_v67_a12_s95)
    return 0;
}
