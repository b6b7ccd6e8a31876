// Occupancy census on gfx950: every block spins 20 us; 2560 blocks on 256 CUs take ceil(10 / resident) * 20 us.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int BYTES> __global__ void __launch_bounds__(256) ks(unsigned* out) {
    __shared__ unsigned s[BYTES / 4];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) {}     // 100 MHz ticks: 20 us
    if (s[threadIdx.x] == 0xFFFFFFFF) out[0] = 1;
}
__global__ void __launch_bounds__(256) kd(unsigned* out) {
    extern __shared__ unsigned s[];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) {}
    if (s[threadIdx.x] == 0xFFFFFFFF) out[0] = 1;
}
template <class F> float timeit(F f) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); for (int i = 0; i < 3; ++i) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / 3;
}
#define S(B) printf("static  %6d B: %.1f us  -> resident/CU = %.1f\n", B, 1e3 * timeit([&] { hipLaunchKernelGGL(ks<B>, dim3(2560), dim3(256), 0, 0, d); }), 200.0 / (1e3 * timeit([&] { hipLaunchKernelGGL(ks<B>, dim3(2560), dim3(256), 0, 0, d); })))
#define D(B) printf("dynamic %6d B: %.1f us  -> resident/CU = %.1f\n", B, 1e3 * timeit([&] { hipLaunchKernelGGL(kd, dim3(2560), dim3(256), B, 0, d); }), 200.0 / (1e3 * timeit([&] { hipLaunchKernelGGL(kd, dim3(2560), dim3(256), B, 0, d); })))
int main() {
    unsigned* d; (void)hipMalloc(&d, 64);
    S(1024); S(8192); S(16384); S(20480); S(28704); S(32768); S(40960); S(49184); S(65536);
    D(8192); D(28704); D(40960);
    (void)hipFuncSetAttribute((const void*)kd, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    printf("after hipFuncSetAttribute(MaxDynamicSharedMemorySize = 160 KB):\n");
    D(8192); D(28704); D(36864); D(40960); D(54272); D(81920);
    return 0;
}
