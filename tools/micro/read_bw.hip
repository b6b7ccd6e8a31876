// Micro-benchmark: HBM read rate of the de-duplication kernel's access pattern on gfx950.  1.6 GB of 64-bit words in 6400
// contiguous "buckets" of 250 KB; a workgroup of 256 threads streams a bucket with LD 16-byte loads per thread in flight
// (iterations of LD x 4 KB), the way k_bucket_dedup does, and xors what it reads.  Variants: workgroups per CU (dynamic
// LDS used as an occupancy limiter, as the kernel's tables do), loads in flight, one bucket per workgroup vs persistent
// workgroups, and a plain grid-stride sweep of the same bytes for reference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64; typedef uint32_t u32;

template <int LD>
__global__ void __launch_bounds__(256) k_bucket(const uint4* __restrict__ src, u32 chunks_per_bucket, u32 n_buckets, u32* sink) {
    extern __shared__ u32 lds[];
    u32 acc = 0;
    for (u32 b = blockIdx.x; b < n_buckets; b += gridDim.x) {
        const uint4* p = src + (size_t)b * chunks_per_bucket;
        for (u32 c = 0; c + LD * 256 <= chunks_per_bucket; c += LD * 256) {
            uint4 v[LD];
#pragma unroll
            for (int q = 0; q < LD; ++q) v[q] = p[c + q * 256 + threadIdx.x];
#pragma unroll
            for (int q = 0; q < LD; ++q) acc ^= v[q].x ^ v[q].y ^ v[q].z ^ v[q].w;
        }
    }
    if (acc == 0x12345678u) { lds[threadIdx.x] = acc; *sink = lds[(threadIdx.x + 1) & 255]; }
}

__global__ void __launch_bounds__(256) k_sweep(const uint4* __restrict__ src, size_t n, u32* sink) {
    u32 acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const uint4 v = src[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}

template <int LD>
void run(const char* name, const uint4* d, u32 cpb, u32 nb, u32 grid, size_t lds, u32* sink) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_bucket<LD>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k_bucket<LD>, dim3(grid), dim3(256), lds, 0, d, cpb, nb, sink);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_bucket<LD>, dim3(grid), dim3(256), lds, 0, d, cpb, nb, sink);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)nb * (cpb / (LD * 256)) * (LD * 256) * 16.0;
    printf("%-64s %8.3f ms  %8.1f GB/s\n", name, ms / 5, bytes / (ms / 5 * 1e-3) / 1e9);
}

int main() {
    const u32 nb = 6400, cpb = 16384;              // 6400 buckets of 16384 x 16 B = 256 KB: 1.68 GB
    uint4* d; u32* sink;
    (void)hipMalloc(&d, (size_t)nb * cpb * 16 + 4096);
    (void)hipMalloc(&sink, 64);
    (void)hipMemset(d, 1, (size_t)nb * cpb * 16);
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    const u32 cu = pr.multiProcessorCount;
    run<4>("one bucket per WG, 4 loads in flight, LDS 28 KB (5 WG/CU)", d, cpb, nb, nb, 28704, sink);
    run<4>("one bucket per WG, 4 loads in flight, LDS 16 KB (8+ WG/CU)", d, cpb, nb, nb, 16384, sink);
    run<4>("one bucket per WG, 4 loads in flight, no LDS", d, cpb, nb, nb, 0, sink);
    run<8>("one bucket per WG, 8 loads in flight, LDS 28 KB (5 WG/CU)", d, cpb, nb, nb, 28704, sink);
    run<16>("one bucket per WG, 16 loads in flight, LDS 28 KB (5 WG/CU)", d, cpb, nb, nb, 28704, sink);
    run<4>("persistent 5 WG/CU, 4 loads in flight, LDS 28 KB", d, cpb, nb, cu * 5, 28704, sink);
    run<8>("persistent 5 WG/CU, 8 loads in flight, LDS 28 KB", d, cpb, nb, cu * 5, 28704, sink);
    run<4>("persistent 8 WG/CU, 4 loads in flight, no LDS", d, cpb, nb, cu * 8, 0, sink);
    run<2>("one bucket per WG, 2 loads in flight, no LDS", d, cpb, nb, nb, 0, sink);
    run<4>("one bucket per WG, 4 loads, LDS 56 KB (2 WG/CU)", d, cpb, nb, nb, 57344, sink);
    {
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        const size_t n = (size_t)nb * cpb;
        for (u32 g : {cu * 8u, cu * 16u, cu * 32u}) {
            hipLaunchKernelGGL(k_sweep, dim3(g), dim3(256), 0, 0, d, n, sink);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(a);
            for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_sweep, dim3(g), dim3(256), 0, 0, d, n, sink);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            printf("grid-stride sweep, %5u workgroups, 1 load in flight              %8.3f ms  %8.1f GB/s\n", g, ms / 5, n * 16.0 / (ms / 5 * 1e-3) / 1e9);
        }
    }
    return 0;
}
