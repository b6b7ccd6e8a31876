// Micro-benchmark: cost of LDS operations with random vs sequential addresses on gfx950 (cycles per wave-instruction,
// per CU, all CUs busy).  Build: hipcc -O3 --offload-arch=gfx950 lds_ops.hip -o lds_ops ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint64_t u64; typedef uint32_t u32;
#define N_ITER 2048
#define TBL 4096

template <int OP, bool RANDOM>
__global__ void __launch_bounds__(256) k(u32* out, u32 seed) {
    __shared__ u64 t64[TBL];
    __shared__ u32 t32[TBL];
    for (int i = threadIdx.x; i < TBL; i += 256) { t64[i] = ~0ull; t32[i] = 0; }
    __syncthreads();
    u32 x = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    u32 acc = 0;
    for (int it = 0; it < N_ITER; ++it) {
        x = x * 1664525u + 1013904223u;
        const u32 h = RANDOM ? (x >> 20) & (TBL - 1) : ((it * 256 + threadIdx.x) & (TBL - 1));
        if (OP == 0) acc += (u32)__hip_atomic_load(&t64[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (OP == 1) atomicAdd(&t32[h], 1u);
        if (OP == 2) acc += atomicAdd(&t32[h], 1u);
        if (OP == 3) acc += (u32)atomicCAS((unsigned long long*)&t64[h], ~0ull, (unsigned long long)x);
        if (OP == 4) t64[h] = x;
        if (OP == 5) acc += t32[h];
        if (OP == 6) { acc += (u32)__hip_atomic_load(&t64[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); atomicAdd(&t32[h], 1u); }
        if (OP == 7) { /* VALU only */ acc += (u32)(((u64)x * 0x9E3779B97F4A7C15ull) >> 40); }
        if (OP == 8) atomicAdd(&t32[(x >> 20) & 63], 1u);          // 64 hot addresses
        if (OP == 9) acc += atomicAdd(&t32[(x >> 20) & 63], 1u);   // 64 hot addresses, returning
    }
    if (acc == 0x12345) out[0] = acc;
    if (threadIdx.x == 0 && OP == 1) out[1 + blockIdx.x % 7] = t32[5];
}

template <int OP, bool RANDOM>
double run(u32* d_out, int blocks) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<OP, RANDOM>), dim3(blocks), dim3(256), 0, 0, d_out, 1u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<OP, RANDOM>), dim3(blocks), dim3(256), 0, 0, d_out, 7u + r);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}

int main() {
    u32* d_out; hipMalloc(&d_out, 64);
    const int blocks = 256 * 3;   // 3 workgroups (12 waves) per CU, like the 48 KB kernels
    const char* names[] = {"ds_read_b64", "ds_add_u32", "ds_add_rtn_u32", "ds_cmpst_rtn_b64", "ds_write_b64", "ds_read_b32",
                           "read_b64+add_u32", "valu_only(mul64)", "ds_add_u32 64 hot", "ds_add_rtn 64 hot"};
    double ms[10][2];
#define RUN(i) ms[i][0] = run<i, false>(d_out, blocks); ms[i][1] = run<i, true>(d_out, blocks);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9)
    // wave-instructions per CU = 12 waves * N_ITER; cycles at 2.4 GHz
    for (int i = 0; i < 10; ++i)
        for (int r = 0; r < 2; ++r) {
            const double cyc = ms[i][r] * 1e-3 * 2.4e9 / (12.0 * N_ITER);
            printf("%-22s %-10s %8.4f ms  %7.2f cycles per wave-instruction per CU (12 waves/CU)\n", names[i], r ? "random" : "sequential", ms[i][r], cyc);
        }
    return 0;
}
