// Micro-benchmark of the LDS hash-insert loop of k_bucket_dedup in isolation (keys generated in registers).
// 6400 workgroups x 256 threads x 15 iterations x 8 keys = 196.6 M keys, ~600 distinct per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64; typedef uint32_t u32;
#define EMPTY 0xFFFFFFFFFFFFFFFFull
#define ITERS 15
__device__ __forceinline__ u32 hash64(u64 x) { return (u32)((x * 0x9E3779B97F4A7C15ull) >> 32); }

template <int TBL, int LOG, int VAR, int DIST>
__global__ void __launch_bounds__(256) k(u32* out) {
    __shared__ u64 t_key[TBL];
    __shared__ u32 t_cnt[TBL];
    __shared__ u32 s_nd;
    for (int i = threadIdx.x; i < TBL; i += 256) { t_key[i] = EMPTY; t_cnt[i] = 0; }
    if (threadIdx.x == 0) s_nd = 0;
    __syncthreads();
    u32 x = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u) ^ 0x1234567u;
    auto slow = [&](u64 key) {
        u32 h = hash64(key) >> (32 - LOG);
        for (u32 p = 0; p < TBL; ++p) {
            const u64 cur = __hip_atomic_load(&t_key[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (cur == key) { atomicAdd(&t_cnt[h], 1u); return; }
            if (cur == EMPTY) {
                const u64 old = atomicCAS((unsigned long long*)&t_key[h], (unsigned long long)EMPTY, (unsigned long long)key);
                if (old == EMPTY) { atomicAdd(&s_nd, 1u); atomicAdd(&t_cnt[h], 1u); return; }
                if (old == key) { atomicAdd(&t_cnt[h], 1u); return; }
            }
            h = (h + 1) & (TBL - 1);
        }
    };
    for (int it = 0; it < ITERS; ++it) {
        u64 kx[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            x = x * 1664525u + 1013904223u;
            const u32 idx = (x >> 8) % DIST;
            kx[q] = ((u64)(idx + 1) * 0xD6E8FEB86659FD93ull + blockIdx.x) & 0x3FFFFFFFFFFFFFFFull;
        }
        if (VAR == 0) {          // current: batched first probe, serial slow path
            u64 cur[8]; u32 hx[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { hx[q] = hash64(kx[q]) >> (32 - LOG); cur[q] = __hip_atomic_load(&t_key[hx[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#pragma unroll
            for (int q = 0; q < 8; ++q) { if (cur[q] == kx[q]) atomicAdd(&t_cnt[hx[q]], 1u); else slow(kx[q]); }
        } else if (VAR == 1) {   // plain serial
#pragma unroll
            for (int q = 0; q < 8; ++q) slow(kx[q]);
        } else if (VAR == 2) {   // rounds: all pending keys probe together
            u32 hx[8]; u32 pending = 0xFF;
#pragma unroll
            for (int q = 0; q < 8; ++q) hx[q] = hash64(kx[q]) >> (32 - LOG);
            for (int round = 0; round < TBL && pending; ++round) {
                u64 cur[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) cur[q] = (pending >> q) & 1 ? __hip_atomic_load(&t_key[hx[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (!((pending >> q) & 1)) continue;
                    if (cur[q] == kx[q]) { atomicAdd(&t_cnt[hx[q]], 1u); pending &= ~(1u << q); }
                    else if (cur[q] == EMPTY) {
                        const u64 old = atomicCAS((unsigned long long*)&t_key[hx[q]], (unsigned long long)EMPTY, (unsigned long long)kx[q]);
                        if (old == EMPTY) { atomicAdd(&s_nd, 1u); atomicAdd(&t_cnt[hx[q]], 1u); pending &= ~(1u << q); }
                        else if (old == kx[q]) { atomicAdd(&t_cnt[hx[q]], 1u); pending &= ~(1u << q); }
                        else hx[q] = (hx[q] + 1) & (TBL - 1);
                    } else hx[q] = (hx[q] + 1) & (TBL - 1);
                }
            }
        } else if (VAR == 3) {   // VALU only (key generation + hash), no LDS
            u32 a = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) a += hash64(kx[q]) >> (32 - LOG);
            if (a == 0x7654321) out[3] = a;
        }
    }
    __syncthreads();
    // validation: counts must add up to all keys inserted
    u32 c = 0;
    for (int i = threadIdx.x; i < TBL; i += 256) c += t_cnt[i];
    atomicAdd(&out[0], c);
    if (threadIdx.x == 0) atomicAdd(&out[1], s_nd);
}

template <int TBL, int LOG, int VAR, int DIST>
void run(const char* name, u32* d_out) {
    const int blocks = 6400;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k<TBL, LOG, VAR, DIST>), dim3(blocks), dim3(256), 0, 0, d_out);
    (void)hipDeviceSynchronize();
    (void)hipMemset(d_out, 0, 64);
    (void)hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<TBL, LOG, VAR, DIST>), dim3(blocks), dim3(256), 0, 0, d_out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    u32 h[4]; (void)hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
    const double keys = (double)blocks * 256 * ITERS * 8;
    printf("%-44s %8.4f ms/launch  counted %.0f of %.0f keys, distinct/WG %.1f\n", name, ms / 5, h[0] / 5.0, VAR == 3 ? 0.0 : keys, h[1] / 5.0 / blocks);
}

int main() {
    u32* d_out; (void)hipMalloc(&d_out, 64);
    run<2048, 11, 3, 600>("VALU only (keygen+hash)", d_out);
    run<2048, 11, 0, 600>("V0 batched first probe, TBL 2048, D 600", d_out);
    run<2048, 11, 1, 600>("V1 serial, TBL 2048, D 600", d_out);
    run<2048, 11, 2, 600>("V2 rounds, TBL 2048, D 600", d_out);
    run<4096, 12, 0, 600>("V0 batched first probe, TBL 4096, D 600", d_out);
    run<4096, 12, 2, 600>("V2 rounds, TBL 4096, D 600", d_out);
    run<4096, 12, 2, 1400>("V2 rounds, TBL 4096, D 1400", d_out);
    run<2048, 11, 2, 100>("V2 rounds, TBL 2048, D 100", d_out);
    return 0;
}
