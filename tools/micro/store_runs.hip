// Micro-benchmark: HBM write rate of "bucketed run" stores on gfx950.  Each wave-instruction writes 64 lanes x W bytes
// as 64*W/(8R) ... runs: lane l -> run l / LPR (LPR lanes per run), consecutive inside the run; every run goes to a different
// bucket stream (64 streams per workgroup-segment, 1.6 GB total footprint), streams advance iteration by iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64; typedef uint32_t u32;
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

// W = bytes per lane (8 or 16), LPR = lanes per run
template <int W, int LPR, bool NT, int MIS>
__global__ void __launch_bounds__(256) k(u64* out, u64 stream_words, u32 iters) {
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6, ln = threadIdx.x & 63;
    const u32 nwaves = gridDim.x * 4;
    const u32 run = ln / LPR, inrun = ln % LPR;
    constexpr u32 RUNS = 64 / LPR;                  // runs per wave-instruction
    constexpr u32 WPL = W / 8;                      // words per lane
    // 6400 streams; wave w writes runs to streams (w*7 + it*RUNS + run) % 6400; position inside a stream advances with a per-wave slot
    for (u32 it = 0; it < iters; ++it) {
        const u32 stream = (wave * 7u + it * RUNS + run) % 6400u;
        const u64 pos = ((u64)wave + (u64)nwaves * (it / (6400u / RUNS + 1))) * (LPR * WPL) % (stream_words - LPR * WPL);
        u64* dst = out + (u64)stream * stream_words + (pos / WPL) * WPL + inrun * WPL + (MIS ? ((stream * 5u + it) % (16 / WPL)) * WPL : 0);
        if (W == 8) { if (NT) __builtin_nontemporal_store((u64)(it + ln), dst); else *dst = it + ln; }
        else { u64x2 v; v.x = it; v.y = ln; if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u64x2*>(dst)); else *reinterpret_cast<u64x2*>(dst) = v; }
    }
}

template <int W, int LPR, bool NT, int MIS = 0>
void run(const char* name, u64* d, u64 stream_words) {
    const u32 iters = 3072 * 8 / W / 1;   // bytes per wave constant: 3072*8*... each wave writes iters*64*W bytes
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int blocks = 256 * 8;
    hipLaunchKernelGGL((k<W, LPR, NT, MIS>), dim3(blocks), dim3(256), 0, 0, d, stream_words, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<W, LPR, NT, MIS>), dim3(blocks), dim3(256), 0, 0, d, stream_words, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)blocks * 4 * iters * 64 * W;
    printf("%-46s %8.3f ms  %8.1f GB/s  (%.2f GB per launch)\n", name, ms / 3, bytes / (ms / 3 * 1e-3) / 1e9, bytes / 1e9);
}

int main() {
    const u64 stream_words = 32768;                 // 256 KB per stream, 6400 streams = 1.6 GB
    u64* d; (void)hipMalloc(&d, 6400ull * stream_words * 8);
    run<8, 64, false>("8 B/lane, 1 run of 64 (512 B contiguous)", d, stream_words);
    run<16, 64, false>("16 B/lane, 1 run of 64 (1 KB contiguous)", d, stream_words);
    run<8, 16, false>("8 B/lane, 4 runs of 16 lanes (128 B)", d, stream_words);
    run<8, 8, false>("8 B/lane, 8 runs of 8 lanes (64 B)", d, stream_words);
    run<8, 4, false>("8 B/lane, 16 runs of 4 lanes (32 B)", d, stream_words);
    run<16, 8, false>("16 B/lane, 8 runs of 8 lanes (128 B)", d, stream_words);
    run<16, 16, false>("16 B/lane, 4 runs of 16 lanes (256 B)", d, stream_words);
    run<16, 8, true>("16 B/lane nt, 8 runs of 8 lanes (128 B)", d, stream_words);
    run<8, 16, true>("8 B/lane nt, 4 runs of 16 lanes (128 B)", d, stream_words);
    run<16, 32, false>("16 B/lane, 2 runs of 32 lanes (512 B)", d, stream_words);
    run<8, 16, false, 1>("8 B/lane, 4 runs of 16 lanes, MISALIGNED", d, stream_words);
    run<8, 64, false, 1>("8 B/lane, 1 run of 64 lanes, MISALIGNED", d, stream_words);
    run<8, 32, false, 1>("8 B/lane, 2 runs of 32 lanes, MISALIGNED", d, stream_words);
    run<16, 16, false, 1>("16 B/lane, 4 runs of 16 lanes, MISALIGNED", d, stream_words);
    return 0;
}
