// dedup_order.h — the ordering phase of a de-duplicated bucket, shared by k_bucket_dedup (kernels_build.hip) and
// k_bucket_merge (kernels_pool.hip).  Device code is not linked across translation units here (no -fgpu-rdc), hence a header.
#pragma once
#include "device_utils.h"
#include "keyops.h"

// The d distinct keys of the LDS table end up sorted in t_key[0, d) with their counts in t_cnt[0, d), and the fine
// directory of the bucket is written.  Counting sort on the key bits below the bucket prefix (TBL/4 bins; close to
// uniform there) + per-bin insertion sort; if any bin is long (skewed keys) a bitonic sort.  s_start must be zero on
// entry (unless BINS_IN_TABLE), s_tmp[6] (longest bin) too; every thread of the workgroup calls it (it contains barriers).
// BINS_IN_TABLE: s_start / s_cur point into the table itself, behind the LIMIT entries the sorted result can have (the
// caller's LDS is then the table alone — one more workgroup per CU); they are zeroed here, once every slot is in registers.
// RANGED (k_bucket_dedup_multi: the table holds one key sub-range of the bucket): only the bins [bin_lo, bin_hi) of the
// fine directory are written, offset by fdir_base (the distinct keys of the sub-ranges before), and not its last entry.
template <class K, int TBL, bool BINS_IN_TABLE = false, bool RANGED = false>
__device__ __forceinline__ void dedup_order(K* t_key, u32* t_cnt, u32* s_start, u32* s_cur, u32* s_tmp, u16* __restrict__ fdir,
                                            u32 bucket, int low_bits, u32 d, u32 fdir_base = 0, u32 bin_lo = 0, u32 bin_hi = TBL / 4) {
    constexpr int BINS = TBL / 4;
    constexpr int SL = TBL / GASM_WG;
    constexpr int LOG_TBL = TBL == 4096 ? 12 : 11;
    constexpr bool WIDE = sizeof(K) == 16;
    // ---- every thread pulls its slots (stride 256: conflict-free) into registers and bins them
    const int bshift = low_bits > (LOG_TBL - 2) ? low_bits - (LOG_TBL - 2) : 0;
    K rk[SL];
    u32 rc[SL];
#pragma unroll
    for (int q = 0; q < SL; ++q) {
        rk[q] = t_key[q * GASM_WG + threadIdx.x];
        rc[q] = t_cnt[q * GASM_WG + threadIdx.x];
        if (WIDE && rc[q] == 0) rk[q] = key_empty<K>();      // 128-bit tables mark free slots by the count
    }
    if constexpr (BINS_IN_TABLE) {
        __syncthreads();   // the table is in registers: its tail may become the bins
        for (u32 i = threadIdx.x; i < (u32)BINS; i += GASM_WG) s_start[i] = 0;
        if (threadIdx.x == 0) s_tmp[6] = 0;      // (s_tmp may live in the table too)
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < SL; ++q)
        if (!kis_empty(rk[q])) atomicAdd(&s_start[kfield(rk[q], bshift) & (BINS - 1)], 1u);
    __syncthreads();   // all table reads and all bin counts are done
    {
        constexpr int PER = BINS / GASM_WG;   // 4 or 2
        u32 c[PER], sum = 0, mx = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) { c[q] = s_start[threadIdx.x * PER + q]; sum += c[q]; mx = c[q] > mx ? c[q] : mx; }
        u32 tot;
        u32 ex = block_excl_scan<GASM_WG>(sum, s_tmp, &tot);
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            s_start[threadIdx.x * PER + q] = ex;
            s_cur[threadIdx.x * PER + q] = ex;
            const u32 bin = threadIdx.x * PER + q;
            if (!RANGED || (bin >= bin_lo && bin < bin_hi))
                fdir[(u64)bucket * (BINS + 1) + bin] = (u16)(fdir_base + ex);   // fine directory for the graph kernels
            ex += c[q];
        }
        if (!RANGED && threadIdx.x == GASM_WG - 1) fdir[(u64)bucket * (BINS + 1) + BINS] = (u16)ex;
        if (mx > 1) atomicMax(&s_tmp[6], mx);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < SL; ++q) {
        if (!kis_empty(rk[q])) {
            const u32 pos = atomicAdd(&s_cur[kfield(rk[q], bshift) & (BINS - 1)], 1u);
            t_key[pos] = rk[q];
            t_cnt[pos] = rc[q];
        }
    }
    __syncthreads();
    const u32 longest = s_tmp[6];
    if (longest <= 24) {
        // per-bin insertion sort (bins of 0/1 keys need nothing)
        for (u32 b = threadIdx.x; b < (u32)BINS; b += GASM_WG) {
            const u32 lo = s_start[b], hi = s_cur[b];
            for (u32 i = lo + 1; i < hi; ++i) {
                const K kx = t_key[i];
                const u32 cx = t_cnt[i];
                u32 j = i;
                while (j > lo && kless(kx, t_key[j - 1])) { t_key[j] = t_key[j - 1]; t_cnt[j] = t_cnt[j - 1]; --j; }
                t_key[j] = kx;
                t_cnt[j] = cx;
            }
        }
        __syncthreads();
    } else {
        // skewed keys: bitonic sort of the compacted entries
        u32 p2 = 2;
        while (p2 < d) p2 <<= 1;
        for (u32 i = d + threadIdx.x; i < p2; i += GASM_WG) { t_key[i] = key_empty<K>(); t_cnt[i] = 0; }
        __syncthreads();
        for (u32 kk = 2; kk <= p2; kk <<= 1) {
            for (u32 j = kk >> 1; j > 0; j >>= 1) {
                for (u32 t = threadIdx.x; t < (p2 >> 1); t += GASM_WG) {
                    const u32 lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    const u32 hi = lo | j;
                    const K a = t_key[lo], b = t_key[hi];
                    const bool up = (lo & kk) == 0;
                    if (kless(b, a) == up && !keq(a, b)) {
                        t_key[lo] = b; t_key[hi] = a;
                        const u32 ca = t_cnt[lo], cb = t_cnt[hi];
                        t_cnt[lo] = cb; t_cnt[hi] = ca;
                    }
                }
                __syncthreads();
            }
        }
    }
}
