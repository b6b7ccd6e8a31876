// kernels_pool.hip — kernels of the pooled (multi-GPU) build: SURVEY §8(e) mode 2.
//
// In a pooled build the reads of every segment are spread evenly over the ranks.  Each rank de-duplicates its own
// k-mers per (segment, bucket) with the single-GPU kernels (kernels_build.hip); what it then holds per bucket is a sorted
// run of distinct (key, count) records.  Runs travel between ranks (RCCL all-to-all, driven by the host layer) and
// are combined where they arrive:
//   k_pack_runs      the runs of a list of buckets, back to back in a send buffer (keys and counts as separate streams:
//                    12 bytes per record for k <= 31, 20 for k <= 63 — the (segment, bucket) of a record is implied by
//                    its position in a run directory both sides compute from the partition function);
//   k_bucket_merge   all runs received for one bucket -> one sorted run with the counts added up + the bucket's fine
//                    directory; same LDS table, same ordering code as k_bucket_dedup, with a count per insertion;
//   k_repack_reads   a range of the packed read stream copied to a word boundary (the reads of a segment on their way
//                    to the rank that scores the segment).
// None of this is on the single-GPU hot path; the records are the ~cov-fold reduced output of the local de-duplication.
#include "device_utils.h"
#include "keyops.h"
#include "kernels.h"
#include "dedup_order.h"

typedef unsigned long long u64x2p __attribute__((ext_vector_type(2)));

// one workgroup per listed bucket: keys[bstart[gb] .. + bucket_d[gb]) -> out_keys[dst_off[i] ..)
template <class K>
__global__ void __launch_bounds__(GASM_WG) k_pack_runs(const K* __restrict__ keys, const u32* __restrict__ mult, const u64* __restrict__ bstart,
                                                       const u32* __restrict__ bucket_d, const u32* __restrict__ gb_list,
                                                       const u64* __restrict__ dst_off, K* __restrict__ out_keys, u32* __restrict__ out_cnt) {
    const u32 gb = gb_list[blockIdx.x];
    const u64 src = bstart[gb], dst = dst_off[blockIdx.x];
    const u32 d = bucket_d[gb];
    for (u32 i = threadIdx.x; i < d; i += GASM_WG) { out_keys[dst + i] = keys[src + i]; out_cnt[dst + i] = mult[src + i]; }
}
template __global__ void k_pack_runs<u64>(const u64*, const u32*, const u64*, const u32*, const u32*, const u64*, u64*, u32*);
template __global__ void k_pack_runs<K128>(const K128*, const u32*, const u64*, const u32*, const u32*, const u64*, K128*, u32*);

// ---- insertion with a count (the slow-path step of k_bucket_dedup, kernels_build.hip, generalised)
#define GASM_SLOT_LOCKED 0xFFFFFFFFu
template <int TBL>
__device__ __forceinline__ bool merge_step(u64* t_key, u32* t_cnt, u32* n_distinct, u64 key, u32 w, u32& set) {
    constexpr u32 NSETS = TBL / 2;
    __asm__ volatile("" ::: "memory");
    const u64x2p c = *reinterpret_cast<const u64x2p*>(&t_key[2 * set]);
    int slot = c.x == key ? 0 : c.y == key ? 1 : -1;
    if (slot < 0) {
        const int emp = c.x == GASM_EMPTY64 ? 0 : c.y == GASM_EMPTY64 ? 1 : -1;
        if (emp < 0) { set = (set + 1) & (NSETS - 1); return false; }
        const u64 old = atomicCAS(reinterpret_cast<unsigned long long*>(&t_key[2 * set + emp]), (unsigned long long)GASM_EMPTY64,
                                  (unsigned long long)key);
        if (old == GASM_EMPTY64) atomicAdd(n_distinct, 1u);
        else if (old != key) return false;       // someone else took the slot: look at the set again
        slot = emp;
    }
    atomicAdd(&t_cnt[2 * set + slot], w);
    return true;
}
// 128-bit keys: the slot's count word is the lock (0 free, LOCKED being written, else ready).  A record's count is never
// 0 or LOCKED (a multiplicity of 2^32 - 1 is refused by the host: the counts are 32-bit).
template <int TBL>
__device__ __forceinline__ bool merge_step(K128* t_key, u32* t_cnt, u32* n_distinct, const K128& key, u32 w, u32& set) {
    constexpr u32 NSETS = TBL / 2;
    __asm__ volatile("" ::: "memory");
    const uint2 c = *reinterpret_cast<const uint2*>(&t_cnt[2 * set]);
    const u32 cc[2] = {c.x, c.y};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (cc[q] == GASM_SLOT_LOCKED) return false;
        if (cc[q] == 0) {
            if (atomicCAS(&t_cnt[2 * set + q], 0u, GASM_SLOT_LOCKED) != 0u) return false;
            t_key[2 * set + q] = key;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __hip_atomic_store(&t_cnt[2 * set + q], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            atomicAdd(n_distinct, 1u);
            return true;
        }
        __asm__ volatile("" ::: "memory");
        const u64x2p kq = *reinterpret_cast<const u64x2p*>(&t_key[2 * set + q]);
        if (kq.x == key.hi && kq.y == key.lo) { atomicAdd(&t_cnt[2 * set + q], w); return true; }
    }
    set = (set + 1) & (NSETS - 1);
    return false;
}

// One workgroup per output bucket j.  Its inputs are n_src runs: run s lies at in_keys[run_off[j * n_src + s] ..) and has
// run_len[j * n_src + s] records (0 = none).  Output: the merged sorted run at out_keys[bstart[j] ..) (capacity
// bstart[j+1] - bstart[j] >= min(sum of the lengths, table limit)), its length in bucket_d[j], the fine directory of
// bucket j.  *overflow is raised when the union does not fit the table (the partition needs more bucket bits).
template <class K, int TBL>
__global__ void __launch_bounds__(GASM_WG, TBL == 4096 ? 2 : 3)      // (LDS: 56 KB / 44 KB per workgroup)
k_bucket_merge(const K* __restrict__ in_keys, const u32* __restrict__ in_cnt, const u64* __restrict__ run_off, const u32* __restrict__ run_len,
               u32 n_src, K* __restrict__ out_keys, u32* __restrict__ out_cnt, const u64* __restrict__ bstart, u32* __restrict__ bucket_d,
               u32* __restrict__ overflow, u16* __restrict__ fdir, int low_bits, const u64* __restrict__ src_base) {
    constexpr int LIMIT = TBL / 16 * 11;
    constexpr int BINS = TBL / 4;
    constexpr int LOG_SETS = (TBL == 4096 ? 12 : 11) - 1;
    constexpr u32 NSETS = 1u << LOG_SETS;
    __shared__ __align__(32) K t_key[TBL];
    __shared__ __align__(16) u32 t_cnt[TBL];
    __shared__ u32 s_start[BINS];
    __shared__ u32 s_cur[BINS];
    __shared__ u32 s_tmp[8];
    const u32 j = blockIdx.x;
    // ---- a bucket with a single source run (every bucket of the second exchange; all of them with one rank) is sorted and
    // distinct already: copy it and build its fine directory by searching the bin boundaries
    {
        u32 n_nonempty = 0, only = 0;
        for (u32 s = 0; s < n_src; ++s) if (run_len[(u64)j * n_src + s]) { ++n_nonempty; only = s; }
        if (n_nonempty <= 1) {
            const u32 len = n_nonempty ? run_len[(u64)j * n_src + only] : 0u;
            const u64 off = n_nonempty ? run_off[(u64)j * n_src + only] + (src_base ? src_base[only] : 0ull) : 0ull;
            if (len > (u32)LIMIT) {
                for (u32 i = threadIdx.x; i <= (u32)BINS; i += GASM_WG) fdir[(u64)j * (BINS + 1) + i] = 0;
                if (threadIdx.x == 0) { atomicOr(overflow, 1u); bucket_d[j] = 0; }
                return;
            }
            const u64 beg = bstart[j];
            for (u32 i = threadIdx.x; i < len; i += GASM_WG) { out_keys[beg + i] = in_keys[off + i]; out_cnt[beg + i] = in_cnt[off + i]; }
            constexpr int LOG_TBL = TBL == 4096 ? 12 : 11;
            const int bshift = low_bits > (LOG_TBL - 2) ? low_bits - (LOG_TBL - 2) : 0;
            for (u32 b = threadIdx.x; b <= (u32)BINS; b += GASM_WG) {
                u32 lo = 0, hi = len;                       // first record whose bin is >= b (bins ascend along a sorted run)
                while (lo < hi) {
                    const u32 m = (lo + hi) >> 1;
                    if ((kfield(in_keys[off + m], bshift) & (u32)(BINS - 1)) < b) lo = m + 1; else hi = m;
                }
                fdir[(u64)j * (BINS + 1) + b] = (u16)(b == (u32)BINS ? len : lo);
            }
            if (threadIdx.x == 0) bucket_d[j] = len;
            return;
        }
    }
    for (u32 i = threadIdx.x; i < TBL; i += GASM_WG) { t_key[i] = key_empty<K>(); t_cnt[i] = 0; }
    for (u32 i = threadIdx.x; i < BINS; i += GASM_WG) s_start[i] = 0;
    if (threadIdx.x == 0) { s_tmp[4] = 0; s_tmp[5] = 0; s_tmp[6] = 0; }
    __syncthreads();
    for (u32 s = 0; s < n_src; ++s) {
        const u64 off = run_off[(u64)j * n_src + s] + (src_base ? src_base[s] : 0ull);    // (src_base: run_off counts from the source's first record)
        const u32 len = run_len[(u64)j * n_src + s];
        for (u32 i = threadIdx.x; i < len; i += GASM_WG) {
            if (__hip_atomic_load(&s_tmp[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > (u32)LIMIT) { s_tmp[5] = 1; break; }
            const K key = in_keys[off + i];
            const u32 w = in_cnt[off + i];
            u32 st = khash(key) >> (32 - LOG_SETS);
            bool ok = false;
            for (u32 probe = 0; probe < 8 * NSETS && !ok; ++probe) ok = merge_step<TBL>(t_key, t_cnt, &s_tmp[4], key, w, st);
            if (!ok) s_tmp[5] = 1;
        }
    }
    __syncthreads();
    if (s_tmp[5] || s_tmp[4] > (u32)LIMIT) {
        for (u32 i = threadIdx.x; i <= (u32)BINS; i += GASM_WG) fdir[(u64)j * (BINS + 1) + i] = 0;      // empty and searchable
        if (threadIdx.x == 0) { atomicOr(overflow, 1u); bucket_d[j] = 0; }
        return;
    }
    const u32 d = s_tmp[4];
    dedup_order<K, TBL>(t_key, t_cnt, s_start, s_cur, s_tmp, fdir, j, low_bits, d);
    const u64 beg = bstart[j];
    for (u32 i = threadIdx.x; i < d; i += GASM_WG) { out_keys[beg + i] = t_key[i]; out_cnt[beg + i] = t_cnt[i]; }
    if (threadIdx.x == 0) bucket_d[j] = d;
}
template __global__ void k_bucket_merge<u64, 4096>(const u64*, const u32*, const u64*, const u32*, u32, u64*, u32*, const u64*, u32*, u32*, u16*, int, const u64*);
template __global__ void k_bucket_merge<K128, 2048>(const K128*, const u32*, const u64*, const u32*, u32, K128*, u32*, const u64*, u32*, u32*, u16*, int, const u64*);

// Piece blockIdx.y: words_out[woff + w] = the 32 bases from base b0 + 32 w of the packed stream `src`, zero-filled past base
// b1, for w < ceil((b1 - b0) / 32); dir = {b0, b1, woff} per piece.  One launch for all pieces (a launch per segment was
// 100 x 6.5 us of launch overhead for 0.02 ms of copying).
__global__ void __launch_bounds__(GASM_WG) k_repack_reads(const u64* __restrict__ src, const u64* __restrict__ dir, u64* __restrict__ words_out) {
    const u64 b0 = dir[3 * blockIdx.y], b1 = dir[3 * blockIdx.y + 1], woff = dir[3 * blockIdx.y + 2];
    const u64 n_words = (b1 - b0 + 31) / 32;
    for (u64 w = (u64)blockIdx.x * GASM_WG + threadIdx.x; w < n_words; w += (u64)gridDim.x * GASM_WG) {
        const u64 p = b0 + 32 * w;
        u64 v = window32(src, p);
        const u64 left = b1 - p;
        if (left < 32) v &= ~0ull << (64 - 2 * left);
        words_out[woff + w] = v;
    }
}

// base position of every read of a pooled rank's own segments: read r lies in piece p = the last piece with
// piece_first[p] <= r, at word piece_word_off[p], its (r - piece_first[p])-th read of fixed_len bases
__global__ void __launch_bounds__(GASM_WG) k_piece_positions(const u64* __restrict__ piece_first, const u64* __restrict__ piece_word_off, u32 n_pieces,
                                                             u64 n_reads, u32 fixed_len, u64* __restrict__ pos) {
    for (u64 r = (u64)blockIdx.x * GASM_WG + threadIdx.x; r < n_reads; r += (u64)gridDim.x * GASM_WG) {
        const u32 p = upper_seg<u64>(piece_first, n_pieces, r);
        pos[r] = piece_word_off[p] * 32 + (r - piece_first[p]) * fixed_len;
    }
}

// out[s] = a[off[s + 1] - 1] (0 for an empty slice): the totals of per-segment inclusive scans
__global__ void __launch_bounds__(GASM_WG) k_slice_last(const u32* __restrict__ a, const u64* __restrict__ off, u32 n, u32* __restrict__ out) {
    const u32 s = blockIdx.x * GASM_WG + threadIdx.x;
    if (s < n) out[s] = off[s + 1] > off[s] ? a[off[s + 1] - 1] : 0u;
}

// ================================================================================================================
// Exchange plans (exchange.hip): everything the host layer of round 2 computed with numpy between two all-to-alls — which
// run goes where, at which record offset it lands, how many records every peer gets — is computed here, on the device,
// from run-length tables that never leave it.  The host sees one small report per exchange (the per-peer totals it must
// hand to ncclSend / ncclRecv, which take their counts from the host).
// ================================================================================================================
// Exclusive scan of get(0..n) with one workgroup of 1024 threads (running carry, 8 entries per thread and pass);
// put(i, exclusive prefix) for every i; returns the total to every thread.  s_wave: 16 u64 of LDS.
template <class G, class P>
__device__ __forceinline__ u64 wg_scan_seq(u32 n, u64* s_wave, G&& get, P&& put) {
    const u32 ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
    u64 carry = 0;
    for (u32 base = 0; base < n; base += 8192) {
        const u32 i0 = base + threadIdx.x * 8;
        u32 v[8];
        u64 sum = 0;
#pragma unroll
        for (u32 q = 0; q < 8; ++q) { v[q] = i0 + q < n ? get(i0 + q) : 0u; sum += v[q]; }
        u64 inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u64 o = __shfl_up(inc, d, 64);
            if ((int)ln >= d) inc += o;
        }
        if (ln == 63) s_wave[wv] = inc;
        __syncthreads();
        u64 before = 0, tot = 0;
#pragma unroll
        for (u32 w = 0; w < 16; ++w) { const u64 t = s_wave[w]; if (w < wv) before += t; tot += t; }
        __syncthreads();
        u64 ex = carry + before + inc - sum;
#pragma unroll
        for (u32 q = 0; q < 8; ++q) {
            if (i0 + q < n) put(i0 + q, ex);
            ex += v[q];
        }
        carry += tot;
    }
    return carry;
}

// the row a rank contributes to the all-gather of exchange 1: its NBT run lengths (already in place) + its flag word
__global__ void k_x_flag_word(const u32* __restrict__ flags, u32* __restrict__ row_tail) { if (threadIdx.x == 0) *row_tail = flags[0]; }

// Exchange 1 (every bucket's runs to the bucket's owner), rank r of W.  lens_all[s * stride + gb] = rank s's run length of
// bucket gb, [.. + nbt] = rank s's flag word.  order[0..nbt) = the buckets sorted by owner (dst_first[d] .. dst_first[d+1]:
// the buckets rank d merges, increasing); mine[0..n_mine) = the buckets this rank merges.
//   block s < W : what arrives from rank s — run_len[j * W + s], run_off[j * W + s] (records from the source's first), recv_tot[s]
//   block W     : what leaves — send_off[i] for order[i], send_tot[d]; flags_or = OR of all ranks' flag words
//   block W + 1 : capacity of the merged runs — bstart[j] = sum over j' < j of min(sum_s len, limit)
__global__ void __launch_bounds__(1024) k_x1_plan(const u32* __restrict__ lens_all, u64 stride, u32 nbt, const u32* __restrict__ order,
                                                  const u32* __restrict__ dst_first, const u32* __restrict__ mine, u32 n_mine, u32 W, u32 r, u32 limit,
                                                  u64* __restrict__ send_off, u64* __restrict__ send_tot, u64* __restrict__ run_off,
                                                  u32* __restrict__ run_len, u64* __restrict__ recv_tot, u64* __restrict__ bstart,
                                                  u32* __restrict__ flags_or) {
    __shared__ u64 s_wave[16];
    const u32 b = blockIdx.x;
    if (b < W) {
        const u32* row = lens_all + (u64)b * stride;
        const u64 tot = wg_scan_seq(n_mine, s_wave, [&](u32 j) { return row[mine[j]]; },
                                    [&](u32 j, u64 ex) { run_off[(u64)j * W + b] = ex; run_len[(u64)j * W + b] = row[mine[j]]; });
        if (threadIdx.x == 0) recv_tot[b] = tot;
    } else if (b == W) {
        const u32* row = lens_all + (u64)r * stride;
        const u64 tot = wg_scan_seq(nbt, s_wave, [&](u32 i) { return row[order[i]]; }, [&](u32 i, u64 ex) { send_off[i] = ex; });
        if (threadIdx.x == 0) send_off[nbt] = tot;
        __syncthreads();
        if (threadIdx.x < W) send_tot[threadIdx.x] = send_off[dst_first[threadIdx.x + 1]] - send_off[dst_first[threadIdx.x]];
        if (threadIdx.x == 0) {
            u32 f = 0;
            for (u32 s = 0; s < W; ++s) f |= lens_all[(u64)s * stride + nbt];
            *flags_or = f;
        }
    } else {
        const u64 tot = wg_scan_seq(n_mine, s_wave,
                                    [&](u32 j) {
                                        u64 sum = 0;
                                        for (u32 s = 0; s < W; ++s) sum += lens_all[(u64)s * stride + mine[j]];
                                        return (u32)(sum < limit ? sum : limit);
                                    },
                                    [&](u32 j, u64 ex) { bstart[j] = ex; });
        if (threadIdx.x == 0) bstart[n_mine] = tot;
    }
}

// the table every rank contributes to the all-reduce of exchange 2: merged length of the buckets it owns, zero elsewhere
// (the table is zeroed by a fill before), + its flag word at [nbt]
__global__ void __launch_bounds__(GASM_WG) k_x2_fill(u32* __restrict__ G, u32 nbt, const u32* __restrict__ mine, u32 n_mine,
                                                     const u32* __restrict__ bucket_d, const u32* __restrict__ flags) {
    const u32 j = blockIdx.x * GASM_WG + threadIdx.x;
    if (j < n_mine) G[mine[j]] = bucket_d[j];
    if (j == 0) G[nbt] = flags[0];
}

// Exchange 2 (the merged runs to their segment's owner), rank r of W.  G[gb] = merged length of bucket gb (all-reduced),
// own1[gb] = the rank that merged it; this rank's segments are [seg_first[r], seg_first[r + 1]) = buckets gb_lo .. gb_lo + n_out.
//   block s < W : run_len2[i * W + s] = (own1[gb_lo + i] == s) ? G : 0, run_off2 from the source's first record, recv_tot[s]
//   block W     : send_off2[j] of this rank's merged run j (bucket mine[j]); send_tot[d] (runs bound for one rank are consecutive)
//   block W + 1 : bstart2[i] (exact), the rank's distinct k-mers in all (info[0]) and of its largest segment (info[1])
__global__ void __launch_bounds__(1024) k_x2_plan(const u32* __restrict__ G, const u16* __restrict__ own1, u32 W, u32 r, u32 gb_lo, u32 n_out, int bbits,
                                                  const u32* __restrict__ mine, u32 n_mine, const u32* __restrict__ seg_first,
                                                  u64* __restrict__ send_off, u64* __restrict__ send_tot, u64* __restrict__ run_off,
                                                  u32* __restrict__ run_len, u64* __restrict__ recv_tot, u64* __restrict__ bstart, u64* __restrict__ info) {
    __shared__ u64 s_wave[16];
    __shared__ u32 s_max;
    const u32 b = blockIdx.x;
    if (b < W) {
        const u64 tot = wg_scan_seq(n_out, s_wave, [&](u32 i) { return own1[gb_lo + i] == b ? G[gb_lo + i] : 0u; },
                                    [&](u32 i, u64 ex) { run_off[(u64)i * W + b] = ex; run_len[(u64)i * W + b] = own1[gb_lo + i] == b ? G[gb_lo + i] : 0u; });
        if (threadIdx.x == 0) recv_tot[b] = tot;
    } else if (b == W) {
        const u64 tot = wg_scan_seq(n_mine, s_wave, [&](u32 j) { return G[mine[j]]; }, [&](u32 j, u64 ex) { send_off[j] = ex; });
        if (threadIdx.x == 0) send_off[n_mine] = tot;
        __syncthreads();
        if (threadIdx.x < W) {
            const u32 d = threadIdx.x;
            const u32 lo = lower_bound_dev<u32>(mine, 0, n_mine, seg_first[d] << bbits), hi = lower_bound_dev<u32>(mine, 0, n_mine, seg_first[d + 1] << bbits);
            send_tot[d] = send_off[hi] - send_off[lo];
        }
    } else {
        if (threadIdx.x == 0) s_max = 0;
        const u64 tot = wg_scan_seq(n_out, s_wave, [&](u32 i) { return G[gb_lo + i]; }, [&](u32 i, u64 ex) { bstart[i] = ex; });
        const u32 nb = 1u << bbits;
        u32 mx = 0;
        for (u32 s = threadIdx.x; s < n_out / nb; s += 1024) {
            u64 d = 0;
            for (u32 q = 0; q < nb; ++q) d += G[gb_lo + s * nb + q];
            mx = max(mx, (u32)min(d, (u64)0xFFFFFFFFu));
        }
        atomicMax(&s_max, mx);
        __syncthreads();
        if (threadIdx.x == 0) { bstart[n_out] = tot; info[0] = tot; info[1] = s_max; }
    }
}

// The report of an exchange plan: W send totals, W receive totals, up to four more words, then the ticket (pinned memory).
__global__ void __launch_bounds__(64) k_x_report(const u64* __restrict__ send_tot, const u64* __restrict__ recv_tot, u32 W, const u64* __restrict__ info, u32 n_info,
                                                 const u32* __restrict__ flags_or, u64* __restrict__ report, u64 ticket) {
    for (u32 i = threadIdx.x; i < W; i += 64) { report[i] = send_tot[i]; report[W + i] = recv_tot[i]; }
    if (threadIdx.x < n_info) report[2 * W + threadIdx.x] = info[threadIdx.x];
    if (threadIdx.x == 0) report[2 * W + 4] = flags_or ? (u64)*flags_or : 0ull;
    __threadfence_system();
    if (threadIdx.x == 0) report[2 * W + 5] = ticket;
}

// sum of n u32 arrays, element by element (the all-reduce of the virtual communicator)
__global__ void __launch_bounds__(GASM_WG) k_x_add_u32(u32* __restrict__ acc, const u32* __restrict__ v, u64 n) {
    for (u64 i = (u64)blockIdx.x * GASM_WG + threadIdx.x; i < n; i += (u64)gridDim.x * GASM_WG) acc[i] += v[i];
}
