// kernels_build.hip — graph-construction kernels (gfx950): packing, k-mer bucket partition, LDS hash de-duplication,
// (k-1)-mer graph, list ranking, contig emission.  Integer / hash / gather work: no MFMA anywhere; the levers are
// coalesced HBM streams, LDS-resident tables and wave-level primitives.
//
// What each kernel restates of the reference (paths relative to the reference root):
//   k_pack_ascii      new surface (the reference keeps std::string)
//   k_bucket_partition
//                     lib/DeNovoAssembler.R:109-130 (every k-mer of every read) fused with the first half of the
//                     de-duplication that lib/DeNovoAssembler.cpp:104-122 does through its hash map: one pass into
//                     per-(segment, bucket) regions
//   k_tile_hist / k_tile_scan /
//   k_bucket_scatter  the same in two passes with an exact layout (the retry path of the one above)
//   k_bucket_dedup    lib/DeNovoAssembler.cpp:104-122 (distinct edges) + multiplicities
//   k_edge_target / k_edge_multi / k_node_flags
//                     lib/DeNovoAssembler.cpp:125-169 (in/out degree over distinct edges, branching nodes)
//   k_edge_next       lib/DeNovoAssembler.cpp:172-189, one step of the walk: successor edge or stop
//   k_rank_rulers / k_rank_lds / k_link_jump
//                     the walk itself as list ranking (the reference walks node by node): every second edge ranked by
//                     pointer doubling inside LDS, the rest finished by a step or two; whole-GPU doubling for
//                     segments of more than 65534 edges
//   k_chain_len / k_contig_scan (+ seg_offsets_wave: directories and report) / k_contig_place / k_contig_emit
//                     lib/DeNovoAssembler.cpp:183-192: contig text, in sorted order (contigs start with distinct
//                     k-mers, so sorting contigs = sorting their first edges)
#include "device_utils.h"
#include "keyops.h"
#include "kernels.h"
#include "dedup_order.h"

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

// The key array is written once (partition) and read once (de-duplication): 1.6 GB per step that nothing touches twice.  Both
// accesses are marked non-temporal (`nt`).  Measured on MI355X, cfg2, same box back to back (NOTES_r3.md): plain 1.168 ms/step
// (de-duplication 0.455 ms); nt loads 1.103 (0.393); nt stores 1.113 (0.401 — the stores are the partition's, the gain is the
// de-duplication's: it no longer reads through dirty lines the previous kernel left in L2 / Infinity Cache); both 1.081 (0.378).
// Other policies on the stores (sc1, sc0 sc1, sc1 nt, sc0 sc1 nt, sc0 nt, inline asm) were all slower than plain nt.
// GASM_NO_NT (compile time) restores plain accesses.
typedef unsigned int u32x4nt __attribute__((ext_vector_type(4)));
#ifndef GASM_NO_NT
__device__ __forceinline__ uint4 stream_load16(const uint4* p) {
    const u32x4nt v = __builtin_nontemporal_load(reinterpret_cast<const u32x4nt*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
#define GASM_STREAM_LOAD(p) stream_load16(p)
#define GASM_STREAM_STORE(v, p) __builtin_nontemporal_store((v), (p))
#else
#define GASM_STREAM_LOAD(p) (*(p))
#define GASM_STREAM_STORE(v, p) (*(p) = (v))
#endif

// ================================================================================================================
// ASCII -> 2-bit.  One thread per output word (32 bases) and grid-stride round; 2 x 16-byte loads where the word is fully
// inside.  nbases_dev (optional): the number of bases lives on the device (the contigs of a build the host has not
// waited for).  Four zero padding words follow the last base.
// ================================================================================================================
__global__ void __launch_bounds__(GASM_WG) k_pack_ascii(const u8* __restrict__ ascii, u64 nbases_host, const u64* __restrict__ nbases_dev,
                                                        u64* __restrict__ words, u32* __restrict__ err) {
    const u64 nbases = nbases_dev ? *nbases_dev : nbases_host;
    const u64 nwords = (nbases + 31) / 32 + 4;
    bool ok = true;
    for (u64 t = (u64)blockIdx.x * GASM_WG + threadIdx.x; t < nwords; t += (u64)gridDim.x * GASM_WG) {
        const u64 b0 = t << 5;
        u64 w = 0;
        if (b0 + 32 <= nbases) {
            const uint4* src = reinterpret_cast<const uint4*>(ascii + b0);
            const uint4 v0 = src[0], v1 = src[1];
            const u32 d[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const u8 c = (u8)(d[i] >> (8 * j));
                    ok = ok && base_ok(c);
                    w = (w << 2) | base_code(c);
                }
            }
        } else {
            for (int j = 0; j < 32; ++j) {
                u8 c = 'A';
                if (b0 + j < nbases) { c = ascii[b0 + j]; ok = ok && base_ok(c); }
                w = (w << 2) | base_code(c);
            }
        }
        words[t] = w;
    }
    if (!ok && err) atomicOr(err, 1u);
}

// ================================================================================================================
// Tiles.  A tile is one workgroup-pass of the two tile kernels: up to ipt = GASM_TILE_WG / g consecutive reads of one
// segment at one offset round o; the g threads that share a read take KT consecutive k-mer starts each, at offsets
// o*g*KT + lane*KT + [0, KT) (KT = 16 for 64-bit keys, 8 for 128-bit keys; o > 0 only for reads with more than g*KT
// k-mers).  The windows come out of a few 64-bit words held in registers (Roll<K>, keyops.h), so a k-mer costs a funnel
// shift, not two loads.  The host builds the tile table (DevReads::set_tiles): one uniform 16-byte load per tile.
// ================================================================================================================
struct TileInfo { u32 seg; u64 r0; u32 nitems; u32 o; };

// `tinfo` is the kernels' own __restrict__ copy of rs.tile_info: only then may the compiler use a scalar load (the
// scalar cache is not coherent with the kernel's vector stores, so it needs the no-alias guarantee)
__device__ __forceinline__ TileInfo tile_decode(const uint4* __restrict__ tinfo, u32 tile) {
    const uint4 e = tinfo[tile];               // uniform across the workgroup
    TileInfo ti;
    ti.seg = e.x;
    ti.nitems = e.y;
    ti.r0 = (u64)e.z | ((u64)(e.w & 0xFFFFu) << 32);
    ti.o = e.w >> 16;
    return ti;
}

// The words thread `tid` needs in tile `ti`, and how many of its KT starts are k-mers.
template <class K>
__device__ __forceinline__ void tile_fetch(const ReadSet& rs, const TileInfo& ti, u32 g, int k, Roll<K>& r, u32& nv) {
    constexpr u32 KT = KeyTraits<K>::KT;
    const u32 item = threadIdx.x / g, lane = threadIdx.x % g;
    u64 p0 = 0;
    u32 nk = 0;
    if (item < ti.nitems) {
        u32 len;
        read_span(rs, ti.r0 + item, &p0, &len);
        nk = len >= (u32)k ? len - k + 1 : 0;
    }
    const u32 off0 = ti.o * g * KT + lane * KT;
    nv = off0 < nk ? min(nk - off0, KT) : 0u;
    r.load(rs.words, nv ? p0 + off0 : 0);
}

// Bucket counts of every tile: tcnt[tile * nb + b] = k-mers of the tile whose first `bbits` bits are b, as four 16-bit
// sub-counts (a tile holds at most GASM_TILE_WG * KT <= 8192): sub-count s is what the threads with (thread & 3) == s
// contribute.  The same split is used when k_bucket_scatter ranks the k-mers; with it 64 lanes hitting 64 buckets
// collide on an LDS word a quarter as often (same-address LDS atomics serialise: 12.4 vs 8.4 cycles per instruction).  Buckets are key ranges (bbits <= 2(k-1), bbits <= 10): concatenating
// sorted buckets gives a sorted segment.
template <class K>
__global__ void __launch_bounds__(GASM_TILE_WG) k_tile_hist(ReadSet rs, const uint4* __restrict__ tinfo, int k, int bbits, u32 g, u32 n_tiles,
                                                            ushort4* __restrict__ tcnt) {
    constexpr u32 KT = KeyTraits<K>::KT;
    extern __shared__ u32 s_h[];   // [nb][4]
    const u32 nb = 1u << bbits, sub = threadIdx.x & 3u;
    for (u32 e = threadIdx.x; e < 4 * nb; e += GASM_TILE_WG) s_h[e] = 0;
    __syncthreads();
    // the next tile's words are requested before this tile's atomics (one tile of load latency hidden per tile)
    u32 tile = blockIdx.x;
    if (tile >= n_tiles) return;
    Roll<K> r;
    u32 nv;
    tile_fetch<K>(rs, tile_decode(tinfo, tile), g, k, r, nv);
    for (;;) {
        const auto w = r.prep();
        const u32 nv_now = nv;
        const u32 tnext = tile + gridDim.x;
        if (tnext < n_tiles) tile_fetch<K>(rs, tile_decode(tinfo, tnext), g, k, r, nv);
        static_for<KT>([&](auto J) {
            constexpr u32 j = J;
            if (j < nv_now) atomicAdd(&s_h[(bbits ? w.template top_hi<j>() >> (32 - bbits) : 0u) * 4 + sub], 1u);
        });
        __syncthreads();
        for (u32 e = threadIdx.x; e < nb; e += GASM_TILE_WG) {
            const uint4 c = *reinterpret_cast<const uint4*>(&s_h[4 * e]);
            tcnt[(u64)tile * nb + e] = make_ushort4((u16)c.x, (u16)c.y, (u16)c.z, (u16)c.w);
            *reinterpret_cast<uint4*>(&s_h[4 * e]) = make_uint4(0, 0, 0, 0);
        }
        __syncthreads();
        if (tnext >= n_tiles) break;
        tile = tnext;
    }
}
template __global__ void k_tile_hist<u64>(ReadSet, const uint4*, int, int, u32, u32, ushort4*);
template __global__ void k_tile_hist<K128>(ReadSet, const uint4*, int, int, u32, u32, ushort4*);

// Per segment and bucket: the running sum toff[tile * nb + b] of the tiles' padded counts (offset of the tile's run
// inside its (segment,bucket) range) and the bucket total hist[seg * nb + b].  One workgroup per segment, thread =
// bucket.  Each (tile, bucket) run is rounded up to padm + 1 keys — one 128-byte line when the bucket count allows —
// so that no cache line is written by two workgroups: partial-line writes cost more than half the store bandwidth.
__device__ __forceinline__ u32 sub_total(ushort4 c) { return (u32)c.x + c.y + c.z + c.w; }
__global__ void __launch_bounds__(1024) k_tile_scan(ReadSet rs, int bbits, u32 padm, const ushort4* __restrict__ tcnt,
                                                    u32* __restrict__ toff, u32* __restrict__ hist, u32* __restrict__ flags) {
    __shared__ u32 s_part[1024];
    const u32 nb = 1u << bbits, seg = blockIdx.x;
    // the build's 64 flag words start at zero (a fill would be a launch of its own; the first writer is k_bucket_dedup)
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64) flags[threadIdx.x] = 0;
    const u32 t0 = rs.seg_tile_start[seg], t1 = rs.seg_tile_start[seg + 1];
    // a workgroup = 32 buckets (blockIdx.y) x 32 slices of the segment's tile range: slice sums first, then the offsets
    const u32 nbw = min(nb, 32u), grp = 1024u / nbw, sl = threadIdx.x / nbw, bl = threadIdx.x % nbw;
    const u32 per = (t1 - t0 + grp - 1) / grp;
    const u32 ta = min(t1, t0 + sl * per), tb = min(t1, ta + per);
    const u32 b = blockIdx.y * nbw + bl;
    u32 sum = 0;
    for (u32 t = ta; t < tb; ++t) sum += (sub_total(tcnt[(u64)t * nb + b]) + padm) & ~padm;
    s_part[threadIdx.x] = sum;
    __syncthreads();
    u32 run = 0;
    for (u32 s = 0; s < sl; ++s) run += s_part[s * nbw + bl];
    if (sl == grp - 1) hist[(u64)seg * nb + b] = run + sum;
    for (u32 t = ta; t < tb; ++t) {
        toff[(u64)t * nb + b] = run;
        run += (sub_total(tcnt[(u64)t * nb + b]) + padm) & ~padm;
    }
}

// Exclusive scan of n u32 into n+1 outputs of type TO; one workgroup of 1024 threads, running carry.
template <class TO>
__global__ void __launch_bounds__(1024) k_scan_excl(const u32* __restrict__ in, TO* __restrict__ out, u32 n) {
    __shared__ u64 s_wave[16];
    const u32 ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
    u64 carry = 0;
    // eight consecutive entries per thread and pass: one scan of the workgroup (two barriers) per 8192 entries.  The
    // partial sums are 64-bit throughout (a pass may cover billions of k-mers).
    for (u32 base = 0; base < n; base += 8192) {
        const u32 i0 = base + threadIdx.x * 8;
        u32 v[8];
        u64 sum = 0;
#pragma unroll
        for (u32 q = 0; q < 8; ++q) { v[q] = i0 + q < n ? in[i0 + q] : 0u; sum += v[q]; }
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
            if (i0 + q < n) out[i0 + q] = (TO)ex;
            ex += v[q];
        }
        carry += tot;
    }
    if (threadIdx.x == 0) out[n] = (TO)carry;
}
template __global__ void k_scan_excl<u64>(const u32*, u64*, u32);
template __global__ void k_scan_excl<u32>(const u32*, u32*, u32);

// Exclusive scans of the per-segment contig counts and bases (k_contig_scan) into the segment directories — and the
// build's REPORT: the last kernel of a build writes everything the host needs afterwards into the batch's pinned report
// area in one go, the ticket last.  Nothing before this point makes the host wait: the build is queued in full with
// sizes taken from upper bounds, and the host reads the report when somebody asks for results (pipeline_build_finish).
//   report[0 .. S]            first distinct k-mer of every segment (+ the total)
//   report[S+1 .. 2S+1]       first contig of every segment (+ the total)
//   report[2S+2 .. 4S+3]      first contig base of every segment (+ the total), (lo, hi) pairs
//   report[4S+4]              flags[0]: a bucket overflowed its table        report[4S+5]  flags[1]: list ranking gave up
//   report[4S+6]              ticket
// (body: one wave; `seg_ncontig` / `seg_cbases` are read with agent-scope loads — in k_contig_scan's last workgroup they come
// from workgroups of other XCDs, whose L2 this one does not share)
__device__ __forceinline__ void seg_offsets_wave(const u32* __restrict__ seg_ncontig, const u64* __restrict__ seg_cbases, u32 S,
                                                 u32* __restrict__ seg_cstart, u64* __restrict__ seg_bstart,
                                                 const u32* __restrict__ dstart, u32 nb, const u32* __restrict__ flags,
                                                 u32* __restrict__ report, u32 ticket) {
    // one wave, 64 segments at a time (a batch rarely has more than a few hundred segments)
    const u32 ln = threadIdx.x & 63;
    u32 ccarry = 0;
    u64 bcarry = 0;
    u32* const rc = report + S + 1;
    u32* const rb = report + 2 * S + 2;
    for (u32 base = 0; base < S; base += 64) {
        const u32 i = base + ln;
        const u32 c = i < S ? __hip_atomic_load(&seg_ncontig[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        const u64 b = i < S ? __hip_atomic_load(&seg_cbases[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        const u32 cinc = wave_incl_scan(c);
        u64 binc = b;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u64 o = __shfl_up(binc, d, 64);
            if ((int)ln >= d) binc += o;
        }
        if (i < S) {
            const u32 cs = ccarry + cinc - c;
            const u64 bs = bcarry + binc - b;
            seg_cstart[i] = cs; seg_bstart[i] = bs;
            rc[i] = cs;
            rb[2 * i] = (u32)bs; rb[2 * i + 1] = (u32)(bs >> 32);
            report[i] = dstart[(u64)i * nb];
        }
        ccarry += wave_last(cinc);
        bcarry += __shfl(binc, 63, 64);
    }
    if (ln == 0) {
        seg_cstart[S] = ccarry; seg_bstart[S] = bcarry;
        rc[S] = ccarry;
        rb[2 * S] = (u32)bcarry; rb[2 * S + 1] = (u32)(bcarry >> 32);
        report[S] = dstart[(u64)S * nb];
        report[4 * S + 4] = flags[0];
        report[4 * S + 5] = flags[1];
    }
    __threadfence_system();                    // (one wave: every lane's report words are out before lane 0's ticket)
    if (ln == 0) report[4 * S + 6] = ticket;
}

// ================================================================================================================
// Scatter the k-mers of every tile into their (segment, bucket) ranges — whole cache lines only.
//
// The workgroup (512 threads = 8 waves) stages one tile in LDS, bucket by bucket: the buckets' staging ranges are
// the exclusive scan of the tile's padded counts (k_tile_hist), a k-mer's slot inside its range comes from an LDS
// atomic on the bucket's cursor, and the padding slots hold filler keys (top bit set + the bucket's prefix; the
// de-duplication skips them).  Staging index i then maps to global index comb[bucket] + i, and because staging ranges
// and global runs are both multiples of a line, 16 consecutive lanes (8 for 128-bit keys) write exactly one aligned
// 128-byte line.  Measured on MI355X: runs that start at arbitrary 16-byte offsets reach ~3 TB/s, aligned lines
// ~5.9 TB/s — with the stores pointed at an L2-resident scratch area the kernel ran in 0.24 ms instead of 0.57 ms, so
// it is the write pattern at the memory side, not the CU side, that sets the time.
//
// Memory operations of a wave retire in order (one counter, vmcnt, for loads and stores).  A load issued after the
// flush would therefore wait for all of the flush's stores.  So the next tile's inputs (bucket counts, bases, read
// words) are requested BEFORE the flush, and the flush issues exactly NFL stores per thread whatever the tile holds
// (lanes past the end write to a per-wave scratch slot): the wait at the top of the next tile is then "all but the
// last NFL operations", written out explicitly.
// LDS: NFL * 512 x 16 bytes (72 KB) + 64 trash slots + nb * 24 + 96 -> two workgroups per CU, for both key widths.
// ================================================================================================================
template <class K> struct TilePrefetch {
    Roll<K> rl;
    u32 nv, c01, c23, toff, bs_lo, bs_hi; // sub-counts packed as loaded (the sums are formed after the wait, not at the request)
    // explicit wait: everything requested at least N vector-memory operations ago has arrived
    template <int N> __device__ __forceinline__ void wait_all_but() {
        rl.template wait_all_but<N>();
        __asm__ volatile("" : "+v"(c01), "+v"(c23), "+v"(toff), "+v"(bs_lo), "+v"(bs_hi));     // ordered behind the wait above ("memory")
    }
};

template <class K>
__global__ void __launch_bounds__(GASM_TILE_WG, 1024 / GASM_TILE_WG) k_bucket_scatter(ReadSet rs, const uint4* __restrict__ tinfo, int k, int bbits, u32 g, u32 padm,
                                                                  u32 n_tiles, const u64* __restrict__ bstart, const u32* __restrict__ toff,
                                                                  const ushort4* __restrict__ tcnt, K* __restrict__ keys, u64 scratch) {
    extern __shared__ __align__(16) unsigned char s_raw[];
    constexpr u32 KT = KeyTraits<K>::KT;
    constexpr u32 NFL = KeyTraits<K>::NFL;               // flush passes of 16 bytes per thread: the tile + room for the padding
    constexpr u32 KPU = 16 / sizeof(K);                  // keys per 16-byte unit
    constexpr u32 CAP = NFL * GASM_TILE_WG * KPU;
    const u32 nb = 1u << bbits;
    const u32 tid = threadIdx.x, wv = tid >> 6, ln = tid & 63, sub = tid & 3u;
    K* s_key = reinterpret_cast<K*>(s_raw);                                  // CAP + eight trash slots per wave
    u64* s_comb = reinterpret_cast<u64*>(s_key + CAP + GASM_TILE_WG / 8);    // nb
    u32* s_cur = reinterpret_cast<u32*>(s_comb + ((nb + 1) & ~1u));          // 4 per bucket (sub-cursors) + dummy bin; 16-byte aligned
    u32* s_tmp = s_cur + 4 * nb + 4;                                         // 12
    const int bshift = 2 * k - bbits;
    const bool one = nb <= GASM_TILE_WG;                                     // one bucket per thread, prefetched

    // a workgroup takes a contiguous range of tiles (neighbouring runs of a bucket then come from the same L2)
    const u32 per_wg = (n_tiles + gridDim.x - 1) / gridDim.x;
    const u32 tile_end = min(n_tiles, (blockIdx.x + 1) * per_wg);
    u32 tile = blockIdx.x * per_wg;
    if (tile >= tile_end) return;
    const u64 my_scratch = scratch + (u64)((blockIdx.x & 1023u) * (GASM_TILE_WG / 64) + wv) * KPU;     // one 16-byte slot per wave (see k_bucket_partition)

    auto fetch = [&](u32 t, const TileInfo& ti, TilePrefetch<K>& pf) {
        pf.c01 = 0; pf.c23 = 0; pf.toff = 0; pf.bs_lo = 0; pf.bs_hi = 0;
        if (one && tid < nb) {
            const uint2 c = reinterpret_cast<const uint2*>(tcnt)[(u64)t * nb + tid];
            pf.c01 = c.x; pf.c23 = c.y;
            pf.toff = toff[(u64)t * nb + tid];
            const uint2 b = reinterpret_cast<const uint2*>(bstart)[(u64)ti.seg * nb + tid];
            pf.bs_lo = b.x; pf.bs_hi = b.y;
        }
        tile_fetch<K>(rs, ti, g, k, pf.rl, pf.nv);
    };
    TileInfo ti = tile_decode(tinfo, tile);
    TilePrefetch<K> pf;
    fetch(tile, ti, pf);
    pf.template wait_all_but<0>();
    for (;;) {
        // ---- staging ranges of the buckets
        u32 total;
        if (one) {
            const u32 c0 = pf.c01 & 0xFFFFu, c1 = pf.c01 >> 16, c2 = pf.c23 & 0xFFFFu, c3 = pf.c23 >> 16;
            const u32 cnt = c0 + c1 + c2 + c3, padc = (cnt + padm) & ~padm;
            const u32 soff = block_excl_scan<GASM_TILE_WG>(padc, s_tmp, &total);
            if (tid < nb) {
                *reinterpret_cast<uint4*>(&s_cur[4 * tid]) = make_uint4(soff, soff + c0, soff + c0 + c1, soff + c0 + c1 + c2);
                s_comb[tid] = (((u64)pf.bs_hi << 32) | pf.bs_lo) + pf.toff - soff;   // staging index i goes to keys[s_comb[bucket] + i]
                for (u32 i = cnt; i < padc; ++i) s_key[soff + i] = key_filler<K>(tid, bshift);
            }
        } else {
            u32 carry = 0;
            for (u32 b0 = 0; b0 < nb; b0 += GASM_TILE_WG) {
                const u32 b = b0 + tid;
                const ushort4 c = tcnt[(u64)tile * nb + b];
                const u32 cnt = sub_total(c), padc = (cnt + padm) & ~padm;
                u32 tot;
                const u32 soff = carry + block_excl_scan<GASM_TILE_WG>(padc, s_tmp, &tot);
                *reinterpret_cast<uint4*>(&s_cur[4 * b]) = make_uint4(soff, soff + c.x, soff + c.x + c.y, soff + c.x + c.y + c.z);
                s_comb[b] = bstart[(u64)ti.seg * nb + b] + toff[(u64)tile * nb + b] - soff;
                for (u32 i = cnt; i < padc; ++i) s_key[soff + i] = key_filler<K>(b, bshift);
                carry += tot;
            }
            total = carry;
        }
        __syncthreads();
        // ---- rank and stage.  The ds_add_rtn of a thread are issued back to back (a start past the end of the read
        // ranks into a dummy bin) and waited for once; a branch per k-mer would make them dependent LDS round trips.
        const auto w = pf.rl.prep();
        const u32 nv = pf.nv;
        K key[KT];
        u32 idx[KT];
        static_for<KT>([&](auto J) {
            constexpr u32 j = J;
            key[j] = w.template key<j>(k);
            const u32 bkt = bbits ? w.template top_hi<j>() >> (32 - bbits) : 0u;
            idx[j] = atomicAdd(&s_cur[j < nv ? 4 * bkt + sub : 4 * nb], 1u);
        });
        // ---- the next tile's inputs, requested ahead of this tile's stores (the last tile re-requests its own)
        const u32 tnext = tile + 1 < tile_end ? tile + 1 : tile;
        const TileInfo tin = tile_decode(tinfo, tnext);
        fetch(tnext, tin, pf);
        // (a start past the end of the read goes to a trash slot: cheaper than a branch per k-mer.  Eight slots per wave,
        // not one per thread: with 128-bit keys the per-thread slots were the 8 KB that kept a second workgroup off the CU)
        const u32 trash = CAP + wv * 8 + (ln & 7u);
#pragma unroll
        for (u32 j = 0; j < KT; ++j) s_key[j < nv ? idx[j] : trash] = key[j];
        __syncthreads();
        // ---- stream out: NFL 16-byte stores per thread (a pair of 64-bit keys or one 128-bit key; staged runs are even,
        // so a pair never straddles two buckets), three at a time so that their LDS reads overlap
        static_assert(NFL % 3 == 0, "flush passes come in threes");
#pragma unroll
        for (u32 u0 = 0; u0 < NFL; u0 += 3) {
            u64x2 w[3];
            u64 cb[3];
#pragma unroll
            for (u32 u = 0; u < 3; ++u) w[u] = *reinterpret_cast<const u64x2*>(s_key + (tid + GASM_TILE_WG * (u0 + u)) * KPU);
#pragma unroll
            for (u32 u = 0; u < 3; ++u) {
                u32 bkt = 0;
                if (bbits) {
                    if constexpr (KPU == 2) bkt = (u32)(w[u].x >> bshift) & (nb - 1);
                    else bkt = kfield(K128{w[u].x, w[u].y}, bshift) & (nb - 1);
                }
                cb[u] = s_comb[bkt];
            }
#pragma unroll
            for (u32 u = 0; u < 3; ++u) {
                const u32 i = (tid + GASM_TILE_WG * (u0 + u)) * KPU;
                GASM_STREAM_STORE(w[u], reinterpret_cast<u64x2*>(keys + (i < total ? cb[u] + i : my_scratch)));   // (no branch: the number of stores must not vary)
            }
        }
        if (++tile >= tile_end) break;
        __syncthreads();          // staging is free again
        ti = tin;
        pf.template wait_all_but<NFL>();
    }
}
template __global__ void k_bucket_scatter<u64>(ReadSet, const uint4*, int, int, u32, u32, u32, const u64*, const u32*, const ushort4*, u64*, u64);
template __global__ void k_bucket_scatter<K128>(ReadSet, const uint4*, int, int, u32, u32, u32, const u64*, const u32*, const ushort4*, K128*, u64);

// ================================================================================================================
// Single-pass partition: k_tile_hist + k_tile_scan + k_scan_excl + k_bucket_scatter in one kernel.
//
// The two-pass form above needs every (tile, bucket) count before the first key moves — a second pass over the reads that
// does nothing but count (2·10^8 LDS atomics for cfg2: 0.07 ms, plus two scans).  Here every (segment, bucket) owns a
// region of fixed capacity (bstart[gb] .. bstart[gb + 1], sized by the host from the segment's k-mer count with some room
// to spare) and a cursor; a tile counts its k-mers per bucket in LDS — the ds_add_rtn that counts IS the rank of the k-mer
// in its (bucket, sub-counter) — then reserves its line-padded runs with one global atomic per bucket and flushes whole
// lines as before.  Runs of a bucket therefore lie in the order the tiles got there, which the de-duplication does not
// care about (its result is ordered and counted).  cursor[gb] ends up as the bucket's padded length (what `hist` is in the
// two-pass form).  A run that does not fit its region goes to the scratch line, bit 1 of flags[0] is raised, the
// de-duplication leaves such a bucket empty, and pipeline_build_finish repeats the build with the two-pass kernels.
// Memory operations of a wave: [next tile's words, region bounds] [cursor atomic: wave(s) of the bucket threads]
// [NFL stores]; the atomic's value is used before the flush, so the wait at the top stays "all but the NFL stores".
// LDS: staging as above + nb * 44 + 80 bytes: two workgroups per CU up to 128 buckets.
// ================================================================================================================
template <class K>
__global__ void __launch_bounds__(GASM_TILE_WG, 4) k_bucket_partition(ReadSet rs, const uint4* __restrict__ tinfo, int k, int bbits, u32 g, u32 padm,
                                                                    u32 n_tiles, const u64* __restrict__ bstart, u32* __restrict__ cursor,
                                                                    K* __restrict__ keys, u64 scratch, u32* __restrict__ flags) {
    extern __shared__ __align__(16) unsigned char s_raw[];
    constexpr u32 KT = KeyTraits<K>::KT;
    constexpr u32 NFL = KeyTraits<K>::NFL;
    constexpr u32 KPU = 16 / sizeof(K);
    constexpr u32 CAP = NFL * GASM_TILE_WG * KPU;
    const u32 nb = 1u << bbits;                                               // <= GASM_TILE_WG: one bucket per thread
    const u32 tid = threadIdx.x, wv = tid >> 6, ln = tid & 63, sub = tid & 3u;
    K* s_key = reinterpret_cast<K*>(s_raw);                                  // CAP + eight trash slots per wave
    u64* s_comb = reinterpret_cast<u64*>(s_key + CAP + GASM_TILE_WG / 8);    // nb
    u32* s_base = reinterpret_cast<u32*>(s_comb + ((nb + 1) & ~1u));         // 4 per bucket: staging index of rank 0 of every sub-counter
    u32* s_cnt = s_base + 4 * nb + 4;                                        // 4 per bucket (sub-counters) + dummy bin; 16-byte aligned
    u32* s_tmp = s_cnt + 4 * nb + 4;                                         // 12
    u32* s_fill = s_tmp + 12;                                                // nb: where a bucket's fillers start << 4 | how many
    const int bshift = 2 * k - bbits;

    const u32 per_wg = (n_tiles + gridDim.x - 1) / gridDim.x;
    const u32 tile_end = min(n_tiles, (blockIdx.x + 1) * per_wg);
    u32 tile = blockIdx.x * per_wg;
    if (tile >= tile_end) return;
    for (u32 e = tid; e < 4 * nb + 4; e += GASM_TILE_WG) s_cnt[e] = 0;

    Roll<K> rl;
    u32 nv;
    auto fetch = [&](const TileInfo& t) { tile_fetch<K>(rs, t, g, k, rl, nv); };
    TileInfo ti = tile_decode(tinfo, tile);
    fetch(ti);
    rl.template wait_all_but<0>();
    __syncthreads();
    for (;;) {
        // ---- count = rank
        const auto w = rl.prep();
        const u32 nv_now = nv;
        const u32 seg = ti.seg;
        // (the keys themselves are taken from the window again when they are staged — two funnel shifts each — rather than
        // kept in 32 registers across the two barriers: with them the kernel does not fit the 128 registers of two workgroups per CU)
        u32 idx[KT];
        static_for<KT>([&](auto J) {
            constexpr u32 j = J;
            const u32 bkt = bbits ? w.template top_hi<j>() >> (32 - bbits) : 0u;
            idx[j] = atomicAdd(&s_cnt[j < nv_now ? 4 * bkt + sub : 4 * nb], 1u);
        });
        // ---- the next tile's inputs, requested ahead of this tile's stores (the last tile re-requests its own)
        const u32 tnext = tile + 1 < tile_end ? tile + 1 : tile;
        const TileInfo tin = tile_decode(tinfo, tnext);
        fetch(tin);
        __syncthreads();
        // ---- staging ranges of the buckets, and the reservation of their runs.  `t` is the thread index again, opaque to
        // the compiler: everything a bucket thread addresses through it (counters, bases, cursor, region, scratch line) is then
        // computed here, a few instructions per tile, instead of being kept in ~20 registers over the whole loop — which
        // spilled, and a spill reload is a vector-memory load: its wait also waits for the next tile's words and the stores
        u32 t = tid, zero = 0;
        __asm__ volatile("" : "+v"(t), "+v"(zero));
        u32 c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        if (t < nb) {
            const uint4 c = *reinterpret_cast<const uint4*>(&s_cnt[4 * t]);
            c0 = c.x; c1 = c.y; c2 = c.z; c3 = c.w;
        }
        const u32 cnt = c0 + c1 + c2 + c3, padc = (cnt + padm) & ~padm;
        u32 total;
        const u32 soff = block_excl_scan_open<GASM_TILE_WG>(padc, s_tmp, &total);      // (s_tmp is next written two barriers on)
        u32 run = 0;
        u64 beg = 0, end = 0;                      // the region of bucket `tid` (a few KB of directory per segment: L2 hits, used after the staging)
        if (t < nb) {
            const u32 gb = seg * nb + t;
            if (padc) run = atomicAdd(&cursor[gb], padc);
            beg = bstart[gb]; end = bstart[gb + 1];
            *reinterpret_cast<uint4*>(&s_base[4 * t]) = make_uint4(soff, soff + c0, soff + c0 + c1, soff + c0 + c1 + c2);
            *reinterpret_cast<uint4*>(&s_cnt[4 * t]) = make_uint4(zero, zero, zero, zero);
            s_fill[t] = ((soff + cnt) << 4) | (padc - cnt);
        }
        __syncthreads();
        // ---- fillers of the padded runs: eight threads per bucket (one wave writing up to 15 fillers for each of its buckets
        // one after the other kept the other seven waiting at the barrier above)
        for (u32 x = t; x < 8 * nb; x += GASM_TILE_WG) {
            const u32 b = x >> 3, v = s_fill[b];
            for (u32 f = x & 7u; f < (v & 15u); f += 8) s_key[(v >> 4) + f] = key_filler<K>(b, bshift);
        }
        // ---- stage
        {
            // (in place: the rank becomes the staging index; the bucket is taken from the window again — keeping the 16 slots
            // beside the 16 ranks is what pushed the kernel over its registers.  A start past the read's end: trash slot)
            static_for<KT>([&](auto J) {
                constexpr u32 j = J;
                const u32 bkt = bbits ? w.template top_hi<j>() >> (32 - bbits) : 0u;
                idx[j] += s_base[4 * bkt + sub];
            });
            const u32 trash = CAP + wv * 8 + (ln & 7u);
            static_for<KT>([&](auto J) {
                constexpr u32 j = J;
                s_key[j < nv_now ? idx[j] : trash] = w.template key<j>(k);
            });
        }
        if (t < nb) {
            const bool fits = (u64)run + padc <= end - beg;
            s_comb[t] = fits ? beg + run - soff : ~0ull;        // staging index i goes to keys[s_comb[bucket] + i]
            if (!fits) atomicOr(flags, 2u);
        }
        __syncthreads();
        // (one 16-byte slot per wave, shared by its lanes past the end: 128 KB in all, resident in L2 — a slot per lane was an
        // 8 MB region that kept being written back: 0.2 GB of the kernel's 1.9 GB of HBM writes)
        const u64 my_scratch = scratch + (u64)((blockIdx.x & 1023u) * (GASM_TILE_WG / 64) + (t >> 6)) * KPU;
        // ---- stream out (as k_bucket_scatter; a run without room goes to the scratch line)
        static_assert(NFL % 3 == 0, "flush passes come in threes");
#pragma unroll
        for (u32 u0 = 0; u0 < NFL; u0 += 3) {
            u64x2 wd[3];
            u64 cb[3];
#pragma unroll
            for (u32 u = 0; u < 3; ++u) wd[u] = *reinterpret_cast<const u64x2*>(s_key + (t + GASM_TILE_WG * (u0 + u)) * KPU);
#pragma unroll
            for (u32 u = 0; u < 3; ++u) {
                u32 bkt = 0;
                if (bbits) {
                    if constexpr (KPU == 2) bkt = (u32)(wd[u].x >> bshift) & (nb - 1);
                    else bkt = kfield(K128{wd[u].x, wd[u].y}, bshift) & (nb - 1);
                }
                cb[u] = s_comb[bkt];
            }
#pragma unroll
            for (u32 u = 0; u < 3; ++u) {
                const u32 i = (t + GASM_TILE_WG * (u0 + u)) * KPU;
                GASM_STREAM_STORE(wd[u], reinterpret_cast<u64x2*>(keys + ((i < total && cb[u] != ~0ull) ? cb[u] + i : my_scratch)));
            }
        }
        if (++tile >= tile_end) break;
        // (no barrier here: the next tile's first writes to the staging area, the bases and the run offsets all lie behind its
        // first barrier, which a thread reaches only after its own reads of this flush; its counting touches the counters only)
        ti = tin;
        rl.template wait_all_but<NFL>();
    }
}
template __global__ void k_bucket_partition<u64>(ReadSet, const uint4*, int, int, u32, u32, u32, const u64*, u32*, u64*, u64, u32*);
template __global__ void k_bucket_partition<K128>(ReadSet, const uint4*, int, int, u32, u32, u32, const u64*, u32*, K128*, u64, u32*);

// ================================================================================================================
// De-duplicate one bucket: stream its keys through an LDS table (count per distinct key), then order the distinct keys
// and write them and their multiplicities back over the start of the bucket's own range.
//
// The table is made of small sets of slots: two 64-bit keys (one 16-byte LDS read per probe) or four 128-bit keys.  A
// key lives in the first set with room, counted from its home set; sets fill left to right.  One probe = the whole set,
// so most keys are found in the first probe; the rest — new keys, and keys whose home set had overflowed — are noted
// and worked off in one loop per iteration in which every lane takes its own next missed key.  (With 64 lanes some lane
// misses for almost every key index; handling misses in place ran the slow probe loop for the whole wave each time.)
// The kernel is LDS-bound — the random 16-byte reads conflict on banks — so a probe reads as few bytes as it can.
//   64-bit keys : slot claimed by a 64-bit CAS on the key itself (EMPTY -> key);
//   128-bit keys: no 128-bit CAS in LDS — the slot's count word is the lock: CAS 0 -> LOCKED, write the key, then
//                 count = 1.  Readers compare keys only in slots whose count says "ready"; a LOCKED slot means "try
//                 again" (never a spin inside a divergent branch: the caller's round loop simply comes back).
// Ordering: a counting sort on the key bits below the bucket prefix (TBL/4 bins; close to uniform there) puts every key
// into its bin's range, bins of more than one key are finished by a per-bin insertion sort; if any bin is long (skewed
// keys) the workgroup falls back to a bitonic sort.  The bin offsets are kept as the fine directory of the graph
// kernels.  A bucket with more than 11/16*TBL distinct keys raises *overflow (the host re-partitions).
// LDS and launch bounds: the workgroup's LDS is the table and nothing else (the ordering phase's bins live in the table's
// tail, and with 128-bit keys the workgroup's counters live in the table's last set), so 2048 slots of 128-bit keys are
// exactly a quarter of the CU's 160 KB: four workgroups per CU instead of three took the kernel from 2.46 to 2.04 ms on
// cfg4 — it lives on LDS round trips in flight, i.e. on waves.  64-bit keys: 24 KB, six workgroups would fit, but 80
// registers per lane spill in the streaming loop (0.61 ms against 0.45): five.  4096 slots: three.
// ================================================================================================================
#define GASM_SLOT_LOCKED 0xFFFFFFFFu

// 16-byte LDS read that the compiler may not reuse from an earlier read (other lanes change the table meanwhile)
__device__ __forceinline__ u64x2 lds_load128(const void* p) {
    __asm__ volatile("" ::: "memory");
    return *reinterpret_cast<const u64x2*>(p);
}
__device__ __forceinline__ uint4 lds_load128u(const void* p) {
    __asm__ volatile("" ::: "memory");
    return *reinterpret_cast<const uint4*>(p);
}

// One probe of one key on set `set`: true when the key is counted, false to probe again (same set after a lost race
// or a locked slot, next set when this one is full of other keys).
template <int TBL>
__device__ __forceinline__ bool dedup_step(u64* t_key, u32* t_cnt, u32* n_distinct, u64 key, u32& set) {
    constexpr u32 NSETS = TBL / 2;                            // 64-bit keys: sets of two = one 16-byte LDS read per probe
    const u64x2 c = lds_load128(&t_key[2 * set]);
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
    atomicAdd(&t_cnt[2 * set + slot], 1u);
    return true;
}
template <int TBL>
__device__ __forceinline__ bool dedup_step(K128* t_key, u32* t_cnt, u32* n_distinct, const K128& key, u32& set) {
    constexpr u32 NSETS = TBL / 2;
    __asm__ volatile("" ::: "memory");
    const uint2 c = *reinterpret_cast<const uint2*>(&t_cnt[2 * set]);   // counts first: a ready count guarantees a complete key
    const u32 cc[2] = {c.x, c.y};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (cc[q] == GASM_SLOT_LOCKED) return false;          // being written: may be this very key
        if (cc[q] == 0) {
            if (atomicCAS(&t_cnt[2 * set + q], 0u, GASM_SLOT_LOCKED) != 0u) return false;
            t_key[2 * set + q] = key;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __hip_atomic_store(&t_cnt[2 * set + q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            atomicAdd(n_distinct, 1u);
            return true;
        }
        const u64x2 kq = lds_load128(&t_key[2 * set + q]);
        if (kq.x == key.hi && kq.y == key.lo) { atomicAdd(&t_cnt[2 * set + q], 1u); return true; }
    }
    set = set + 1 >= NSETS - 1 ? 0u : set + 1;                // (the last set is not part of the table: k_bucket_dedup's counters)
    return false;
}

// 16-byte loads per thread and iteration of the streaming loop (= keys handled together: 2 or 1 per load) and workgroups
// per CU, measured on MI355X: 64-bit keys 3 loads 0.463 ms (4: 0.475, 2: 0.489; five or six workgroups per CU: the same),
// 128-bit keys (cfg4) 6 loads 1.94 ms (5: 2.01, 4: 2.04, 3: 2.26) — the wide table costs three LDS reads per key, and more
// keys per wave in flight is what hides them.
#ifndef GASM_DEDUP_NLD
#define GASM_DEDUP_NLD 3
#endif
#ifndef GASM_DEDUP_NLDW
#define GASM_DEDUP_NLDW 6
#endif
#ifndef GASM_DEDUP_WGS
#define GASM_DEDUP_WGS 6
#endif
// "Last workgroup done" without a release fence.  An agent-scope release (__threadfence) on this part writes back EVERY dirty
// line of the XCD's L2 (buffer_wbl2) — with the de-duplication's 6 400 workgroups and an L2 full of freshly written keys that
// took the kernel from 0.38 to 0.63 ms (measured; round 3).  What the last workgroup needs from the others is one word each,
// so that word is published by an agent-scope atomic store (written through, sc1), the thread waits for the store to complete
// (s_waitcnt: the recipe of the memory model's release minus the write-back that only plain stores need) and then counts
// itself in with a relaxed atomic; the reader uses agent-scope atomic loads.  Returns true in the workgroup that came last.
__device__ __forceinline__ void publish_u32(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void publish_u64(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool count_in_last(u32* done) {       // (by the thread that published)
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
}

// The last workgroup of a de-duplication launch to get here turns the buckets' distinct counts into `dstart` — what
// k_scan_excl<u32> did in a launch of its own (one workgroup, 8 us of work that waited 15-65 us for a slot on a chip shared
// with the other steps' kernels).  `done`: a zeroed flag word of the build.  s_word: one LDS word; thread 0 has published
// bucket_d[blockIdx.x] (publish_u32) before the call.
__device__ __forceinline__ void dedup_last_scan(const u32* __restrict__ bucket_d, u32* __restrict__ dstart, u32 n, u32* __restrict__ done, u32* s_word,
                                                u32* s_waves /* GASM_WG / 64 words */) {
    if (!dstart) return;
    __syncthreads();                                                // (the LDS words are free from here on)
    if (threadIdx.x == 0) *s_word = count_in_last(done) ? 1u : 0u;
    __syncthreads();
    if (!*s_word) return;
    const u32 per = (n + GASM_WG - 1) / GASM_WG, a = min(n, threadIdx.x * per), b = min(n, a + per);
    u32 sum = 0;
    for (u32 i = a; i < b; ++i) sum += __hip_atomic_load(&bucket_d[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u32 inc = wave_incl_scan(sum);
    __syncthreads();                                                // (s_word has been read by everybody)
    if ((threadIdx.x & 63) == 63) s_waves[threadIdx.x >> 6] = inc;
    __syncthreads();
    u32 ex = inc - sum, tot = 0;
    for (u32 w = 0; w < GASM_WG / 64; ++w) { const u32 t = s_waves[w]; if (w < (threadIdx.x >> 6)) ex += t; tot += t; }
    for (u32 i = a; i < b; ++i) { dstart[i] = ex; ex += __hip_atomic_load(&bucket_d[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    if (threadIdx.x == 0) dstart[n] = tot;
}

template <class K, int TBL>
__global__ void __launch_bounds__(GASM_WG, TBL == 4096 ? 3 : sizeof(K) == 8 ? GASM_DEDUP_WGS : 4)
k_bucket_dedup(K* __restrict__ keys, u32* __restrict__ mult, const u64* __restrict__ bstart, const u32* __restrict__ blen, u32* __restrict__ bucket_d,
               u32* __restrict__ overflow, u16* __restrict__ fdir, int low_bits, int dbg, unsigned long long* __restrict__ stamps,
               u32* __restrict__ dstart) {
    constexpr int LIMIT = TBL / 16 * 11;
    constexpr int BINS = TBL / 4;
    constexpr int LOG_TBL = TBL == 4096 ? 12 : 11;
    constexpr bool WIDE = sizeof(K) == 16;
    constexpr int LOG_SETS = LOG_TBL - 1;                     // sets of two slots
    constexpr u32 NSETS = 1u << LOG_SETS;
    static_assert(TBL == 4096 || TBL == 2048, "table size");
    __shared__ __align__(32) K t_key[TBL];
    __shared__ __align__(16) u32 t_cnt[TBL];
    // 64-bit keys: eight words beside the table.  128-bit keys: the table alone is 40 KB, a quarter of the CU's LDS, and four
    // workgroups fit only if not a word is added — its last set is kept out of the hashing, the two count words of that set
    // hold [4] and [5] while the bucket streams in, and the ordering phase (which needs eight words) uses the set's key slots
    u32 *s_tmp, *w_distinct, *w_overflow;                  // [4] distinct keys so far, [5] the table cannot take the bucket
    if constexpr (WIDE) {
        s_tmp = reinterpret_cast<u32*>(&t_key[TBL - 2]);
        w_distinct = &t_cnt[TBL - 2];
        w_overflow = &t_cnt[TBL - 1];
    } else {
        __shared__ u32 s_tmp_narrow[8];
        s_tmp = s_tmp_narrow;
        w_distinct = &s_tmp_narrow[4];
        w_overflow = &s_tmp_narrow[5];
    }
    // the bins of the ordering phase live in the table's tail, behind the LIMIT entries a sorted bucket can have (dedup_order):
    // the workgroup's LDS is the table (+ 32 bytes for 64-bit keys) — 24 / 40 / 48 KB: four workgroups per CU with 128-bit
    // keys, three with 4096 slots; with 64-bit keys and 2048 slots registers, not LDS, keep it at five
    static_assert((TBL - LIMIT) * sizeof(K) >= 2 * BINS * sizeof(u32), "the bins must fit behind the sorted entries");
    u32* const s_start = reinterpret_cast<u32*>(t_key + LIMIT);
    u32* const s_cur = s_start + BINS;
    // diagnostic only (stamps == nullptr in production): per-phase wave-0 tick totals, summed over workgroups
    unsigned long long tph = stamps ? wall_clock64() : 0ull;
    auto phase = [&](int i) {
        if (stamps && threadIdx.x == 0) { const unsigned long long t = wall_clock64(); atomicAdd(&stamps[i], t - tph); tph = t; }
    };
    const u32 bucket = blockIdx.x;
    const u64 beg = bstart[bucket], end = bstart[bucket + 1];
    u64 n = end - beg;
    if (blen) {
        // single-pass partition (k_bucket_partition): the bucket's region is [beg, end), its keys the first blen[bucket] of it.
        // A bucket that outgrew its region lost runs (bit 1 of *overflow is up, the build will be repeated): empty and searchable
        const u64 len = blen[bucket];
        if (len > n) {
            for (u32 i = threadIdx.x; i <= (u32)BINS; i += GASM_WG) fdir[(u64)bucket * (BINS + 1) + i] = 0;
            if (threadIdx.x == 0) publish_u32(&bucket_d[bucket], 0u);
            dedup_last_scan(bucket_d, dstart, gridDim.x, overflow + 8, s_tmp + 7, s_tmp);
            return;
        }
        n = len;
    }
    const u32 warm = (u32)(dbg >> 2) & 3u;        // iterations taken key by key (64-bit keys)
    for (u32 i = threadIdx.x; i < TBL; i += GASM_WG) { t_key[i] = key_empty<K>(); t_cnt[i] = 0; }
    if (!WIDE && threadIdx.x == 0) { *w_distinct = 0; *w_overflow = 0; s_tmp[6] = 0; }   // ([6] longest bin: dedup_order zeroes it when the bins live in the table)
    __syncthreads();
    phase(0);
    // stream: NLD 16-byte loads per thread in flight (the loop is latency-bound otherwise), each fully coalesced across
    // the wave.  Bucket ranges start and end on 128-byte lines (filler keys = EMPTY are skipped).
    constexpr int NLD = WIDE ? GASM_DEDUP_NLDW : TBL == 2048 ? GASM_DEDUP_NLD : 4;   // 16-byte loads per thread and iteration
    constexpr int KPL = NLD * 16 / sizeof(K);    // keys per thread and iteration
    const u64 nch = n * sizeof(K) / 16;          // 16-byte chunks in the bucket
    const uint4* src = reinterpret_cast<const uint4*>(keys + beg);
    if (dbg & 16) src = reinterpret_cast<const uint4*>(keys + bstart[bucket & 7u]);      // ablation: every workgroup streams one of eight buckets (L2-resident): the table work without HBM
    // the loads of the next iteration are issued before this iteration's keys go into the table, so the table work
    // (LDS latency) and the HBM latency overlap inside every wave
    uint4 v[NLD];
    auto fetch = [&](u64 c) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) v[q] = c + (u64)q * GASM_WG < nch ? GASM_STREAM_LOAD(&src[c + (u64)q * GASM_WG]) : make_uint4(~0u, ~0u, ~0u, ~0u);
    };
    if (threadIdx.x < nch) fetch(threadIdx.x);
    for (u64 c = threadIdx.x; c < nch; c += NLD * GASM_WG) {
        if (c == (u64)threadIdx.x + NLD * GASM_WG) phase(1);   // first iteration (table fill) done
        K kx[KPL];
        if constexpr (WIDE) {
#pragma unroll
            for (int q = 0; q < NLD; ++q) { kx[q].hi = (u64)v[q].x | ((u64)v[q].y << 32); kx[q].lo = (u64)v[q].z | ((u64)v[q].w << 32); }
        } else {
#pragma unroll
            for (int q = 0; q < NLD; ++q) { kx[2 * q] = (u64)v[q].x | ((u64)v[q].y << 32); kx[2 * q + 1] = (u64)v[q].z | ((u64)v[q].w << 32); }
        }
        if (c + NLD * GASM_WG < nch) fetch(c + NLD * GASM_WG);
        if ((dbg & 3) == 1) { u64 x = 0; for (int q = 0; q < KPL; ++q) x ^= khash(kx[q]); if (x == 0x1234567) *w_overflow = 1; continue; }
        if (__hip_atomic_load(w_distinct, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > (u32)LIMIT) { *w_overflow = 1; break; }
        if constexpr (!WIDE) {
            // While the table fills (the first `warm` iterations: 2048 keys each, ~600 distinct per bucket) nearly every
            // key misses its home set, and the batched form below would send all eight keys of every lane through the slow
            // loop — 17 of a workgroup's 95 us (GASM_DBG_STAMPS).  There the keys are taken one after the other instead:
            // a key inserted by any lane at step q is a plain hit for everybody from step q + 1 on.
            if (c < (u64)warm * NLD * GASM_WG) {
#pragma unroll
                for (int q = 0; q < KPL; ++q) {
                    const u64 key = kx[q];
                    if (kis_filler(key)) continue;
                    u32 st = khash(key) >> (32 - LOG_SETS);
                    bool ok = false;
                    for (u32 probe = 0; probe < 8 * NSETS && !ok; ++probe) ok = dedup_step<TBL>(t_key, t_cnt, w_distinct, key, st);
                    if (!ok) *w_overflow = 1;
                }
                continue;
            }
            // two batches of four keys: the four home sets are read together (8 x ds_read_b128 in flight) and hits are
            // counted.  Keys that miss their home set (new keys, and the ~1.5 % whose home set has overflowed) are only
            // noted: with 64 lanes some lane misses for almost every key index, and handling misses in place ran the
            // slow probe loop — a few dependent LDS round trips — eight times per iteration for the whole wave.  They
            // are worked off afterwards in one loop in which every lane takes its own next missed key.
            u32 missed = 0;
            {
                u32 set[KPL];
                u64x2 c[KPL];
#pragma unroll
                for (int q = 0; q < KPL; ++q) {
                    set[q] = khash(kx[q]) >> (32 - LOG_SETS);
                    c[q] = lds_load128(&t_key[2 * set[q]]);
                }
#pragma unroll
                for (int q = 0; q < KPL; ++q) {
                    const u64 key = kx[q];
                    if (kis_filler(key)) continue;            // (before the slot test: the all-ones filler equals a free slot)
                    const int slot = c[q].x == key ? 0 : c[q].y == key ? 1 : -1;
                    if (slot >= 0) atomicAdd(&t_cnt[2 * set[q] + slot], 1u);
                    else missed |= 1u << q;
                }
            }
            while (missed) {
                const u32 q = (u32)__builtin_ctz(missed);
                missed &= missed - 1;
                u64 key = kx[0];
#pragma unroll
                for (u32 e = 1; e < (u32)KPL; ++e) if (e == q) key = kx[e];
                u32 st = khash(key) >> (32 - LOG_SETS);
                bool ok = false;
                for (u32 probe = 0; probe < 8 * NSETS && !ok; ++probe) ok = dedup_step<TBL>(t_key, t_cnt, w_distinct, key, st);
                if (!ok) *w_overflow = 1;
            }
        } else {
            // 128-bit keys: the four home sets' count words and both slots are read together; the misses are worked off
            // afterwards, every lane on its own next missed key (as above)
            u32 set[KPL];
            uint2 cn[KPL];
            u64x2 k0[KPL], k1[KPL];
#pragma unroll
            for (int q = 0; q < KPL; ++q) {
                set[q] = min(khash(kx[q]) >> (32 - LOG_SETS), NSETS - 2);      // (the last set holds the workgroup's counters)
                __asm__ volatile("" ::: "memory");
                cn[q] = *reinterpret_cast<const uint2*>(&t_cnt[2 * set[q]]);
                k0[q] = lds_load128(&t_key[2 * set[q]]);
                k1[q] = lds_load128(&t_key[2 * set[q] + 1]);
            }
            u32 missed = 0;
#pragma unroll
            for (int q = 0; q < KPL; ++q) {
                const K128 key = kx[q];
                if (kis_filler(key)) continue;
                const bool r0 = cn[q].x != 0 && cn[q].x != GASM_SLOT_LOCKED, r1 = cn[q].y != 0 && cn[q].y != GASM_SLOT_LOCKED;
                if (r0 && k0[q].x == key.hi && k0[q].y == key.lo) atomicAdd(&t_cnt[2 * set[q]], 1u);
                else if (r0 && r1 && k1[q].x == key.hi && k1[q].y == key.lo) atomicAdd(&t_cnt[2 * set[q] + 1], 1u);
                else missed |= 1u << q;
            }
            while (missed) {
                const u32 q = (u32)__builtin_ctz(missed);
                missed &= missed - 1;
                K128 key = kx[0];
#pragma unroll
                for (u32 e = 1; e < (u32)KPL; ++e) if (e == q) key = kx[e];
                u32 st = min(khash(key) >> (32 - LOG_SETS), NSETS - 2);
                bool ok = false;
                for (u32 probe = 0; probe < 8 * NSETS && !ok; ++probe) ok = dedup_step<TBL>(t_key, t_cnt, w_distinct, key, st);
                if (!ok) *w_overflow = 1;
            }
        }
    }
    phase(2);
    __syncthreads();
    phase(3);
    if (*w_overflow || *w_distinct > (u32)LIMIT) {
        // The host repeats the build with a larger configuration — but it does not wait for this report before the graph
        // kernels of THIS attempt run (pipeline_build_finish), so the bucket must be left empty AND searchable: an all-zero
        // fine directory (graph_lower_bound would otherwise bisect between whatever the allocation held)
        for (u32 i = threadIdx.x; i <= (u32)BINS; i += GASM_WG) fdir[(u64)bucket * (BINS + 1) + i] = 0;
        if (threadIdx.x == 0) { atomicOr(overflow, 1u); publish_u32(&bucket_d[bucket], 0u); }
        dedup_last_scan(bucket_d, dstart, gridDim.x, overflow + 8, s_tmp + 7, s_tmp);
        return;
    }
    const u32 d = *w_distinct;
    if ((dbg & 3) == 1 || (dbg & 3) == 2) { if (threadIdx.x == 0) publish_u32(&bucket_d[bucket], d); dedup_last_scan(bucket_d, dstart, gridDim.x, overflow + 8, s_tmp + 7, s_tmp); return; }
    dedup_order<K, TBL, true>(t_key, t_cnt, s_start, s_cur, s_tmp, fdir, bucket, low_bits, d);
    phase(4);
    for (u32 i = threadIdx.x; i < d; i += GASM_WG) { keys[beg + i] = t_key[i]; mult[beg + i] = t_cnt[i]; }
    if (threadIdx.x == 0) publish_u32(&bucket_d[bucket], d);
    phase(5);
    if (stamps && threadIdx.x == 0) stamps[8 + 3 * (u64)blockIdx.x + 1] = wall_clock64();
    dedup_last_scan(bucket_d, dstart, gridDim.x, overflow + 8, s_tmp + 7, s_tmp);
}
template __global__ void k_bucket_dedup<u64, 4096>(u64*, u32*, const u64*, const u32*, u32*, u32*, u16*, int, int, unsigned long long*, u32*);
template __global__ void k_bucket_dedup<u64, 2048>(u64*, u32*, const u64*, const u32*, u32*, u32*, u16*, int, int, unsigned long long*, u32*);
template __global__ void k_bucket_dedup<K128, 2048>(K128*, u32*, const u64*, const u32*, u32*, u32*, u16*, int, int, unsigned long long*, u32*);

// ================================================================================================================
// De-duplication of buckets that no table can hold (the last rung of pipeline_build_finish's ladder: ten bucket bits and
// still more than LIMIT distinct keys with one prefix — skewed base composition, e.g. two-letter sequences).  Buckets are
// key ranges, so a bucket splits into 2^r ordered sub-ranges by the r bits below its prefix: the workgroup makes 2^r
// passes over the bucket, each pass puts the keys of one sub-range through the table, orders them and appends them to the
// output — which is therefore sorted as a whole.  r grows until every sub-range fits.  The passes re-read the bucket, so
// the output goes to a second array (keys_out, same layout), and the fine directory is written range by range.
// 4096-slot tables for both key widths; a bucket may hold up to GASM_BUCKET_MAX distinct keys (16-bit directory).
// Speed is not the point here.
// ================================================================================================================
template <class K>
__global__ void __launch_bounds__(GASM_WG) k_bucket_dedup_multi(const K* __restrict__ keys, K* __restrict__ keys_out, u32* __restrict__ mult,
                                                                const u64* __restrict__ bstart, u32* __restrict__ bucket_d,
                                                                u32* __restrict__ overflow, u16* __restrict__ fdir, int low_bits) {
    constexpr int TBL = 4096, LIMIT = TBL / 16 * 11, BINS = TBL / 4, LOG_SETS = 11, LOG_BINS = 10;
    constexpr u32 NSETS = 1u << LOG_SETS;
    __shared__ __align__(32) K t_key[TBL];
    __shared__ __align__(16) u32 t_cnt[TBL];
    __shared__ u32 s_start[BINS];
    __shared__ u32 s_cur[BINS];
    __shared__ u32 s_tmp[8];
    const u32 bucket = blockIdx.x;
    const u64 beg = bstart[bucket], n = bstart[bucket + 1] - beg;
    const int bin_bits = low_bits < LOG_BINS ? low_bits : LOG_BINS;      // the bits of a key that select its bin (dedup_order)
    const int r_max = low_bits < 16 ? low_bits : 16;            // at most 65536 passes
    u16* const fd = fdir + (u64)bucket * (BINS + 1);
    auto give_up = [&]() {                                      // (an empty, searchable bucket: see k_bucket_dedup)
        for (u32 i = threadIdx.x; i <= (u32)BINS; i += GASM_WG) fd[i] = 0;
        if (threadIdx.x == 0) { atomicOr(overflow, 1u); bucket_d[bucket] = 0; }
    };
    for (int r = 0;; ++r) {
        if (r > r_max) { give_up(); return; }
        const u32 npass = 1u << r;
        u32 total = 0;
        bool fits = true;
        for (u32 p = 0; p < npass; ++p) {
            for (u32 i = threadIdx.x; i < TBL; i += GASM_WG) { t_key[i] = key_empty<K>(); t_cnt[i] = 0; }
            for (u32 i = threadIdx.x; i < BINS; i += GASM_WG) s_start[i] = 0;
            if (threadIdx.x == 0) { s_tmp[4] = 0; s_tmp[5] = 0; s_tmp[6] = 0; }
            __syncthreads();
            for (u64 i = threadIdx.x; i < n; i += GASM_WG) {
                const K key = keys[beg + i];
                if (kis_filler(key)) continue;
                if (r && (kfield(key, low_bits - r) & (npass - 1)) != p) continue;
                if (__hip_atomic_load(&s_tmp[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > (u32)LIMIT) break;
                u32 st = khash(key) >> (32 - LOG_SETS);
                bool ok = false;
                for (u32 probe = 0; probe < 8 * NSETS && !ok; ++probe) ok = dedup_step<TBL>(t_key, t_cnt, &s_tmp[4], key, st);
                if (!ok) s_tmp[5] = 1;
            }
            __syncthreads();
            const u32 d = s_tmp[4];
            if (s_tmp[5] || d > (u32)LIMIT) { fits = false; break; }            // (the same for every thread)
            if (total + d > GASM_BUCKET_MAX) { give_up(); return; }
            // the bins of this sub-range; with more sub-range bits than bin bits several passes share a bin, and the first
            // of them writes its directory entry
            u32 lo, hi;
            if (r <= bin_bits) { lo = p << (bin_bits - r); hi = (p + 1) << (bin_bits - r); }
            else { lo = p >> (r - bin_bits); hi = (p & ((1u << (r - bin_bits)) - 1u)) ? lo : lo + 1; }
            dedup_order<K, TBL, false, true>(t_key, t_cnt, s_start, s_cur, s_tmp, fdir, bucket, low_bits, d, total, lo, hi);
            for (u32 i = threadIdx.x; i < d; i += GASM_WG) { keys_out[beg + total + i] = t_key[i]; mult[beg + total + i] = t_cnt[i]; }
            total += d;
            __syncthreads();
        }
        if (!fits) { __syncthreads(); continue; }
        // bins no key can fall into (fewer than nine bits below the prefix), and the end of the directory
        for (u32 i = (1u << bin_bits) + threadIdx.x; i <= (u32)BINS; i += GASM_WG) fd[i] = (u16)total;
        if (threadIdx.x == 0) bucket_d[bucket] = total;
        return;
    }
}
template __global__ void k_bucket_dedup_multi<u64>(const u64*, u64*, u32*, const u64*, u32*, u32*, u16*, int);
template __global__ void k_bucket_dedup_multi<K128>(const K128*, K128*, u32*, const u64*, u32*, u32*, u16*, int);

// Gather the per-bucket distinct runs into the dense per-segment arrays.
template <class K>
__global__ void __launch_bounds__(GASM_WG) k_bucket_gather(const K* __restrict__ keys, const u32* __restrict__ mult,
                                                           const u64* __restrict__ bstart, const u32* __restrict__ dstart,
                                                           K* __restrict__ dk_key, u32* __restrict__ dk_cnt, u32* __restrict__ claim,
                                                           u8* __restrict__ eflag, u32* __restrict__ flags) {
    const u32 bucket = blockIdx.x;
    // the graph kernels' flags — [1] ranking gave up, [16..] "still active" words of the k_link_jump launches — start at
    // zero; [0], the de-duplication's overflow flag, stays
    if (bucket == 0 && threadIdx.x >= 1 && threadIdx.x < 64) flags[threadIdx.x] = 0;
    const u64 src = bstart[bucket];
    const u32 dst = dstart[bucket], d = dstart[bucket + 1] - dst;
    for (u32 i = threadIdx.x; i < d; i += GASM_WG) {
        dk_key[dst + i] = keys[src + i];
        dk_cnt[dst + i] = mult[src + i];
        claim[dst + i] = GASM_NONE32;     // initial state of the degree kernels (k_edge_target)
        eflag[dst + i] = 0;
    }
}
template __global__ void k_bucket_gather<u64>(const u64*, const u32*, const u64*, const u32*, u64*, u32*, u32*, u8*, u32*);
template __global__ void k_bucket_gather<K128>(const K128*, const u32*, const u64*, const u32*, K128*, u32*, u32*, u8*, u32*);

// ================================================================================================================
// Graph over the sorted distinct k-mers (= distinct edges) of each segment.  Edge i: key = x·M·y, source node
// u = key>>2 (x·M), target node v = key & mask (M·y).  blockIdx.y = segment.
// ================================================================================================================
template <class K>
__device__ __forceinline__ bool kmer_exists(const GraphView& gv, u32 seg, const K& t) {
    u32 hi;
    const u32 j = graph_lower_bound<K>(gv, seg, t, &hi);
    return j < hi && keq(reinterpret_cast<const K*>(gv.dk_key)[j], t);
}

// Degrees without counting.  Every edge looks up the first out-edge j of its target node once (k_edge_target) and
// writes its own index into claim[j]; whichever edge wins, an edge that later finds somebody else's index there
// (k_edge_multi) knows the node has a second in-edge and sets bit 1 of eflag[j].  So for the node whose out-edges
// start at r: in-degree 0 <=> claim[r] untouched, >= 2 <=> bit 1 of eflag[r], else 1 — plain stores only, one lookup
// per edge instead of the four "does x.u exist" lookups of a direct in-degree count.
//   eflag bit0: the edge's source node is a branching node (in != 1 or out != 1); it has out-edges by construction.
//   eflag bit1: (on the first out-edge of a node) the node has two or more in-edges.
// k_bucket_gather initialises claim = none and eflag = 0.
template <class K>
__global__ void __launch_bounds__(GASM_WG) k_edge_target(GraphView gv, u32 n_segments, u32 chunks, u32* __restrict__ tgt,
                                                         u32* __restrict__ claim) {
    const K* dk = reinterpret_cast<const K*>(gv.dk_key);
    for_seg_edges(gv.dstart, 1u << gv.bbits, n_segments, chunks, [&](u32 seg, u32, u32, u32 i) {
        const K v = klowbits(dk[i], 2 * (gv.k - 1));
        // the run of v may continue into the next bucket only if bbits > 2(k-1), which the host never chooses
        u32 bhi;
        const u32 j = graph_lower_bound<K>(gv, seg, kshl(v, 2), &bhi);      // smallest k-mer with prefix v
        const bool valid = j < bhi && keq(kshr(dk[j], 2), v);
        tgt[i] = valid ? j : GASM_NONE32;
        if (valid) claim[j] = i;
    });
}
template __global__ void k_edge_target<u64>(GraphView, u32, u32, u32*, u32*);
template __global__ void k_edge_target<K128>(GraphView, u32, u32, u32*, u32*);

__global__ void __launch_bounds__(GASM_WG) k_edge_multi(GraphView gv, u32 n_segments, u32 chunks, const u32* __restrict__ tgt,
                                                        const u32* __restrict__ claim, u8* __restrict__ eflag) {
    for_seg_edge_groups(gv.dstart, 1u << gv.bbits, n_segments, chunks, [&](u32, u32, u32, const u32 (&i)[GASM_EDGE_ILP], const bool (&ok)[GASM_EDGE_ILP]) {
        u32 j[GASM_EDGE_ILP], c[GASM_EDGE_ILP];
#pragma unroll
        for (int q = 0; q < GASM_EDGE_ILP; ++q) j[q] = ok[q] ? tgt[i[q]] : GASM_NONE32;
#pragma unroll
        for (int q = 0; q < GASM_EDGE_ILP; ++q) c[q] = j[q] != GASM_NONE32 ? claim[j[q]] : 0u;
#pragma unroll
        for (int q = 0; q < GASM_EDGE_ILP; ++q)
            if (j[q] != GASM_NONE32 && c[q] != i[q]) eflag[j[q]] = 2;
    });
}

template <class K>
__global__ void __launch_bounds__(GASM_WG) k_node_flags(GraphView gv, u32 n_segments, u32 chunks, const u32* __restrict__ claim,
                                                        u8* __restrict__ eflag, u64* __restrict__ link, u32* __restrict__ clen) {
    const K* dk = reinterpret_cast<const K*>(gv.dk_key);
    for_seg_edges(gv.dstart, 1u << gv.bbits, n_segments, chunks, [&](u32, u32 lo, u32 hi, u32 i) {
        const K u = kshr(dk[i], 2);
        // out-degree of u: the run of keys sharing key>>2 is contiguous in the sorted list; r = its first edge
        u32 r = i;
        while (r > lo && keq(kshr(dk[r - 1], 2), u)) --r;
        u32 e = i + 1;
        while (e < hi && keq(kshr(dk[e], 2), u)) ++e;
        const u32 outd = e - r;
        const bool in_one = claim[r] != GASM_NONE32 && !(eflag[r] & 2);
        eflag[i] = (eflag[i] & 2) | ((!in_one || outd != 1) ? 1 : 0);
        // (no initial link: k_edge_next writes one for every edge — a head's by the head itself, any other edge's by its one
        // predecessor: its source node has exactly one in-edge, whose target's first and only out-edge it is)
        clen[i] = 0;
    });
}
template __global__ void k_node_flags<u64>(GraphView, u32, u32, const u32*, u8*, u64*, u32*);
template __global__ void k_node_flags<K128>(GraphView, u32, u32, const u32*, u8*, u64*, u32*);

// Successor edge of every edge (GASM_NONE32 when the walk stops at its target), and the initial ancestor links:
// link = ancestor << 32 | done << 31 | distance, done = "the ancestor is the head of the chain".  Heads are their own
// ancestor at distance 0.  The done bit travels with the link, so pointer doubling needs one gather per round.
// (nxt may be the array claim lived in: claim is dead by now.)
__global__ void __launch_bounds__(GASM_WG) k_edge_next(GraphView gv, u32 n_segments, u32 chunks, const u32* __restrict__ tgt,
                                                       const u8* __restrict__ eflag, u32* __restrict__ nxt, u64* __restrict__ link) {
    for_seg_edge_groups(gv.dstart, 1u << gv.bbits, n_segments, chunks, [&](u32, u32, u32, const u32 (&i)[GASM_EDGE_ILP], const bool (&ok)[GASM_EDGE_ILP]) {
        u32 j[GASM_EDGE_ILP], efi[GASM_EDGE_ILP], efj[GASM_EDGE_ILP];
#pragma unroll
        for (int q = 0; q < GASM_EDGE_ILP; ++q) { j[q] = ok[q] ? tgt[i[q]] : GASM_NONE32; efi[q] = ok[q] ? (u32)eflag[i[q]] : 0u; }
#pragma unroll
        for (int q = 0; q < GASM_EDGE_ILP; ++q) efj[q] = j[q] != GASM_NONE32 ? (u32)eflag[j[q]] : 1u;
#pragma unroll
        for (int q = 0; q < GASM_EDGE_ILP; ++q) {
            if (!ok[q]) continue;
            const u32 n = (j[q] != GASM_NONE32 && !(efj[q] & 1)) ? j[q] : GASM_NONE32;   // the target has out-edges and is not branching
            nxt[i[q]] = n;
            const u64 me_head = (efi[q] & 1) ? GASM_LINK_DONE : 0ull;
            if (n != GASM_NONE32) link[n] = ((u64)i[q] << 32) | me_head | 1ull;
            if (me_head) link[i[q]] = ((u64)i[q] << 32) | GASM_LINK_DONE;
        }
    });
}

// Pointer doubling towards the head of the chain, `jumps` steps per launch.  In place and asynchronous: a link is
// always a consistent (ancestor, done, distance) triple because it is read and written as one 64-bit word, and a stale
// triple is still a true statement about the chain — so neither the steps inside a launch nor the workgroups need to
// be in step; what is guaranteed per launch is that every link's span grows by the factor jumps + 1 (each step adds at
// least the span its ancestor had before the launch).  Whole-GPU launches beat one workgroup per segment: a CU resolves about one scattered address per clock,
// and a segment's ~15 rounds of gathers through a single CU took 0.28 ms against 5 launches of ~10 us here.
// `active` (one word per launch, zeroed by the host): set when some link is still short of its head; a launch returns
// at once when the previous one left it clear.  Members of isolated cycles never finish: the number of launches bounds them.
#define GASM_JUMP_ILP 4      // links per thread, advanced together: the steps are dependent gathers, so the kernel lives on loads in flight
// nxt / clen (optional): the launch also does k_chain_len's work for the links it sees final — a chain's last edge publishes
// the chain's length at its head; the first launch for every edge, later ones for the edges they finish.
__global__ void __launch_bounds__(GASM_WG) k_link_jump(GraphView gv, u32 n_segments, u32 chunks, u64* __restrict__ link,
                                                       const u32* __restrict__ prev_active, u32* __restrict__ active, int jumps,
                                                       const u32* __restrict__ nxt, u32* __restrict__ clen) {
    if (prev_active && *prev_active == 0) return;
    u32 seg, chunk;
    if (!seg_chunk(n_segments, chunks, &seg, &chunk)) return;       // a segment's links stay in one XCD's L2
    const u32 nb = 1u << gv.bbits;
    const u32 lo = gv.dstart[seg * nb], hi = gv.dstart[(seg + 1) * nb];
    // (`chunks` comes from an estimate of the largest segment: further rounds cover a larger one)
    for (u32 base = chunk * (GASM_WG * GASM_JUMP_ILP); base < hi - lo; base += chunks * (GASM_WG * GASM_JUMP_ILP)) {
        const u32 i0 = lo + base + threadIdx.x;
        u64 l[GASM_JUMP_ILP];
        bool act[GASM_JUMP_ILP];
#pragma unroll
        for (int q = 0; q < GASM_JUMP_ILP; ++q) {
            const u32 i = i0 + q * GASM_WG;
            l[q] = i < hi ? link[i] : ~0ull;
            act[q] = (u32)(l[q] >> 32) != GASM_NONE32 && !(l[q] & GASM_LINK_DONE);
        }
        const bool mine = act[0] || act[1] || act[2] || act[3];
        u32 tail = 0;                                 // bit q: edge q is the last of its chain and this launch owes it the length
        if (nxt) {
#pragma unroll
            for (int q = 0; q < GASM_JUMP_ILP; ++q) {
                const u32 i = i0 + q * GASM_WG;
                if (i < hi && (act[q] || !prev_active) && nxt[i] == GASM_NONE32) tail |= 1u << q;
            }
        }
        for (int j = 0; j < jumps && (act[0] || act[1] || act[2] || act[3]); ++j) {
            u64 la[GASM_JUMP_ILP];
#pragma unroll
            for (int q = 0; q < GASM_JUMP_ILP; ++q)      // may be stale: fine
                la[q] = act[q] ? __hip_atomic_load(&link[(u32)(l[q] >> 32)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0ull;
#pragma unroll
            for (int q = 0; q < GASM_JUMP_ILP; ++q) {
                if (!act[q]) continue;
                if ((u32)(la[q] >> 32) == GASM_NONE32) { l[q] = ~0ull; act[q] = false; continue; }   // behind a dropped edge: dropped too
                if (l[q] & GASM_LINK_TAG) {
                    // tagged by k_rank_rulers: `distance` edges ahead of the ruler whose final link this is — same head, that much nearer
                    if (!(la[q] & GASM_LINK_DONE)) { l[q] = ~0ull; act[q] = false; continue; }        // (a ruler is final or dropped by now)
                    l[q] = (la[q] & (0xFFFFFFFF00000000ull | GASM_LINK_DONE)) | (u64)((((u32)la[q] & 0x7FFFFFFFu) - ((u32)l[q] & 0x3FFFFFFFu)) & 0x3FFFFFFFu);
                    act[q] = false;
                    continue;
                }
                l[q] = (la[q] & (0xFFFFFFFF00000000ull | GASM_LINK_DONE)) | (u64)(((u32)l[q] + (u32)la[q]) & 0x7FFFFFFFu);
                if (l[q] & GASM_LINK_DONE) act[q] = false;
            }
        }
#pragma unroll
        for (int q = 0; q < GASM_JUMP_ILP; ++q)
            if (((tail >> q) & 1u) && (u32)(l[q] >> 32) != GASM_NONE32 && (l[q] & GASM_LINK_DONE))
                clen[(u32)(l[q] >> 32)] = ((u32)l[q] & 0x7FFFFFFFu) + 1;
        if (!mine) continue;
        bool open = false;
#pragma unroll
        for (int q = 0; q < GASM_JUMP_ILP; ++q) {
            const u32 i = i0 + q * GASM_WG;
            if (i < hi) { link[i] = l[q]; open = open || ((u32)(l[q] >> 32) != GASM_NONE32 && !(l[q] & GASM_LINK_DONE)); }
        }
        if (open) *active = 1u;
    }
}

// ----------------------------------------------------------------------------------------------------------------
// List ranking through LDS for segments of up to 65534 edges.  Pointer doubling costs ~log2(chain length) scattered
// reads per edge, and scattered 8-byte reads are bound by the L2 request rate (16 per clock and XCD: ~0.25 ms for
// 3.8 M edges).  So only every second edge of a segment — the "rulers", even local index — is ranked by doubling, and
// that happens in LDS:
//   k_rank_rulers   every ruler walks back along the initial links to the nearest ruler or head (two steps on average)
//                   and writes one 32-bit entry: ancestor (local index) << 16 | distance;
//   k_rank_lds      one workgroup per segment doubles the ruler list inside LDS (up to 32767 entries = 128 KB) and
//                   writes the rulers' final links;
//   k_link_jump     finishes the odd edges: the ruler behind them is final, so a step or two each.
// An entry's ancestor is a ruler (even) until the walk or a doubling step reaches a head; heads may be odd or even, an
// even head's own entry is the self-loop (itself, 0).  So "my ancestor is a head" <=> it is odd or its entry loops.
// ----------------------------------------------------------------------------------------------------------------
#define GASM_RANK_NONE 0xFFFFFFFFu
// rulers = edges whose local index is a multiple of 1 << rshift (a launch parameter: 1 for batches of many segments —
// every CU has its own segment, LDS time per segment is what counts — 2 for a few segments, where the LDS kernel's
// rounds are the longest latency of the whole build and the longer walks of the other two kernels are spread over the chip)
__global__ void __launch_bounds__(GASM_WG) k_rank_rulers(GraphView gv, u32 n_segments, u32 chunks, u64* __restrict__ link,
                                                         u32* __restrict__ rtab, u32 rshift, u32* __restrict__ flags) {
    const u32 rmask = (1u << rshift) - 1u;
    u32 seg, chunk;
    if (!seg_chunk(n_segments, chunks, &seg, &chunk)) return;
    const u32 nb = 1u << gv.bbits;
    const u32 lo = gv.dstart[seg * nb], hi = gv.dstart[(seg + 1) * nb];
    const u32 nr = (hi - lo + rmask) >> rshift;
    for (u32 r = chunk * GASM_WG + threadIdx.x; r < nr; r += chunks * GASM_WG) {      // ruler ordinal inside the segment
        const u32 i = lo + (r << rshift);
        u32 cur = i, acc = 0, e = GASM_RANK_NONE;
        int step = 0;
        for (; step < 4096; ++step) {                     // the walk ends at the first ruler: 2^rshift steps on average
            const u64 l = link[cur];
            // an edge the walk passes (no ruler: only this walk ever reads its link) learns where it is: `acc` edges ahead of
            // ruler i.  k_link_jump then finishes it with ONE gather of the ruler's final link, instead of following
            // its predecessors one by one (runs of non-rulers are geometric: the longest of a wave's 256 is ~9 steps)
            if (cur != i) link[cur] = ((u64)i << 32) | GASM_LINK_TAG | acc;
            const u32 a = (u32)(l >> 32);
            if (a == GASM_NONE32) break;
            acc += (u32)l & 0x7FFFFFFFu;
            const u32 al = a - lo;
            if ((l & GASM_LINK_DONE) || !(al & rmask)) { e = (al << 16) | acc; break; }
            if (a == i) break;                            // back at the start: an isolated cycle without a ruler, no contig
            cur = a;
        }
        // 4096 edges in a row without a ruler: the host repeats the ranking with whole-GPU pointer doubling instead of
        // handing out a contig that silently lost edges
        if (step == 4096) flags[1] = 1u;
        rtab[(lo >> rshift) + seg + r] = e;     // segment s owns entries [(lo >> shift) + s, ...): room for the ragged ends
    }
}

// k_rank_lds.  A first version decided "is my ancestor a head" from the ancestor's entry in every round (odd index, or an
// entry that loops with distance 0, or loops with a distance: a folded cycle, or points at me ...) with a branch per case: 78
// instructions per entry and round, 51 of them exec-mask bookkeeping — and with 16 waves on four SIMDs this kernel is bound
// by instruction issue, not by its LDS operations (19 200 entries x 14 rounds: 2.5 LDS operations per clock): 0.097 ms on
// cfg2.  So the list is first rewritten in LDS such that a round needs one test and no branch (17 instructions, 0.051 ms):
//   * an entry whose ancestor is a head is TERMINAL — (0x8000 | its own ruler index) << 16, final from the start;
//   * a dropped entry is DEAD — 0x8000FFFF (the terminal flag over index 0, with a distance no terminal entry has);
//   * every other entry holds its ancestor's RULER INDEX (15 bits) and the distance to it.
// A round: gather my ancestor's entry; bit 31 set -> I stop (my ancestor is a terminal ruler t and my distance to it is
// known — or it is dead, which the end sorts out); else hop.  All of it is selects on integers; entries that have stopped
// gather and rewrite themselves (LDS operations are not what the kernel is short of).  Members of isolated cycles never
// stop and are dropped after max_rounds.  Which head t stands for, and t's distance to it, is t's original record in rtab:
// one gather per ruler at the very end.
#define GASM_RANK_DEAD 0x8000FFFFu
__global__ void __launch_bounds__(1024) k_rank_lds(GraphView gv, const u32* __restrict__ rtab, u64* __restrict__ link, int max_rounds,
                                                    u32 rshift, u32 lds_entries, u32* __restrict__ flags) {
    const u32 rmask = (1u << rshift) - 1u;
    extern __shared__ u32 s_e[];                       // lds_entries + 1 (the last one takes the writes of entries past the list)
    __shared__ u32 s_active;
    const u32 seg = blockIdx.x;
    const u32 nb = 1u << gv.bbits;
    const u32 lo = gv.dstart[seg * nb], hi = gv.dstart[(seg + 1) * nb];
    const u32 nr = (hi - lo + rmask) >> rshift;
    if (nr > lds_entries || hi - lo > 65534u) { if (threadIdx.x == 0) flags[1] = 1u; return; }
    const u32* src = rtab + (lo >> rshift) + seg;
    u32 mine[32];
    u32 live = 0;                                     // bit q: entry q of this thread is still hopping
#pragma unroll
    for (u32 q = 0; q < 32; ++q) {
        const u32 r = threadIdx.x + 1024 * q;
        u32 e = r < nr ? src[r] : GASM_RANK_NONE;
        bool hopping = false;
        if (e == GASM_RANK_NONE) e = GASM_RANK_DEAD;
        else {
            const u32 a = e >> 16;
            if ((a & rmask) || (a == (r << rshift) && !(e & 0xFFFFu))) e = (0x8000u | r) << 16;       // my ancestor is a head (or I am one)
            else if (a == (r << rshift)) e = GASM_RANK_DEAD;                                           // around a cycle, back at myself
            else { e = ((a >> rshift) << 16) | (e & 0xFFFFu); hopping = true; }
        }
        mine[q] = e;
        if (r < nr) s_e[r] = e;
        live |= (u32)hopping << q;
    }
    for (int round = 0; round < max_rounds; ++round) {
        if (threadIdx.x == 0) s_active = 0;
        __syncthreads();
        u32 any = 0;
#pragma unroll
        for (u32 q0 = 0; q0 < 32; q0 += 8) {
            if (((live >> q0) & 0xFFu) == 0) continue;
            u32 x[8];
#pragma unroll
            for (u32 u = 0; u < 8; ++u) x[u] = s_e[(mine[q0 + u] >> 16) & 0x7FFFu];
#pragma unroll
            for (u32 u = 0; u < 8; ++u) {
                const u32 q = q0 + u;
                const u32 r = threadIdx.x + 1024 * q, e = mine[q];
                const u32 lv = (live >> q) & 1u, stop = x[u] >> 31;
                const u32 hop = lv & ~stop;
                const u32 hopped = (x[u] & 0xFFFF0000u) | ((e + x[u]) & 0xFFFFu);
                const u32 ne = hop ? hopped : e;
                live &= ~((lv & stop) << q);
                any |= hop;
                mine[q] = ne;
                s_e[min(r, nr)] = ne;
            }
        }
        if (any) s_active = 1;
        __syncthreads();
        const bool go = s_active != 0;
        __syncthreads();
        if (!go) break;
    }
    // final links: a stopped ruler through its terminal ruler's record (a terminal one through its own); what is still
    // hopping after 2^max_rounds steps' worth of doubling sits on an isolated cycle; behind a dropped ruler: dropped
#pragma unroll
    for (u32 q = 0; q < 32; ++q) {
        const u32 r = threadIdx.x + 1024 * q;
        if (r >= nr) continue;
        const u32 e = mine[q];
        u64 l = ~0ull;
        if (!((live >> q) & 1u) && e != GASM_RANK_DEAD) {
            const u32 t = (e >> 16) & 0x7FFFu;                       // the terminal ruler (myself for a terminal entry)
            if (s_e[t] != GASM_RANK_DEAD) {
                const u32 ot = src[t];
                l = ((u64)(lo + (ot >> 16)) << 32) | GASM_LINK_DONE | (((e & 0xFFFFu) + (ot & 0xFFFFu)) & 0xFFFFu);
            }
        }
        link[lo + (r << rshift)] = l;
    }
}

// Tail edges publish their chain's length (in edges) at the head.  (grid-stride: the number of edges is only known on
// the device, *n_edges_p = dstart[last])
__global__ void __launch_bounds__(GASM_WG) k_chain_len(const u32* __restrict__ nxt, const u64* __restrict__ link, u32* __restrict__ clen,
                                                       const u32* __restrict__ n_edges_p) {
    const u32 n_edges = *n_edges_p;
    for (u32 i = blockIdx.x * GASM_WG + threadIdx.x; i < n_edges; i += gridDim.x * GASM_WG) {
        if (nxt[i] != GASM_NONE32) continue;
        const u64 l = link[i];
        const u32 a = (u32)(l >> 32);
        if (a == GASM_NONE32 || !(l & GASM_LINK_DONE)) continue;  // isolated cycle member: no contig starts there
        clen[a] = ((u32)l & 0x7FFFFFFFu) + 1;
    }
}

// Per segment: rank of every head among the segment's heads and the base offset of its contig inside the segment
// (contig length = k-1 + chain length).  One workgroup of 1024 threads per segment; each of the 16 waves owns a
// contiguous sixteenth of the edges and scans it on its own (DPP scans with a running carry, no barriers), the wave
// totals meet once in LDS, and a second sweep adds the offsets.
// Round 3: the segment directories and the build's report (k_seg_offsets, a launch of one wave) are the job of the LAST
// workgroup to finish — beside other steps' streaming kernels every dependent launch waits for a slot on the chip, and this
// one was 5 us of work behind 15-60 us of waiting.  `done`: a zeroed flag word of the build (k_bucket_gather zeroes it).
__global__ void __launch_bounds__(1024) k_contig_scan(GraphView gv, const u8* __restrict__ eflag, const u32* __restrict__ clen,
                                                      u32* __restrict__ e_cid, u64* __restrict__ e_coff,
                                                      u32* __restrict__ seg_ncontig, u64* __restrict__ seg_cbases,
                                                      u32* __restrict__ done, u32* __restrict__ seg_cstart, u64* __restrict__ seg_bstart,
                                                      const u32* __restrict__ flags, u32* __restrict__ report, u32 ticket) {
    __shared__ u32 s_last;
    __shared__ u32 s_cnt[16];
    __shared__ u64 s_bas[16];
    const u32 seg = blockIdx.x;
    const u32 nb = 1u << gv.bbits;
    const u32 lo = gv.dstart[seg * nb], hi = gv.dstart[(seg + 1) * nb];
    const u32 wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    const u32 per = ((hi - lo + 15) / 16 + 63) & ~63u;          // edges per wave, whole wave-rows
    const u32 wa = min(hi, lo + wv * per), wb = min(hi, wa + per);
    u32 ccarry = 0;
    u64 bcarry = 0;
    for (u32 base = wa; base < wb; base += 1024) {
        u32 ef[16];                                                // (32-bit: a byte array would be packed, one dependent load at a time)
        u32 cl[16];
#pragma unroll
        for (u32 q = 0; q < 16; ++q) {                           // 32 loads in flight, then the scans: the kernel is load latency
            const u32 i = base + q * 64 + ln;
            ef[q] = i < wb ? (u32)eflag[i] : 0u;
            cl[q] = i < wb ? clen[i] : 0u;
        }
        // (without this the compiler sinks each clen load under "is a head", i.e. behind its flag load: one dependent
        // round trip per row instead of 32 loads in flight)
#pragma unroll
        for (u32 q = 0; q < 16; ++q) __asm__ volatile("" : "+v"(cl[q]), "+v"(ef[q]));
#pragma unroll
        for (u32 q = 0; q < 16; ++q) {
            const u32 i = base + q * 64 + ln;
            const bool head = ef[q] & 1;
            const u32 len = head ? (u32)(gv.k - 1) + cl[q] : 0u;
            const u32 cinc = wave_incl_scan(head ? 1u : 0u);
            const u32 binc = wave_incl_scan(len);                // < 2^32 per 64 edges for any sane contig
            if (head) { e_cid[i] = ccarry + cinc - 1; e_coff[i] = bcarry + binc - len; }
            ccarry += wave_last(cinc);
            bcarry += wave_last(binc);
        }
    }
    if (ln == 0) { s_cnt[wv] = ccarry; s_bas[wv] = bcarry; }
    __syncthreads();
    u32 cbefore = 0, ctot = 0;
    u64 bbefore = 0, btot = 0;
#pragma unroll
    for (u32 w = 0; w < 16; ++w) {
        if (w < wv) { cbefore += s_cnt[w]; bbefore += s_bas[w]; }
        ctot += s_cnt[w]; btot += s_bas[w];
    }
    if (wv) {   // heads were only just written by this same thread: plain read-modify-write; 32 flag loads in flight
        for (u32 base = wa + ln; base < wb; base += 2048) {
            u32 ef[32];
#pragma unroll
            for (u32 q = 0; q < 32; ++q) ef[q] = base + 64 * q < wb ? (u32)eflag[base + 64 * q] : 0u;
#pragma unroll
            for (u32 q = 0; q < 32; ++q)
                if (ef[q] & 1) { e_cid[base + 64 * q] += cbefore; e_coff[base + 64 * q] += bbefore; }
        }
    }
    if (threadIdx.x == 0) {
        publish_u32(&seg_ncontig[seg], ctot); publish_u64(&seg_cbases[seg], btot);      // (no release fence: see count_in_last)
        s_last = count_in_last(done) ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last || wv) return;
    seg_offsets_wave(seg_ncontig, seg_cbases, gridDim.x, seg_cstart, seg_bstart, gv.dstart, nb, flags, report, ticket);
}

// Heads: make contig ids and offsets global; record offset and length per contig.
__global__ void __launch_bounds__(GASM_WG) k_contig_place(GraphView gv, const u8* __restrict__ eflag,
                                                          const u32* __restrict__ seg_cstart, const u64* __restrict__ seg_bstart,
                                                          u32* __restrict__ e_cid, u64* __restrict__ e_coff,
                                                          u64* __restrict__ c_off, u32 n_segments, u32 chunks) {
    if (blockIdx.x == 0 && threadIdx.x == 0) c_off[seg_cstart[n_segments]] = seg_bstart[n_segments];      // end of the last contig
    for_seg_edges(gv.dstart, 1u << gv.bbits, n_segments, chunks, [&](u32 seg, u32, u32, u32 i) {
        if (!(eflag[i] & 1)) return;
        const u32 cid = seg_cstart[seg] + e_cid[i];
        const u64 off = seg_bstart[seg] + e_coff[i];
        e_cid[i] = cid;
        e_coff[i] = off;
        c_off[cid] = off;
    });
}

// Contig text.  The edge at distance d from its chain's head ends at base (k-1) + d of the contig, and its key holds the
// k bases before that: so only every eighth edge of a chain (d % 8 == 0) and the last one write — the eight bases that end
// with their own, as ONE 8-byte store — and the head adds the rest of its node.  (One byte per edge was 3.8 M scattered store
// requests and as many gathers of the head's offset; the graph kernels are bound by the number of requests they send.)
__device__ __forceinline__ u32 klow16(u64 a) { return (u32)a & 0xFFFFu; }
__device__ __forceinline__ u32 klow16(const K128& a) { return (u32)a.lo & 0xFFFFu; }
template <class K>
__global__ void __launch_bounds__(GASM_WG) k_contig_emit(GraphView gv, const u64* __restrict__ link, const u32* __restrict__ nxt,
                                                         const u64* __restrict__ e_coff, u8* __restrict__ out, u32 n_segments,
                                                         u32 chunks) {
    // (segment-major: the stores into a segment's contigs merge in one L2)
    for_seg_edge_groups(gv.dstart, 1u << gv.bbits, n_segments, chunks, [&](u32, u32, u32, const u32 (&i)[GASM_EDGE_ILP], const bool (&ok)[GASM_EDGE_ILP]) {
        u64 l[GASM_EDGE_ILP], off[GASM_EDGE_ILP];
        u32 nx[GASM_EDGE_ILP];
        K key[GASM_EDGE_ILP];
        bool on[GASM_EDGE_ILP];
        const int k = gv.k;
        const bool wide = k >= 8;            // (a key of fewer than eight bases: every edge writes its own last base)
#pragma unroll
        for (int q = 0; q < GASM_EDGE_ILP; ++q) {
            l[q] = ok[q] ? link[i[q]] : ~0ull;
            nx[q] = ok[q] ? nxt[i[q]] : 0u;
        }
#pragma unroll
        for (int q = 0; q < GASM_EDGE_ILP; ++q) {
            const u32 a = (u32)(l[q] >> 32), d = (u32)l[q] & 0x7FFFFFFFu;
            on[q] = a != GASM_NONE32 && (l[q] & GASM_LINK_DONE) && (!wide || !(d & 7u) || nx[q] == GASM_NONE32);
            off[q] = on[q] ? e_coff[a] : 0ull;
            key[q] = reinterpret_cast<const K*>(gv.dk_key)[on[q] ? i[q] : 0u];
        }
#pragma unroll
        for (int q = 0; q < GASM_EDGE_ILP; ++q) {
            if (!on[q]) continue;
            const u32 d = (u32)l[q] & 0x7FFFFFFFu, low = klow16(key[q]);
            const u64 last = off[q] + (u32)(k - 1) + d;                 // where my own last base goes
            if (wide) {
                u64 txt = 0;
#pragma unroll
                for (u32 t = 0; t < 8; ++t) txt |= (u64)((0x54474341u >> (8u * ((low >> (2u * (7u - t))) & 3u))) & 0xFFu) << (8u * t);   // "ACGT"
                __builtin_memcpy(out + last - 7, &txt, 8);
            } else {
                out[last] = "ACGT"[low & 3u];
            }
            if ((u32)(l[q] >> 32) == i[q]) {                           // the head: the bases of its node the store above does not reach
                for (int j = 0; j < (wide ? k - 8 : k - 1); ++j) out[off[q] + j] = "ACGT"[klow2(kshr(key[q], 2 * (k - 1 - j)))];
            }
        }
    });
}
template __global__ void k_contig_emit<u64>(GraphView, const u64*, const u32*, const u64*, u8*, u32, u32);
template __global__ void k_contig_emit<K128>(GraphView, const u64*, const u32*, const u64*, u8*, u32, u32);
