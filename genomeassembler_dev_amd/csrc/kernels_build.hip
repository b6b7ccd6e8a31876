// kernels_build.hip — graph-construction kernels (gfx950): packing, k-mer bucket partition, LDS hash de-duplication,
// (k-1)-mer graph, list ranking, contig emission.  Integer / hash / gather work: no MFMA anywhere; the levers are
// coalesced HBM streams, LDS-resident tables and wave-level primitives.
//
// What each kernel restates of the reference (paths relative to the reference root):
//   k_pack_ascii      new surface (the reference keeps std::string)
//   k_bucket_hist /
//   k_bucket_scatter  lib/DeNovoAssembler.R:109-130 (every k-mer of every read) fused with the first half of the
//                     de-duplication that lib/DeNovoAssembler.cpp:104-122 does through its hash map
//   k_bucket_dedup    lib/DeNovoAssembler.cpp:104-122 (distinct edges) + multiplicities
//   k_node_flags      lib/DeNovoAssembler.cpp:125-169 (in/out degree over distinct edges, branching nodes)
//   k_edge_next       lib/DeNovoAssembler.cpp:172-189, one step of the walk: successor edge or stop
//   k_link_jump       the walk itself as pointer doubling (the reference walks node by node)
//   k_chain_len / k_contig_scan / k_contig_place / k_contig_emit
//                     lib/DeNovoAssembler.cpp:183-192: contig text, in sorted order (contigs start with distinct
//                     k-mers, so sorting contigs = sorting their first edges)
#include "device_utils.h"
#include "keyops.h"
#include "kernels.h"

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

// ================================================================================================================
// ASCII -> 2-bit.  One thread per output word (32 bases); 2 x 16-byte loads where the word is fully inside.
// ================================================================================================================
__global__ void __launch_bounds__(GASM_WG) k_pack_ascii(const u8* __restrict__ ascii, u64 nbases,
                                                        u64* __restrict__ words, u64 nwords, u32* __restrict__ err) {
    const u64 t = (u64)blockIdx.x * GASM_WG + threadIdx.x;
    if (t >= nwords) return;
    const u64 b0 = t << 5;
    u64 w = 0;
    bool ok = true;
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
    if (!ok && err) atomicOr(err, 1u);
}

// ================================================================================================================
// Tiles.  A tile is up to `ipt` (= GASM_WG / g) consecutive reads of one segment; g threads share a read and take
// k-mer start offsets lane, lane+g, ...  Segment of a tile: binary search in seg_tile_start (S+1 entries).
// ================================================================================================================
struct TileInfo { u32 seg; u64 r0; u32 nitems; };

__device__ __forceinline__ TileInfo tile_decode(const ReadSet& rs, u32 tile, u32 ipt) {
    const uint4 e = rs.tile_info[tile];        // uniform across the workgroup: a scalar load
    TileInfo ti;
    ti.seg = e.x;
    ti.nitems = e.y;
    ti.r0 = (u64)e.z | ((u64)e.w << 32);
    return ti;
}

__device__ __forceinline__ void read_span(const ReadSet& rs, u64 r, u64* p0, u32* len) {
    if (rs.fixed_len) { *p0 = r * rs.fixed_len; *len = rs.fixed_len; }
    else { const u64 a = rs.read_off[r]; *p0 = a; *len = (u32)(rs.read_off[r + 1] - a); }
}

// Thread `lane` of the g threads that share a read takes KT consecutive k-mer starts per round: offsets
// round*g*KT + lane*KT + [0, KT) (KT = 16 for 64-bit keys, 8 for 128-bit keys).  The windows come out of a few 64-bit
// words held in registers (Roll<K>, keyops.h), so a k-mer costs a funnel shift, not two loads.

// Count cube: for every tile, bucket, round q and wave w the number of k-mers wave w meets in round q of the tile
// that fall into the bucket — cube[(tile * nb + b) * rt4 + q * 4 + w], 16-bit (a wave-round holds at most 1024).
// The bucket of a k-mer is its first `bbits` bits (bbits <= 2(k-1), bbits <= 10), i.e. buckets are key ranges:
// concatenating sorted buckets gives a sorted segment.  A tile is `tr` groups of GASM_WG/g reads x `orr` offset rounds;
// round q = t * orr + o handles read group t, k-mer starts o*g*KT + lane*KT + [0, KT).  k_bucket_scatter walks the tile
// in exactly the same order, so it needs no counting of its own.
template <class K>
__global__ void __launch_bounds__(GASM_WG) k_tile_hist(ReadSet rs, int k, int bbits, u32 g, u32 tr, u32 orr, u32 n_tiles,
                                                       u16* __restrict__ cube) {
    constexpr u32 KT = KeyTraits<K>::KT;
    extern __shared__ u32 s_h[];   // [rt][4][nb]
    const u32 nb = 1u << bbits, ipt = GASM_WG / g, rt = tr * orr, rt4 = rt * 4;
    const u32 item = threadIdx.x / g, lane = threadIdx.x % g, wv = threadIdx.x >> 6;
    for (u32 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        for (u32 e = threadIdx.x; e < rt4 * nb; e += GASM_WG) s_h[e] = 0;
        __syncthreads();
        const TileInfo ti = tile_decode(rs, tile, ipt * tr);
        for (u32 t = 0; t < tr; ++t) {
            const u32 it = t * ipt + item;
            if (it >= ti.nitems) break;
            u64 p0; u32 len;
            read_span(rs, ti.r0 + it, &p0, &len);
            const u32 nk = len >= (u32)k ? len - k + 1 : 0;
            for (u32 o = 0; o < orr; ++o) {
                const u32 off0 = o * g * KT + lane * KT;
                if (off0 >= nk) break;
                Roll<K> r;
                r.load(rs.words, p0 + off0);
                u32* row = s_h + ((t * orr + o) * 4 + wv) * nb;
#pragma unroll
                for (u32 j = 0; j < KT; ++j) {
                    if (off0 + j < nk) {
                        const u32 bkt = bbits ? (u32)(r.top(j) >> (64 - bbits)) : 0u;
                        atomicAdd(&row[bkt], 1u);
                    }
                }
            }
        }
        __syncthreads();
        u16* dst = cube + (u64)tile * nb * rt4;
        for (u32 e = threadIdx.x; e < rt4 * nb; e += GASM_WG) {
            const u32 b = e / rt4, qw = e - b * rt4;
            dst[e] = (u16)s_h[qw * nb + b];
        }
        __syncthreads();
    }
}
template __global__ void k_tile_hist<u64>(ReadSet, int, int, u32, u32, u32, u32, u16*);
template __global__ void k_tile_hist<K128>(ReadSet, int, int, u32, u32, u32, u32, u16*);

// Per segment and bucket: tile totals from the cube, their running sum toff[tile * nb + b] (offset of the tile inside
// its (segment,bucket) range) and the bucket total hist[seg * nb + b].  One workgroup per segment, thread = bucket.
// Each tile's run is rounded up to 16 k-mers = one 128-byte line, so no cache line is shared by two workgroups: partial
// line writes from different L2s cost more than half the store bandwidth (measured: 2.1 vs 5.9 TB/s).
__global__ void __launch_bounds__(1024) k_tile_scan(ReadSet rs, int bbits, u32 rt4, const u16* __restrict__ cube,
                                                    u32* __restrict__ toff, u32* __restrict__ hist) {
    const u32 nb = 1u << bbits, seg = blockIdx.x;
    const u32 t0 = rs.seg_tile_start[seg], t1 = rs.seg_tile_start[seg + 1];
    for (u32 b = threadIdx.x; b < nb; b += blockDim.x) {
        u32 run = 0;
        for (u32 t = t0; t < t1; ++t) {
            const u16* c = cube + ((u64)t * nb + b) * rt4;
            u32 tot = 0;
            if ((rt4 & 7) == 0) {
                for (u32 e = 0; e < rt4; e += 8) {
                    const uint4 v = *reinterpret_cast<const uint4*>(c + e);
                    tot += (v.x & 0xFFFF) + (v.x >> 16) + (v.y & 0xFFFF) + (v.y >> 16) + (v.z & 0xFFFF) + (v.z >> 16) + (v.w & 0xFFFF) + (v.w >> 16);
                }
            } else {
                for (u32 e = 0; e < rt4; ++e) tot += c[e];
            }
            toff[(u64)t * nb + b] = run;
            run += (tot + 15u) & ~15u;      // every (tile, bucket) run starts on a 128-byte line (filler: see k_bucket_scatter)
        }
        hist[(u64)seg * nb + b] = run;
    }
}

// Exclusive scan of n u32 into n+1 outputs of type TO; one workgroup of 1024 threads, running carry.
template <class TO>
__global__ void __launch_bounds__(1024) k_scan_excl(const u32* __restrict__ in, TO* __restrict__ out, u32 n) {
    __shared__ u32 s_tmp[16];
    TO carry = 0;
    for (u32 base = 0; base < n; base += 1024) {
        const u32 i = base + threadIdx.x;
        const u32 v = i < n ? in[i] : 0u;
        u32 tot;
        const u32 ex = block_excl_scan<1024>(v, s_tmp, &tot);
        if (i < n) out[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) out[n] = carry;
}
template __global__ void k_scan_excl<u64>(const u32*, u64*, u32);
template __global__ void k_scan_excl<u32>(const u32*, u32*, u32);

__global__ void k_copy_u64(const u64* __restrict__ a, u64* __restrict__ b, u32 n) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
}

// ================================================================================================================
// Scatter: every k-mer of every read is written once, 8 bytes, into its (segment,bucket) range of `keys`.
// Where a wave's k-mers of one round go is fully determined beforehand: bstart[seg,bucket] + toff[tile,bucket]
// (k_tile_scan) + the cube counts of the earlier rounds and the lower waves (k_tile_hist).  So there is no global
// atomic, no counting pass and no workgroup barrier: each wave ranks its 1024 k-mers of a round into wave-private LDS
// bins (one ds_add_rtn per k-mer on a cursor that starts at the bin's staging offset), then streams the staged keys
// out bucket by bucket — the global stores of one bucket are consecutive — looking up one combined 64-bit base per key.
// The output layout is deterministic.  LDS per wave: 1024 keys (8 KB) + nb * 12 + 8.
// ================================================================================================================
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// makes the compiler wait for pending loads of these registers at this point
__device__ __forceinline__ void touch_regs(u32 a, u32 b, u32 c, u32 d, u32 e, u32 f, u32 g, u32 h, u32 i, u32 j) {
    __asm__ volatile("" :: "v"(a), "v"(b), "v"(c), "v"(d), "v"(e), "v"(f), "v"(g), "v"(h), "v"(i), "v"(j));
}

template <class K>
__global__ void __launch_bounds__(GASM_WG, 4) k_bucket_scatter(ReadSet rs, int k, int bbits, u32 g, u32 tr, u32 orr, u32 n_tiles,
                                                            const u64* __restrict__ bstart, const u32* __restrict__ toff,
                                                            const u16* __restrict__ cube, K* __restrict__ keys, u64 scratch,
                                                            unsigned long long* __restrict__ stamps) {
    extern __shared__ __align__(16) unsigned char s_raw[];
    // diagnostic build only (-DGASM_SCATTER_STAMPS): shader-clock totals of wave 0 per phase, summed over workgroups.
    // Compiled out otherwise — the accumulators cost 14 registers, and the kernel sits right at 128.
#ifdef GASM_SCATTER_STAMPS
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    auto phase = [&](int i) {
        if (stamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tacc[i] += t - tlast; tlast = t; }
    };
#else
    auto phase = [](int) {};
#endif
    constexpr u32 KT = KeyTraits<K>::KT;
    constexpr u32 WSTAGE = KT * 64;
    const u32 rt = tr * orr, rt4 = rt * 4;
    const u32 nb = 1u << bbits, ipt = GASM_WG / g;
    const u32 wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    K* s_key = reinterpret_cast<K*>(s_raw) + wv * WSTAGE;                                                   // 4 * WSTAGE
    u64* s_comb = reinterpret_cast<u64*>(reinterpret_cast<K*>(s_raw) + 4 * WSTAGE) + wv * nb;               // 4 * nb
    u32* s_cur = reinterpret_cast<u32*>(reinterpret_cast<u64*>(reinterpret_cast<K*>(s_raw) + 4 * WSTAGE) + 4 * nb) + wv * (nb + 2);   // 4 * (nb + 2): + dummy bin
    const u32 item = threadIdx.x / g, lane = threadIdx.x % g;
    const int bshift = 2 * k - bbits;
    const bool fast = rt4 == 16 && nb <= 64;     // one bucket per lane, the bucket's 16 counts live in registers

    // a workgroup takes a contiguous range of tiles (neighbouring runs of a bucket then come from the same L2)
    const u32 per_wg = (n_tiles + gridDim.x - 1) / gridDim.x;
    const u32 tile_end = min(n_tiles, (blockIdx.x + 1) * per_wg);
    for (u32 tile = blockIdx.x * per_wg; tile < tile_end; ++tile) {
        const TileInfo ti = tile_decode(rs, tile, ipt * tr);
        u32 cw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        u64 tbase = 0;
        if (fast && ln < nb) {
            const uint4* c4 = reinterpret_cast<const uint4*>(cube + ((u64)tile * nb + ln) * 16);
            const uint4 v0 = c4[0], v1 = c4[1];
            cw[0] = v0.x; cw[1] = v0.y; cw[2] = v0.z; cw[3] = v0.w; cw[4] = v1.x; cw[5] = v1.y; cw[6] = v1.z; cw[7] = v1.w;
            tbase = bstart[(u64)ti.seg * nb + ln] + toff[(u64)tile * nb + ln];
        }
        // the words a thread needs in round q, and how many of its KT starts are k-mers
        auto fetch_round = [&](u32 q, Roll<K>& r, u32& nv) {
            const u32 t = q / orr, o = q - t * orr;
            const u32 it = t * ipt + item;
            u64 p0 = 0; u32 nk = 0;
            if (it < ti.nitems) {
                u32 len;
                read_span(rs, ti.r0 + it, &p0, &len);
                nk = len >= (u32)k ? len - k + 1 : 0;
            }
            const u32 off0 = o * g * KT + lane * KT;
            nv = off0 < nk ? min(nk - off0, KT) : 0u;
            r.load(rs.words, nv ? p0 + off0 : 0);
        };
        Roll<K> rl;
        u32 nv;
        fetch_round(0, rl, nv);
        // Every load so far is waited for here, once, outside the round loop: a wait placed inside the loop would also
        // cover the loop's own stores (see the note at the prefetch below).
        touch_regs(cw[0], cw[1], cw[2], cw[3], cw[4], cw[5], cw[6], cw[7], (u32)tbase, (u32)(tbase >> 32));
        rl.touch();
        // filler behind the tile's run of every bucket, up to the next 128-byte line (the de-duplication skips it)
        if (wv == 0) {
            for (u32 b = ln; b < nb; b += 64) {
                u32 tot = 0;
                u64 base;
                if (fast) {
#pragma unroll
                    for (u32 e = 0; e < 8; ++e) tot += (cw[e] & 0xFFFF) + (cw[e] >> 16);
                    base = tbase;
                } else {
                    const u16* c = cube + ((u64)tile * nb + b) * rt4;
                    for (u32 e = 0; e < rt4; ++e) tot += c[e];
                    base = bstart[(u64)ti.seg * nb + b] + toff[(u64)tile * nb + b];
                }
                for (u32 i = tot; i < ((tot + 15u) & ~15u); ++i) keys[base + i] = key_empty<K>();
            }
        }
        u32 run_before = 0;   // fast path: k-mers of this lane's bucket in earlier rounds (all waves)
        phase(0);
        for (u32 q = 0; q < rt; ++q) {
            // ---- where this wave's k-mers of round q go, bucket by bucket
            u32 kcar = 0;
            for (u32 b0 = 0; b0 < nb; b0 += 64) {
                const u32 b = b0 + ln;
                u32 own = 0;
                u64 base = 0;
                if (fast) {
                    // counts of round q: words 2q, 2q+1 = waves (0,1), (2,3)
                    u32 w01 = 0, w23 = 0;
#pragma unroll
                    for (u32 e = 0; e < 4; ++e) if (e == q) { w01 = cw[2 * e]; w23 = cw[2 * e + 1]; }
                    const u32 c0 = w01 & 0xFFFF, c1 = w01 >> 16, c2 = w23 & 0xFFFF, c3 = w23 >> 16;
                    own = wv == 0 ? c0 : wv == 1 ? c1 : wv == 2 ? c2 : c3;
                    const u32 lower = (wv > 0 ? c0 : 0) + (wv > 1 ? c1 : 0) + (wv > 2 ? c2 : 0);
                    base = tbase + run_before + lower;
                    run_before += c0 + c1 + c2 + c3;
                } else if (b < nb) {
                    const u16* c = cube + ((u64)tile * nb + b) * rt4;
                    u32 before = 0;
                    for (u32 e = 0; e < q * 4 + wv; ++e) before += c[e];    // earlier rounds, and lower waves of this one
                    own = c[q * 4 + wv];
                    base = bstart[(u64)ti.seg * nb + b] + toff[(u64)tile * nb + b] + before;
                }
                const u32 inc = wave_incl_scan(own);
                const u32 off = kcar + inc - own;                    // staging offset of the bin
                if (b < nb) {
                    s_cur[b] = off;
                    s_comb[b] = base - off;                          // global index = s_comb[bucket] + staging index
                }
                kcar += wave_last(inc);
            }
            const u32 staged = kcar;
            wave_sync_lds();
            phase(1);
            // ---- rank and stage.  The ds_add_rtn of a thread are issued back to back (a position past the end of the read
            // ranks into a dummy bin) and waited for once; a branch per k-mer would make them dependent LDS round trips.
            // the words of this round were requested before the previous round's KT stores (first round: already waited for)
            rl.template wait_all_but<KT>();
            K key[KT];
            u32 idx[KT];
#ifdef GASM_SCATTER_STAMPS
            if (stamps) { key[0] = rl.key(0, k); if (kis_empty(key[0])) tacc[5] += 1; phase(2); }   // words have arrived
#endif
#pragma unroll
            for (u32 j = 0; j < KT; ++j) {
                key[j] = rl.key(j, k);
                const u32 bkt = bbits ? (u32)(rl.top(j) >> (64 - bbits)) : 0u;
                idx[j] = atomicAdd(&s_cur[j < nv ? bkt : nb], 1u);
            }
            // The next round's words are requested now, ahead of this round's stores: memory operations of a wave
            // retire in order (one counter for loads and stores), so a load issued after the stores would wait for all
            // of them.  The flush below issues exactly KT stores whatever `staged` is — the wait for these words can
            // then be "all but the last KT operations" instead of "everything".
            Roll<K> rn;
            u32 nvn;
            fetch_round(min(q + 1, rt - 1), rn, nvn);
#ifdef GASM_SCATTER_STAMPS
            if (stamps) { if (idx[KT - 1] == 0xFFFFFFFFu) tacc[5] += 1; phase(3); }                  // atomics have returned
#endif
#pragma unroll
            for (u32 j = 0; j < KT; ++j)
                if (j < nv) s_key[idx[j]] = key[j];
            wave_sync_lds();
            phase(4);
            // ---- stream out, four keys per thread at a time (their LDS reads overlap); lanes past the end store to a
            // scratch line behind the key array instead of branching
#pragma unroll
            for (u32 u0 = 0; u0 < KT; u0 += 4) {
                K kk[4];
                u64 cb[4];
#pragma unroll
                for (u32 u = 0; u < 4; ++u) kk[u] = s_key[ln + 64 * (u0 + u)];
#pragma unroll
                for (u32 u = 0; u < 4; ++u) cb[u] = s_comb[bbits ? (kfield(kk[u], bshift) & (nb - 1)) : 0u];
#pragma unroll
                for (u32 u = 0; u < 4; ++u) {
                    const u32 i = ln + 64 * (u0 + u);
                    keys[i < staged ? cb[u] + i : scratch + ln] = kk[u];
                }
            }
            wave_sync_lds();
            phase(5);
            rl = rn;
            nv = nvn;
        }
    }
#ifdef GASM_SCATTER_STAMPS
    if (stamps && threadIdx.x == 0)
        for (int i = 0; i < 6; ++i) atomicAdd(&stamps[i], tacc[i]);
#endif
}
template __global__ void k_bucket_scatter<u64>(ReadSet, int, int, u32, u32, u32, u32, const u64*, const u32*, const u16*, u64*, u64, unsigned long long*);
template __global__ void k_bucket_scatter<K128>(ReadSet, int, int, u32, u32, u32, u32, const u64*, const u32*, const u16*, K128*, u64, unsigned long long*);

// ================================================================================================================
// De-duplicate one bucket: stream its keys through an LDS table (count per distinct key), then order the distinct keys
// and write them and their multiplicities back over the start of the bucket's own range.
//
// The table is TBL/4 sets of four slots.  A key lives in the first set with room, counted from its home set; sets fill
// left to right.  One probe = the whole set, so unless a home set has overflowed (rare at <= 40 % load) a key is found
// in the first probe — which matters because a wave moves at the pace of its slowest lane: with one-slot probing some
// lane of 64 always needs a second and third round.
//   64-bit keys : slot claimed by a 64-bit CAS on the key itself (EMPTY -> key);
//   128-bit keys: no 128-bit CAS in LDS — the slot's count word is the lock: CAS 0 -> LOCKED, write the key, then
//                 count = 1.  Readers compare keys only in slots whose count says "ready"; a LOCKED slot means "try
//                 again" (never a spin inside a divergent branch: the caller's round loop simply comes back).
// Ordering: a counting sort on the key bits below the bucket prefix (TBL/4 bins; close to uniform there) puts every key
// into its bin's range, bins of more than one key are finished by a per-bin insertion sort; if any bin is long (skewed
// keys) the workgroup falls back to a bitonic sort.  The bin offsets are kept as the fine directory of the graph
// kernels.  A bucket with more than 11/16*TBL distinct keys raises *overflow (the host re-partitions).
// launch bounds: 2048 slots of 64-bit keys fit five workgroups per CU in LDS (<= 96 registers), the other variants three.
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
    constexpr u32 NSETS = TBL / 4;
    const u64x2 c01 = lds_load128(&t_key[4 * set]);
    const u64x2 c23 = lds_load128(&t_key[4 * set + 2]);
    int slot = c01.x == key ? 0 : c01.y == key ? 1 : c23.x == key ? 2 : c23.y == key ? 3 : -1;
    if (slot < 0) {
        const int emp = c01.x == GASM_EMPTY64 ? 0 : c01.y == GASM_EMPTY64 ? 1 : c23.x == GASM_EMPTY64 ? 2 : c23.y == GASM_EMPTY64 ? 3 : -1;
        if (emp < 0) { set = (set + 1) & (NSETS - 1); return false; }
        const u64 old = atomicCAS(reinterpret_cast<unsigned long long*>(&t_key[4 * set + emp]), (unsigned long long)GASM_EMPTY64,
                                  (unsigned long long)key);
        if (old == GASM_EMPTY64) atomicAdd(n_distinct, 1u);
        else if (old != key) return false;       // someone else took the slot: look at the set again
        slot = emp;
    }
    atomicAdd(&t_cnt[4 * set + slot], 1u);
    return true;
}
template <int TBL>
__device__ __forceinline__ bool dedup_step(K128* t_key, u32* t_cnt, u32* n_distinct, const K128& key, u32& set) {
    constexpr u32 NSETS = TBL / 4;
    const uint4 c = lds_load128u(&t_cnt[4 * set]);            // counts first: a ready count guarantees a complete key
    const u32 cc[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (cc[q] == GASM_SLOT_LOCKED) return false;          // being written: may be this very key
        if (cc[q] == 0) {
            if (atomicCAS(&t_cnt[4 * set + q], 0u, GASM_SLOT_LOCKED) != 0u) return false;
            t_key[4 * set + q] = key;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __hip_atomic_store(&t_cnt[4 * set + q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            atomicAdd(n_distinct, 1u);
            return true;
        }
        const u64x2 kq = lds_load128(&t_key[4 * set + q]);
        if (kq.x == key.hi && kq.y == key.lo) { atomicAdd(&t_cnt[4 * set + q], 1u); return true; }
    }
    set = (set + 1) & (NSETS - 1);
    return false;
}

template <class K, int TBL>
__global__ void __launch_bounds__(GASM_WG, (TBL == 2048 && sizeof(K) == 8) ? 5 : 3)
k_bucket_dedup(K* __restrict__ keys, u32* __restrict__ mult, const u64* __restrict__ bstart, u32* __restrict__ bucket_d,
               u32* __restrict__ overflow, u16* __restrict__ fdir, int low_bits, int dbg, unsigned long long* __restrict__ stamps) {
    constexpr int LIMIT = TBL / 16 * 11;
    constexpr int BINS = TBL / 4;
    constexpr int SL = TBL / GASM_WG;
    constexpr int LOG_TBL = TBL == 4096 ? 12 : 11;
    constexpr int LOG_SETS = LOG_TBL - 2;
    constexpr u32 NSETS = TBL / 4;
    constexpr bool WIDE = sizeof(K) == 16;
    static_assert(TBL == 4096 || TBL == 2048, "table size");
    __shared__ __align__(32) K t_key[TBL];
    __shared__ __align__(16) u32 t_cnt[TBL];
    __shared__ u32 s_start[BINS];
    __shared__ u32 s_cur[BINS];
    __shared__ u32 s_tmp[8];
    // diagnostic only (stamps == nullptr in production): per-phase wave-0 tick totals, summed over workgroups
    unsigned long long tph = stamps ? wall_clock64() : 0ull;
    auto phase = [&](int i) {
        if (stamps && threadIdx.x == 0) { const unsigned long long t = wall_clock64(); atomicAdd(&stamps[i], t - tph); tph = t; }
    };
    const u32 bucket = blockIdx.x;
    const u64 beg = bstart[bucket], end = bstart[bucket + 1];
    const u64 n = end - beg;
    for (u32 i = threadIdx.x; i < TBL; i += GASM_WG) { t_key[i] = key_empty<K>(); t_cnt[i] = 0; }
    for (u32 i = threadIdx.x; i < BINS; i += GASM_WG) s_start[i] = 0;
    if (threadIdx.x == 0) { s_tmp[4] = 0; s_tmp[5] = 0; s_tmp[6] = 0; }  // [4] distinct so far, [5] overflow, [6] longest bin
    __syncthreads();
    phase(0);
    // stream: four 16-byte loads per thread in flight (the loop is latency-bound otherwise), each fully coalesced across
    // the wave.  Bucket ranges start and end on 128-byte lines (filler keys = EMPTY are skipped).
    constexpr int KPL = 64 / sizeof(K);          // keys per thread and iteration: 8 or 4
    const u64 nch = n * sizeof(K) / 16;          // 16-byte chunks in the bucket
    const uint4* src = reinterpret_cast<const uint4*>(keys + beg);
    // the loads of the next iteration are issued before this iteration's keys go into the table, so the table work
    // (LDS latency) and the HBM latency overlap inside every wave
    uint4 v[4];
    auto fetch = [&](u64 c) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = c + (u64)q * GASM_WG < nch ? src[c + (u64)q * GASM_WG] : make_uint4(~0u, ~0u, ~0u, ~0u);
    };
    if (threadIdx.x < nch) fetch(threadIdx.x);
    for (u64 c = threadIdx.x; c < nch; c += 4 * GASM_WG) {
        if (c == (u64)threadIdx.x + 4 * GASM_WG) phase(1);   // first iteration (table fill) done
        K kx[KPL];
        if constexpr (WIDE) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { kx[q].hi = (u64)v[q].x | ((u64)v[q].y << 32); kx[q].lo = (u64)v[q].z | ((u64)v[q].w << 32); }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) { kx[2 * q] = (u64)v[q].x | ((u64)v[q].y << 32); kx[2 * q + 1] = (u64)v[q].z | ((u64)v[q].w << 32); }
        }
        if (c + 4 * GASM_WG < nch) fetch(c + 4 * GASM_WG);
        if ((dbg & 3) == 1) { u64 x = 0; for (int q = 0; q < KPL; ++q) x ^= khash(kx[q]); if (x == 0x1234567) s_tmp[6] = 1; continue; }
        if (__hip_atomic_load(&s_tmp[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > (u32)LIMIT) { s_tmp[5] = 1; break; }
        if constexpr (!WIDE) {
            // two batches of four keys: the four home sets are read together (8 x ds_read_b128 in flight), hits are
            // counted, the few keys that are not done loop on their own
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                u32 set[4];
                u64x2 c01[4], c23[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    set[q] = khash(kx[4 * h + q]) >> (32 - LOG_SETS);
                    c01[q] = lds_load128(&t_key[4 * set[q]]);
                    c23[q] = lds_load128(&t_key[4 * set[q] + 2]);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u64 key = kx[4 * h + q];
                    if (key == GASM_EMPTY64) continue;
                    const int slot = c01[q].x == key ? 0 : c01[q].y == key ? 1 : c23[q].x == key ? 2 : c23[q].y == key ? 3 : -1;
                    if (slot >= 0) atomicAdd(&t_cnt[4 * set[q] + slot], 1u);
                    else {
                        u32 st = set[q];
                        bool ok = false;
                        for (u32 probe = 0; probe < 8 * NSETS && !ok; ++probe) ok = dedup_step<TBL>(t_key, t_cnt, &s_tmp[4], key, st);
                        if (!ok) s_tmp[5] = 1;
                    }
                }
            }
        } else {
            // 128-bit keys: the four home sets' count words and first two slots are read together; anything not a hit in
            // those goes through the step function
            u32 set[4];
            uint4 cn[4];
            u64x2 k0[4], k1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                set[q] = khash(kx[q]) >> (32 - LOG_SETS);
                cn[q] = lds_load128u(&t_cnt[4 * set[q]]);
                k0[q] = lds_load128(&t_key[4 * set[q]]);
                k1[q] = lds_load128(&t_key[4 * set[q] + 1]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const K128 key = kx[q];
                if (kis_empty(key)) continue;
                const bool r0 = cn[q].x != 0 && cn[q].x != GASM_SLOT_LOCKED, r1 = cn[q].y != 0 && cn[q].y != GASM_SLOT_LOCKED;
                if (r0 && k0[q].x == key.hi && k0[q].y == key.lo) atomicAdd(&t_cnt[4 * set[q]], 1u);
                else if (r0 && r1 && k1[q].x == key.hi && k1[q].y == key.lo) atomicAdd(&t_cnt[4 * set[q] + 1], 1u);
                else {
                    u32 st = set[q];
                    bool ok = false;
                    for (u32 probe = 0; probe < 8 * NSETS && !ok; ++probe) ok = dedup_step<TBL>(t_key, t_cnt, &s_tmp[4], key, st);
                    if (!ok) s_tmp[5] = 1;
                }
            }
        }
    }
    phase(2);
    __syncthreads();
    phase(3);
    if (s_tmp[5] || s_tmp[4] > (u32)LIMIT) {
        if (threadIdx.x == 0) { atomicExch(overflow, 1u); bucket_d[bucket] = 0; }
        return;
    }
    const u32 d = s_tmp[4];
    if ((dbg & 3) == 1 || (dbg & 3) == 2) { if (threadIdx.x == 0) bucket_d[bucket] = d; return; }
    // ---- every thread pulls its slots (stride 256: conflict-free) into registers and bins them
    const int bshift = low_bits > (LOG_TBL - 2) ? low_bits - (LOG_TBL - 2) : 0;
    K rk[SL];
    u32 rc[SL];
#pragma unroll
    for (int q = 0; q < SL; ++q) {
        rk[q] = t_key[q * GASM_WG + threadIdx.x];
        rc[q] = t_cnt[q * GASM_WG + threadIdx.x];
        if (WIDE && rc[q] == 0) rk[q] = key_empty<K>();      // 128-bit tables mark free slots by the count
        if (!kis_empty(rk[q])) atomicAdd(&s_start[kfield(rk[q], bshift) & (BINS - 1)], 1u);
    }
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
            fdir[(u64)bucket * (BINS + 1) + threadIdx.x * PER + q] = (u16)ex;   // fine directory for the graph kernels
            ex += c[q];
        }
        if (threadIdx.x == GASM_WG - 1) fdir[(u64)bucket * (BINS + 1) + BINS] = (u16)ex;
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
    phase(4);
    for (u32 i = threadIdx.x; i < d; i += GASM_WG) { keys[beg + i] = t_key[i]; mult[beg + i] = t_cnt[i]; }
    if (threadIdx.x == 0) bucket_d[bucket] = d;
    phase(5);
    if (stamps && threadIdx.x == 0) stamps[8 + 3 * (u64)blockIdx.x + 1] = wall_clock64();
}
template __global__ void k_bucket_dedup<u64, 4096>(u64*, u32*, const u64*, u32*, u32*, u16*, int, int, unsigned long long*);
template __global__ void k_bucket_dedup<u64, 2048>(u64*, u32*, const u64*, u32*, u32*, u16*, int, int, unsigned long long*);
template __global__ void k_bucket_dedup<K128, 2048>(K128*, u32*, const u64*, u32*, u32*, u16*, int, int, unsigned long long*);

// Gather the per-bucket distinct runs into the dense per-segment arrays.
template <class K>
__global__ void __launch_bounds__(GASM_WG) k_bucket_gather(const K* __restrict__ keys, const u32* __restrict__ mult,
                                                           const u64* __restrict__ bstart, const u32* __restrict__ dstart,
                                                           K* __restrict__ dk_key, u32* __restrict__ dk_cnt) {
    const u32 bucket = blockIdx.x;
    const u64 src = bstart[bucket];
    const u32 dst = dstart[bucket], d = dstart[bucket + 1] - dst;
    for (u32 i = threadIdx.x; i < d; i += GASM_WG) { dk_key[dst + i] = keys[src + i]; dk_cnt[dst + i] = mult[src + i]; }
}
template __global__ void k_bucket_gather<u64>(const u64*, const u32*, const u64*, const u32*, u64*, u32*);
template __global__ void k_bucket_gather<K128>(const K128*, const u32*, const u64*, const u32*, K128*, u32*);

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

// flag bit0: the edge's source node is a branching node (in != 1 or out != 1); it has out-edges by construction.
template <class K>
__global__ void __launch_bounds__(GASM_WG) k_node_flags(GraphView gv, u8* __restrict__ eflag) {
    const K* dk = reinterpret_cast<const K*>(gv.dk_key);
    const u32 seg = blockIdx.y;
    const u32 nb = 1u << gv.bbits;
    const u32 lo = gv.dstart[seg * nb], hi = gv.dstart[(seg + 1) * nb];
    const u32 i = lo + blockIdx.x * GASM_WG + threadIdx.x;
    if (i >= hi) return;
    const K key = dk[i];
    const K u = kshr(key, 2);
    // out-degree of u: the run of keys sharing key>>2 is contiguous in the sorted list
    u32 outd = 1;
    for (u32 j = i; j > lo && keq(kshr(dk[j - 1], 2), u); --j) ++outd;
    for (u32 j = i + 1; j < hi && keq(kshr(dk[j], 2), u); ++j) ++outd;
    // in-degree of u: distinct k-mers x·u
    u32 ind = 0;
    const int sh = 2 * (gv.k - 1);
#pragma unroll
    for (u64 x = 0; x < 4; ++x) ind += kmer_exists<K>(gv, seg, kor(kshl(key_from_u64<K>(x), sh), u));
    eflag[i] = (ind != 1 || outd != 1) ? 1 : 0;
}
template __global__ void k_node_flags<u64>(GraphView, u8*);
template __global__ void k_node_flags<K128>(GraphView, u8*);

// Successor edge of every edge (GASM_NONE32 when the walk stops at its target), and the initial ancestor links:
// link = ancestor << 32 | done << 31 | distance, done = "the ancestor is the head of the chain".  Heads are their own
// ancestor at distance 0.  The done bit travels with the link, so pointer doubling needs one gather per round.
template <class K>
__global__ void __launch_bounds__(GASM_WG) k_edge_next(GraphView gv, const u8* __restrict__ eflag, u32* __restrict__ nxt,
                                                       u64* __restrict__ link) {
    const K* dk = reinterpret_cast<const K*>(gv.dk_key);
    const u32 seg = blockIdx.y;
    const u32 nb = 1u << gv.bbits;
    const u32 lo = gv.dstart[seg * nb], hi = gv.dstart[(seg + 1) * nb];
    const u32 i = lo + blockIdx.x * GASM_WG + threadIdx.x;
    if (i >= hi) return;
    const K key = dk[i];
    const int sh = 2 * (gv.k - 1);
    const K v = klowbits(key, sh);
    const K t = kshl(v, 2);  // smallest k-mer with prefix v
    // the run of v may continue into the next bucket only if bbits > 2(k-1), which the host never chooses
    u32 bhi;
    const u32 j = graph_lower_bound<K>(gv, seg, t, &bhi);
    u32 n = GASM_NONE32;
    if (j < bhi && keq(kshr(dk[j], 2), v) && !(eflag[j] & 1)) n = j;  // v has out-edges and is not branching
    nxt[i] = n;
    const u64 me_head = (eflag[i] & 1) ? GASM_LINK_DONE : 0ull;
    if (n != GASM_NONE32) link[n] = ((u64)i << 32) | me_head | 1ull;
    if (me_head) link[i] = ((u64)i << 32) | GASM_LINK_DONE;
}
template __global__ void k_edge_next<u64>(GraphView, const u8*, u32*, u64*);
template __global__ void k_edge_next<K128>(GraphView, const u8*, u32*, u64*);

// One round of pointer doubling towards the head of the chain.  In place and asynchronous: a link is always a
// consistent (ancestor, done, distance) triple because it is read and written as one 64-bit word.
__global__ void __launch_bounds__(GASM_WG) k_link_jump(u64* __restrict__ link, u32 n_edges) {
    const u32 i = blockIdx.x * GASM_WG + threadIdx.x;
    if (i >= n_edges) return;
    const u64 l = link[i];
    const u32 a = (u32)(l >> 32);
    if (a == GASM_NONE32 || (l & GASM_LINK_DONE)) return;
    const u64 la = *reinterpret_cast<volatile const u64*>(&link[a]);
    if ((u32)(la >> 32) == GASM_NONE32) return;
    link[i] = (la & (0xFFFFFFFF00000000ull | GASM_LINK_DONE)) | (u64)(((u32)l + (u32)la) & 0x7FFFFFFFu);
}

// All rounds of pointer doubling for one segment inside one workgroup (no launch per round, early exit when every
// chain has reached its head).  Links are read and written with relaxed workgroup-scope atomics so the
// updates other waves of this workgroup made in the same round or the previous one are seen (workgroup scope: one
// workgroup = one CU = one L1); a stale value would still be a valid (ancestor, distance) pair.  Members of isolated cycles never reach a head: they are dropped (ancestor =
// none) once their distance exceeds the segment's edge count.
#define GASM_RANK_BATCH 8
__global__ void __launch_bounds__(1024) k_link_rank_seg(GraphView gv, u64* __restrict__ link, int max_rounds) {
    __shared__ u32 s_active;
    const u32 seg = blockIdx.x;
    const u32 nb = 1u << gv.bbits;
    const u32 lo = gv.dstart[seg * nb], hi = gv.dstart[(seg + 1) * nb];
    const u32 n = hi - lo;
    for (int r = 0; r < max_rounds; ++r) {
        if (threadIdx.x == 0) s_active = 0;
        __syncthreads();
        bool any = false;
        for (u32 base = lo + threadIdx.x; base < hi; base += 1024 * GASM_RANK_BATCH) {
            u64 l[GASM_RANK_BATCH], la[GASM_RANK_BATCH];
            bool act[GASM_RANK_BATCH];
#pragma unroll
            for (int q = 0; q < GASM_RANK_BATCH; ++q) {
                const u32 i = base + q * 1024;
                l[q] = i < hi ? __hip_atomic_load(&link[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : ~0ull;
            }
#pragma unroll
            for (int q = 0; q < GASM_RANK_BATCH; ++q) {
                const u32 a = (u32)(l[q] >> 32);
                act[q] = a != GASM_NONE32 && !(l[q] & GASM_LINK_DONE);
                la[q] = act[q] ? __hip_atomic_load(&link[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0ull;
            }
#pragma unroll
            for (int q = 0; q < GASM_RANK_BATCH; ++q) {
                if (!act[q]) continue;
                const u32 i = base + q * 1024;
                const u32 a2 = (u32)(la[q] >> 32);
                const u32 d = ((u32)l[q] & 0x7FFFFFFFu) + ((u32)la[q] & 0x7FFFFFFFu);
                u64 nl;
                if (a2 == GASM_NONE32 || d > n) nl = ~0ull;            // on an isolated cycle
                else { nl = ((u64)a2 << 32) | (la[q] & GASM_LINK_DONE) | d; any = any || !(la[q] & GASM_LINK_DONE); }
                __hip_atomic_store(&link[i], nl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (any) s_active = 1;
        __syncthreads();
        const bool go = s_active != 0;
        __syncthreads();
        if (!go) break;
    }
}

// Tail edges publish their chain's length (in edges) at the head.
__global__ void __launch_bounds__(GASM_WG) k_chain_len(const u8* __restrict__ eflag, const u32* __restrict__ nxt,
                                                       const u64* __restrict__ link, u32* __restrict__ clen, u32 n_edges) {
    const u32 i = blockIdx.x * GASM_WG + threadIdx.x;
    if (i >= n_edges) return;
    if (nxt[i] != GASM_NONE32) return;
    const u64 l = link[i];
    const u32 a = (u32)(l >> 32);
    if (a == GASM_NONE32 || !(l & GASM_LINK_DONE)) return;  // isolated cycle member: no contig starts there
    clen[a] = ((u32)l & 0x7FFFFFFFu) + 1;
    (void)eflag;
}

// Per segment: rank of every head among the segment's heads and the base offset of its contig inside the segment
// (contig length = k-1 + chain length).  One workgroup of 1024 threads per segment.
__global__ void __launch_bounds__(1024) k_contig_scan(GraphView gv, const u8* __restrict__ eflag, const u32* __restrict__ clen,
                                                      u32* __restrict__ e_cid, u64* __restrict__ e_coff,
                                                      u32* __restrict__ seg_ncontig, u64* __restrict__ seg_cbases) {
    __shared__ u32 s_tmp[16];
    const u32 seg = blockIdx.x;
    const u32 nb = 1u << gv.bbits;
    const u32 lo = gv.dstart[seg * nb], hi = gv.dstart[(seg + 1) * nb];
    u32 ccarry = 0;
    u64 bcarry = 0;
    for (u32 base = lo; base < hi; base += 1024) {
        const u32 i = base + threadIdx.x;
        const bool head = i < hi && (eflag[i] & 1);
        const u32 len = head ? (u32)(gv.k - 1) + clen[i] : 0u;
        u32 ctot, btot;
        const u32 cex = block_excl_scan<1024>(head ? 1u : 0u, s_tmp, &ctot);
        const u32 bex = block_excl_scan<1024>(len, s_tmp, &btot);  // < 2^32 per 1024 edges for any sane contig
        if (head) { e_cid[i] = ccarry + cex; e_coff[i] = bcarry + bex; }
        ccarry += ctot;
        bcarry += btot;
    }
    if (threadIdx.x == 0) { seg_ncontig[seg] = ccarry; seg_cbases[seg] = bcarry; }
}

// Heads: make contig ids and offsets global; record offset and length per contig.
__global__ void __launch_bounds__(GASM_WG) k_contig_place(GraphView gv, const u8* __restrict__ eflag, const u32* __restrict__ clen,
                                                          const u32* __restrict__ seg_cstart, const u64* __restrict__ seg_bstart,
                                                          u32* __restrict__ e_cid, u64* __restrict__ e_coff,
                                                          u64* __restrict__ c_off) {
    const u32 seg = blockIdx.y;
    const u32 nb = 1u << gv.bbits;
    const u32 lo = gv.dstart[seg * nb], hi = gv.dstart[(seg + 1) * nb];
    const u32 i = lo + blockIdx.x * GASM_WG + threadIdx.x;
    if (i >= hi || !(eflag[i] & 1)) return;
    const u32 cid = seg_cstart[seg] + e_cid[i];
    const u64 off = seg_bstart[seg] + e_coff[i];
    e_cid[i] = cid;
    e_coff[i] = off;
    c_off[cid] = off;
    (void)clen;
}

// Every edge on a chain writes its last base at head offset + (k-1) + distance; the head also writes its node.
template <class K>
__global__ void __launch_bounds__(GASM_WG) k_contig_emit(GraphView gv, const u8* __restrict__ eflag, const u64* __restrict__ link,
                                                         const u64* __restrict__ e_coff, u8* __restrict__ out, u32 n_edges) {
    const u32 i = blockIdx.x * GASM_WG + threadIdx.x;
    if (i >= n_edges) return;
    const u64 l = link[i];
    const u32 a = (u32)(l >> 32);
    if (a == GASM_NONE32 || !(l & GASM_LINK_DONE)) return;
    (void)eflag;
    const K key = reinterpret_cast<const K*>(gv.dk_key)[i];
    const u64 off = e_coff[a];
    const int k = gv.k;
    out[off + (k - 1) + ((u32)l & 0x7FFFFFFFu)] = "ACGT"[klow2(key)];
    if (a == i) {
        for (int j = 0; j < k - 1; ++j) out[off + j] = "ACGT"[klow2(kshr(key, 2 * (k - 1 - j)))];
    }
}
template __global__ void k_contig_emit<u64>(GraphView, const u8*, const u64*, const u64*, u8*, u32);
template __global__ void k_contig_emit<K128>(GraphView, const u8*, const u64*, const u64*, u8*, u32);
