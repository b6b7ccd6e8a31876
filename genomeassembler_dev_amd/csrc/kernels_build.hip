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
#include "kernels.h"

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
    if (!ok) atomicOr(err, 1u);
}

// ================================================================================================================
// Tiles.  A tile is up to `ipt` (= GASM_WG / g) consecutive reads of one segment; g threads share a read and take
// k-mer start offsets lane, lane+g, ...  Segment of a tile: binary search in seg_tile_start (S+1 entries).
// ================================================================================================================
struct TileInfo { u32 seg; u64 r0; u32 nitems; };

__device__ __forceinline__ TileInfo tile_decode(const ReadSet& rs, u32 tile, u32 ipt) {
    TileInfo ti;
    ti.seg = upper_seg<u32>(rs.seg_tile_start, rs.n_segments, tile);
    const u64 first = rs.seg_read_off[ti.seg] + (u64)(tile - rs.seg_tile_start[ti.seg]) * ipt;
    const u64 left = rs.seg_read_off[ti.seg + 1] - first;
    ti.r0 = first;
    ti.nitems = (u32)(left < ipt ? left : ipt);
    return ti;
}

__device__ __forceinline__ void read_span(const ReadSet& rs, u64 r, u64* p0, u32* len) {
    if (rs.fixed_len) { *p0 = r * rs.fixed_len; *len = rs.fixed_len; }
    else { const u64 a = rs.read_off[r]; *p0 = a; *len = (u32)(rs.read_off[r + 1] - a); }
}

// Per-(segment,bucket) k-mer histogram.  The bucket of a k-mer is its first `bbits` bits (bbits <= 2k, bbits <= 10),
// i.e. buckets are key ranges: concatenating sorted buckets gives a sorted segment.
__global__ void __launch_bounds__(GASM_WG) k_bucket_hist(ReadSet rs, int k, int bbits, u32 g, u32 n_tiles,
                                                         u32* __restrict__ hist) {
    extern __shared__ u32 s_h[];
    const u32 nb = 1u << bbits, ipt = GASM_WG / g;
    const u32 item = threadIdx.x / g, lane = threadIdx.x % g;
    for (u32 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        for (u32 b = threadIdx.x; b < nb; b += GASM_WG) s_h[b] = 0;
        __syncthreads();
        const TileInfo ti = tile_decode(rs, tile, ipt);
        if (item < ti.nitems) {
            u64 p0; u32 len;
            read_span(rs, ti.r0 + item, &p0, &len);
            const u32 nk = len >= (u32)k ? len - k + 1 : 0;
            for (u32 off = lane; off < nk; off += g) {
                const u32 bkt = bbits ? (u32)(window32(rs.words, p0 + off) >> (64 - bbits)) : 0u;
                atomicAdd(&s_h[bkt], 1u);
            }
        }
        __syncthreads();
        for (u32 b = threadIdx.x; b < nb; b += GASM_WG) {
            const u32 c = s_h[b];
            if (c) atomicAdd(&hist[(u64)ti.seg * nb + b], c);
        }
        __syncthreads();
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
// Keys are binned in LDS first (rank by ds_add_rtn, one global atomic per workgroup and bucket to reserve the
// output range), so the global stores of one bucket are consecutive.
// LDS: KT*256 keys (8 B) + bucket ids (2 B) + 3 bucket arrays.
// ================================================================================================================

__global__ void __launch_bounds__(GASM_WG) k_bucket_scatter(ReadSet rs, int k, int bbits, u32 g, u32 n_tiles,
                                                            u64* __restrict__ cursor, u64* __restrict__ keys) {
    extern __shared__ __align__(16) unsigned char s_raw[];
    const u32 nb = 1u << bbits, ipt = GASM_WG / g;
    u64* s_key = reinterpret_cast<u64*>(s_raw);                        // GASM_KT*GASM_WG
    u64* s_gbase = s_key + GASM_KT * GASM_WG;                          // nb
    u32* s_cnt = reinterpret_cast<u32*>(s_gbase + nb);                 // nb
    u32* s_off = s_cnt + nb;                                           // nb
    u32* s_tmp = s_off + nb;                                           // 8 (+ max nk at [6])
    u16* s_bkt = reinterpret_cast<u16*>(s_tmp + 8);                    // GASM_KT*GASM_WG
    const u32 item = threadIdx.x / g, lane = threadIdx.x % g;
    const int kshift = 64 - 2 * k;

    for (u32 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const TileInfo ti = tile_decode(rs, tile, ipt);
        u64 p0 = 0; u32 nk = 0;
        if (item < ti.nitems) {
            u32 len;
            read_span(rs, ti.r0 + item, &p0, &len);
            nk = len >= (u32)k ? len - k + 1 : 0;
        }
        if (threadIdx.x == 0) s_tmp[6] = 0;
        for (u32 b = threadIdx.x; b < nb; b += GASM_WG) s_cnt[b] = 0;
        __syncthreads();
        if (lane == 0 && nk) atomicMax(&s_tmp[6], nk);
        __syncthreads();
        const u32 max_nk = s_tmp[6];
        const u32 per_round = g * GASM_KT;
        for (u32 r0 = 0; r0 < max_nk; r0 += per_round) {
            u64 key[GASM_KT];
            u32 meta[GASM_KT];  // bucket << 16 | rank   (rank < 4096)
#pragma unroll
            for (int j = 0; j < GASM_KT; ++j) {
                const u32 off = r0 + j * g + lane;
                meta[j] = GASM_NONE32;
                if (off < nk) {
                    const u64 wdw = window32(rs.words, p0 + off);
                    key[j] = wdw >> kshift;
                    const u32 bkt = bbits ? (u32)(wdw >> (64 - bbits)) : 0u;
                    meta[j] = (bkt << 16) | atomicAdd(&s_cnt[bkt], 1u);
                }
            }
            __syncthreads();
            // exclusive scan of the bucket counts; reserve the global ranges
            {
                const u32 per = nb > GASM_WG ? nb / GASM_WG : 1;
                const u32 base = threadIdx.x * per;
                u32 sum = 0;
                for (u32 q = 0; q < per; ++q) if (base + q < nb) sum += s_cnt[base + q];
                u32 tot;
                u32 ex = block_excl_scan<GASM_WG>(sum, s_tmp, &tot);
                for (u32 q = 0; q < per; ++q) {
                    if (base + q < nb) {
                        const u32 c = s_cnt[base + q];
                        s_off[base + q] = ex;
                        if (c) s_gbase[base + q] = atomicAdd(reinterpret_cast<unsigned long long*>(&cursor[(u64)ti.seg * nb + base + q]),
                                                             (unsigned long long)c);
                        ex += c;
                    }
                }
                if (threadIdx.x == 0) s_tmp[7] = tot;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < GASM_KT; ++j) {
                if (meta[j] != GASM_NONE32) {
                    const u32 bkt = meta[j] >> 16;
                    const u32 idx = s_off[bkt] + (meta[j] & 0xFFFFu);
                    s_key[idx] = key[j];
                    s_bkt[idx] = (u16)bkt;
                }
            }
            __syncthreads();
            const u32 total = s_tmp[7];
            for (u32 i = threadIdx.x; i < total; i += GASM_WG) {
                const u32 bkt = s_bkt[i];
                keys[s_gbase[bkt] + (i - s_off[bkt])] = s_key[i];
            }
            __syncthreads();
            for (u32 b = threadIdx.x; b < nb; b += GASM_WG) s_cnt[b] = 0;
            __syncthreads();
        }
    }
}

// ================================================================================================================
// De-duplicate one bucket: stream its keys through an LDS open-addressing table (64-bit CAS on the key, 32-bit add on
// the multiplicity), compact the table in place, bitonic-sort it, and write the sorted distinct keys and their
// multiplicities back over the start of the bucket's own range.  A bucket with more than GASM_TBL_LIMIT distinct
// keys raises *overflow (the host then re-partitions with more bucket bits).
// ================================================================================================================

__global__ void __launch_bounds__(GASM_WG) k_bucket_dedup(u64* __restrict__ keys, u32* __restrict__ mult,
                                                          const u64* __restrict__ bstart, u32* __restrict__ bucket_d,
                                                          u32* __restrict__ overflow) {
    __shared__ u64 t_key[GASM_TBL];
    __shared__ u32 t_cnt[GASM_TBL];
    __shared__ u32 s_tmp[8];
    const u32 bucket = blockIdx.x;
    const u64 beg = bstart[bucket], end = bstart[bucket + 1];
    const u64 n = end - beg;
    for (u32 i = threadIdx.x; i < GASM_TBL; i += GASM_WG) { t_key[i] = GASM_EMPTY64; t_cnt[i] = 0; }
    if (threadIdx.x == 0) { s_tmp[4] = 0; s_tmp[5] = 0; }  // [4] distinct so far, [5] overflow
    __syncthreads();
    for (u64 i = threadIdx.x; i < n; i += GASM_WG) {
        const u64 key = keys[beg + i];
        if (*reinterpret_cast<volatile u32*>(&s_tmp[4]) > GASM_TBL_LIMIT) { s_tmp[5] = 1; break; }
        u32 h = hash64(key) >> (32 - 12);
        while (true) {
            const u64 cur = *reinterpret_cast<volatile u64*>(&t_key[h]);
            if (cur == key) { atomicAdd(&t_cnt[h], 1u); break; }
            if (cur == GASM_EMPTY64) {
                const u64 old = atomicCAS(reinterpret_cast<unsigned long long*>(&t_key[h]), (unsigned long long)GASM_EMPTY64,
                                          (unsigned long long)key);
                if (old == GASM_EMPTY64) { atomicAdd(&s_tmp[4], 1u); atomicAdd(&t_cnt[h], 1u); break; }
                if (old == key) { atomicAdd(&t_cnt[h], 1u); break; }
            }
            h = (h + 1) & (GASM_TBL - 1);
        }
    }
    __syncthreads();
    if (s_tmp[5]) {
        if (threadIdx.x == 0) { atomicExch(overflow, 1u); bucket_d[bucket] = 0; }
        return;
    }
    // ---- compact in place: every thread pulls its 16 slots (stride 256: conflict-free) into registers
    u64 rk[GASM_TBL / GASM_WG];
    u32 rc[GASM_TBL / GASM_WG];
    u32 mine = 0;
#pragma unroll
    for (int q = 0; q < GASM_TBL / GASM_WG; ++q) {
        rk[q] = t_key[q * GASM_WG + threadIdx.x];
        rc[q] = t_cnt[q * GASM_WG + threadIdx.x];
        mine += rk[q] != GASM_EMPTY64;
    }
    u32 d;
    u32 pos = block_excl_scan<GASM_WG>(mine, s_tmp, &d);  // barriers inside: all reads are done before any write
#pragma unroll
    for (int q = 0; q < GASM_TBL / GASM_WG; ++q) {
        if (rk[q] != GASM_EMPTY64) { t_key[pos] = rk[q]; t_cnt[pos] = rc[q]; ++pos; }
    }
    u32 p2 = 2;
    while (p2 < d) p2 <<= 1;
    __syncthreads();
    for (u32 i = d + threadIdx.x; i < p2; i += GASM_WG) { t_key[i] = GASM_EMPTY64; t_cnt[i] = 0; }
    __syncthreads();
    // ---- bitonic sort of p2 (key, multiplicity) pairs, ascending
    for (u32 kk = 2; kk <= p2; kk <<= 1) {
        for (u32 j = kk >> 1; j > 0; j >>= 1) {
            for (u32 t = threadIdx.x; t < (p2 >> 1); t += GASM_WG) {
                const u32 lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const u32 hi = lo | j;
                const u64 a = t_key[lo], b = t_key[hi];
                const bool up = (lo & kk) == 0;
                if ((a > b) == up) {
                    t_key[lo] = b; t_key[hi] = a;
                    const u32 ca = t_cnt[lo], cb = t_cnt[hi];
                    t_cnt[lo] = cb; t_cnt[hi] = ca;
                }
            }
            __syncthreads();
        }
    }
    for (u32 i = threadIdx.x; i < d; i += GASM_WG) { keys[beg + i] = t_key[i]; mult[beg + i] = t_cnt[i]; }
    if (threadIdx.x == 0) bucket_d[bucket] = d;
}

// Gather the per-bucket distinct runs into the dense per-segment arrays.
__global__ void __launch_bounds__(GASM_WG) k_bucket_gather(const u64* __restrict__ keys, const u32* __restrict__ mult,
                                                           const u64* __restrict__ bstart, const u32* __restrict__ dstart,
                                                           u64* __restrict__ dk_key, u32* __restrict__ dk_cnt) {
    const u32 bucket = blockIdx.x;
    const u64 src = bstart[bucket];
    const u32 dst = dstart[bucket], d = dstart[bucket + 1] - dst;
    for (u32 i = threadIdx.x; i < d; i += GASM_WG) { dk_key[dst + i] = keys[src + i]; dk_cnt[dst + i] = mult[src + i]; }
}

// ================================================================================================================
// Graph over the sorted distinct k-mers (= distinct edges) of each segment.  Edge i: key = x·M·y, source node
// u = key>>2 (x·M), target node v = key & mask (M·y).  blockIdx.y = segment.
// ================================================================================================================
__device__ __forceinline__ bool kmer_exists(const GraphView& gv, u32 seg, u64 t) {
    const u32 nb = 1u << gv.bbits;
    const u32 bkt = gv.bbits ? (u32)(t >> (2 * gv.k - gv.bbits)) : 0u;
    const u32 lo = gv.dstart[seg * nb + bkt], hi = gv.dstart[seg * nb + bkt + 1];
    const u32 j = lower_bound_dev<u64>(gv.dk_key, lo, hi, t);
    return j < hi && gv.dk_key[j] == t;
}

// flag bit0: the edge's source node is a branching node (in != 1 or out != 1); it has out-edges by construction.
__global__ void __launch_bounds__(GASM_WG) k_node_flags(GraphView gv, u8* __restrict__ eflag) {
    const u32 seg = blockIdx.y;
    const u32 nb = 1u << gv.bbits;
    const u32 lo = gv.dstart[seg * nb], hi = gv.dstart[(seg + 1) * nb];
    const u32 i = lo + blockIdx.x * GASM_WG + threadIdx.x;
    if (i >= hi) return;
    const u64 key = gv.dk_key[i];
    const u64 u = key >> 2;
    // out-degree of u: the run of keys sharing key>>2 is contiguous in the sorted list
    u32 outd = 1;
    for (u32 j = i; j > lo && (gv.dk_key[j - 1] >> 2) == u; --j) ++outd;
    for (u32 j = i + 1; j < hi && (gv.dk_key[j] >> 2) == u; ++j) ++outd;
    // in-degree of u: distinct k-mers x·u
    u32 ind = 0;
    const int sh = 2 * (gv.k - 1);
#pragma unroll
    for (u64 x = 0; x < 4; ++x) ind += kmer_exists(gv, seg, (x << sh) | u);
    eflag[i] = (ind != 1 || outd != 1) ? 1 : 0;
}

// Successor edge of every edge (GASM_NONE32 when the walk stops at its target), and the initial ancestor links:
// link = ancestor << 32 | distance.  Heads are their own ancestor at distance 0.
__global__ void __launch_bounds__(GASM_WG) k_edge_next(GraphView gv, const u8* __restrict__ eflag, u32* __restrict__ nxt,
                                                       u64* __restrict__ link) {
    const u32 seg = blockIdx.y;
    const u32 nb = 1u << gv.bbits;
    const u32 lo = gv.dstart[seg * nb], hi = gv.dstart[(seg + 1) * nb];
    const u32 i = lo + blockIdx.x * GASM_WG + threadIdx.x;
    if (i >= hi) return;
    const u64 key = gv.dk_key[i];
    const int sh = 2 * (gv.k - 1);
    const u64 v = sh ? (key & ((1ull << sh) - 1)) : 0ull;
    const u64 t = v << 2;  // smallest k-mer with prefix v
    const u32 bkt = gv.bbits ? (u32)(t >> (2 * gv.k - gv.bbits)) : 0u;
    const u32 blo = gv.dstart[seg * nb + bkt];
    // the run of v may continue into the next bucket only if bbits > 2(k-1), which the host never chooses
    const u32 bhi = gv.dstart[seg * nb + bkt + 1];
    const u32 j = lower_bound_dev<u64>(gv.dk_key, blo, bhi, t);
    u32 n = GASM_NONE32;
    if (j < bhi && (gv.dk_key[j] >> 2) == v && !(eflag[j] & 1)) n = j;  // v has out-edges and is not branching
    nxt[i] = n;
    if (n != GASM_NONE32) link[n] = ((u64)i << 32) | 1ull;
    if (eflag[i] & 1) link[i] = (u64)i << 32;
}

// One round of pointer doubling towards the head of the chain.  In place and asynchronous: a link is always a
// consistent (ancestor, distance) pair because it is read and written as one 64-bit word.
__global__ void __launch_bounds__(GASM_WG) k_link_jump(const u8* __restrict__ eflag, u64* __restrict__ link, u32 n_edges) {
    const u32 i = blockIdx.x * GASM_WG + threadIdx.x;
    if (i >= n_edges) return;
    const u64 l = link[i];
    const u32 a = (u32)(l >> 32);
    if (a == GASM_NONE32 || (eflag[a] & 1)) return;
    const u64 la = *reinterpret_cast<volatile const u64*>(&link[a]);
    if ((u32)(la >> 32) == GASM_NONE32) return;
    link[i] = (la & 0xFFFFFFFF00000000ull) | (u64)((u32)l + (u32)la);
}

// Tail edges publish their chain's length (in edges) at the head.
__global__ void __launch_bounds__(GASM_WG) k_chain_len(const u8* __restrict__ eflag, const u32* __restrict__ nxt,
                                                       const u64* __restrict__ link, u32* __restrict__ clen, u32 n_edges) {
    const u32 i = blockIdx.x * GASM_WG + threadIdx.x;
    if (i >= n_edges) return;
    if (nxt[i] != GASM_NONE32) return;
    const u64 l = link[i];
    const u32 a = (u32)(l >> 32);
    if (a == GASM_NONE32 || !(eflag[a] & 1)) return;  // isolated cycle member: no contig starts there
    clen[a] = (u32)l + 1;
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
__global__ void __launch_bounds__(GASM_WG) k_contig_emit(GraphView gv, const u8* __restrict__ eflag, const u64* __restrict__ link,
                                                         const u64* __restrict__ e_coff, u8* __restrict__ out, u32 n_edges) {
    const u32 i = blockIdx.x * GASM_WG + threadIdx.x;
    if (i >= n_edges) return;
    const u64 l = link[i];
    const u32 a = (u32)(l >> 32);
    if (a == GASM_NONE32 || !(eflag[a] & 1)) return;
    const u64 key = gv.dk_key[i];
    const u64 off = e_coff[a];
    const int k = gv.k;
    out[off + (k - 1) + (u32)l] = "ACGT"[key & 3];
    if (a == i) {
        for (int j = 0; j < k - 1; ++j) out[off + j] = "ACGT"[(key >> (2 * (k - 1 - j))) & 3];
    }
}
