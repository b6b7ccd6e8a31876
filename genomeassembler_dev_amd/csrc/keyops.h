// keyops.h — k-mer keys as integers: u64 for k <= 31, K128 (two words, hi:lo) for 32 <= k <= 63.
// A key holds the k bases right-aligned in 2k bits, first base most significant, so integer order = string order.
// The kernels are templates over the key type and only use the overloaded operations below.
#pragma once
#include "device_utils.h"

struct __attribute__((aligned(16))) K128 {
    u64 hi, lo;
};

// ---- construction / constants
template <class K> __device__ __forceinline__ K key_empty();
template <> __device__ __forceinline__ u64 key_empty<u64>() { return GASM_EMPTY64; }
template <> __device__ __forceinline__ K128 key_empty<K128>() { return K128{GASM_EMPTY64, GASM_EMPTY64}; }
template <class K> __device__ __forceinline__ K key_from_u64(u64 v);
template <> __device__ __forceinline__ u64 key_from_u64<u64>(u64 v) { return v; }
template <> __device__ __forceinline__ K128 key_from_u64<K128>(u64 v) { return K128{0, v}; }

// filler keys of the bucketed key array: top bit set (no k-mer has it: 2k <= 62 resp. 126 bits) over the bucket prefix
template <class K> __device__ __forceinline__ K key_filler(u32 bkt, int bshift);
__device__ __forceinline__ bool kis_filler(u64 a) { return (a >> 63) != 0; }

// ---- comparisons
__device__ __forceinline__ bool keq(u64 a, u64 b) { return a == b; }
__device__ __forceinline__ bool keq(const K128& a, const K128& b) { return a.hi == b.hi && a.lo == b.lo; }
__device__ __forceinline__ bool kless(u64 a, u64 b) { return a < b; }
__device__ __forceinline__ bool kless(const K128& a, const K128& b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }
__device__ __forceinline__ bool kis_empty(u64 a) { return a == GASM_EMPTY64; }
__device__ __forceinline__ bool kis_empty(const K128& a) { return (a.hi & a.lo) == GASM_EMPTY64; }

__device__ __forceinline__ bool kis_filler(const K128& a) { return (a.hi >> 63) != 0; }

// ---- shifts (0 <= s < width) and bit operations
__device__ __forceinline__ u64 kshr(u64 a, int s) { return a >> s; }
__device__ __forceinline__ K128 kshr(const K128& a, int s) {
    if (s == 0) return a;
    if (s >= 64) return K128{0, a.hi >> (s - 64)};
    return K128{a.hi >> s, (a.lo >> s) | (a.hi << (64 - s))};
}
__device__ __forceinline__ u64 kshl(u64 a, int s) { return a << s; }
__device__ __forceinline__ K128 kshl(const K128& a, int s) {
    if (s == 0) return a;
    if (s >= 64) return K128{a.lo << (s - 64), 0};
    return K128{(a.hi << s) | (a.lo >> (64 - s)), a.lo << s};
}
__device__ __forceinline__ u64 kor(u64 a, u64 b) { return a | b; }
__device__ __forceinline__ K128 kor(const K128& a, const K128& b) { return K128{a.hi | b.hi, a.lo | b.lo}; }
// the low `bits` bits (0 <= bits <= width)
__device__ __forceinline__ u64 klowbits(u64 a, int bits) { return bits >= 64 ? a : (a & ((1ull << bits) - 1)); }
__device__ __forceinline__ K128 klowbits(const K128& a, int bits) {
    if (bits >= 128) return a;
    if (bits >= 64) return K128{bits == 64 ? 0 : (a.hi & ((1ull << (bits - 64)) - 1)), a.lo};
    return K128{0, a.lo & ((1ull << bits) - 1)};
}
template <> __device__ __forceinline__ u64 key_filler<u64>(u32 bkt, int bshift) { return (1ull << 63) | ((u64)bkt << bshift); }
template <> __device__ __forceinline__ K128 key_filler<K128>(u32 bkt, int bshift) {
    const K128 b = kshl(K128{0, (u64)bkt}, bshift);
    return K128{b.hi | (1ull << 63), b.lo};
}
// low 32 bits of (a >> s)
__device__ __forceinline__ u32 kfield(u64 a, int s) { return (u32)(a >> s); }
__device__ __forceinline__ u32 kfield(const K128& a, int s) { return (u32)kshr(a, s).lo; }
__device__ __forceinline__ u32 klow2(u64 a) { return (u32)a & 3u; }
__device__ __forceinline__ u32 klow2(const K128& a) { return (u32)a.lo & 3u; }

// Home-set hash of the de-duplication tables.  A 64-bit multiply is a handful of quarter-rate 32-bit multiplies on CDNA —
// a third of the vector time of k_bucket_dedup; two 24-bit multiplies (full rate) over a folded key spread real k-mer
// sets just as evenly (same overflow fraction on random keys, genome buckets and two-letter sequences).
__device__ __forceinline__ u32 khash(u64 a) {
    const u32 lo = (u32)a, hi = (u32)(a >> 32);
    u32 x = lo ^ __builtin_amdgcn_alignbit(hi, hi, 19);      // lo ^ rotl(hi, 13)
    x ^= x >> 17;
    return __umul24(x, 0xB5297Au) ^ __umul24(x >> 8, 0x68E31Du);
}
__device__ __forceinline__ u32 khash(const K128& a) { return hash64(a.lo ^ (a.hi * 0xD6E8FEB86659FD93ull)); }

// ---- windows of a packed base stream
// the k bases starting at base p as a key (k <= 31 for u64, k <= 63 for K128)
template <class K> __device__ __forceinline__ K kmer_key_at(const u64* __restrict__ w, u64 p, int k);
template <> __device__ __forceinline__ u64 kmer_key_at<u64>(const u64* __restrict__ w, u64 p, int k) { return window32(w, p) >> (64 - 2 * k); }
template <> __device__ __forceinline__ K128 kmer_key_at<K128>(const u64* __restrict__ w, u64 p, int k) {
    const K128 win{window32(w, p), window32(w, p + 32)};     // 64 bases from p
    return kshr(win, 128 - 2 * k);
}

// Rolling source: KT consecutive k-mer starts out of a few words held in registers (3 words cover 16 + 31 bases,
// 4 words cover 8 + 63 + 31 bases).  The raw words are what a prefetch holds; prep() aligns them once to the thread's
// first base (a run of 32-bit words d[n-1]..d[0]), after which the window j bases further on is a pair of v_alignbit
// with the constant shift 32 - 2j — two instructions per k-mer and no dependency between k-mers.
__device__ __forceinline__ u64 funnel64(u64 x, u64 y, u32 o) { return (x << o) | ((y >> 1) >> (63 - o)); }   // 0 <= o < 64
// bits [32 + 2j, 2j) ... i.e. the 32-bit word starting 2j bits into hi:lo (0 <= j <= 15)
template <u32 J> __device__ __forceinline__ u32 word_at(u32 hi, u32 lo) {
    if constexpr (J == 0) return hi;
    else return __builtin_amdgcn_alignbit(hi, lo, 32 - 2 * J);
}

template <class K> struct Roll;
template <> struct Roll<u64> {
    u64 w0, w1, w2;
    u32 s;
    __device__ __forceinline__ void load(const u64* __restrict__ w, u64 p) {
        const u64 i = p >> 5;
        w0 = w[i]; w1 = w[i + 1]; w2 = w[i + 2];
        s = (u32)(p & 31) << 1;
    }
    // the 96 bits (48 bases) from the load position
    struct Win {
        u32 d3, d2, d1;
        // the 32 bases starting J bases after the load position (J < 16): also the top of every k-mer there
        template <u32 J> __device__ __forceinline__ u64 top() const { return ((u64)word_at<J>(d3, d2) << 32) | word_at<J>(d2, d1); }
        template <u32 J> __device__ __forceinline__ u32 top_hi() const { return word_at<J>(d3, d2); }
        template <u32 J> __device__ __forceinline__ u64 key(int k) const { return top<J>() >> (64 - 2 * k); }
    };
    __device__ __forceinline__ Win prep() const {
        const u64 hi = funnel64(w0, w1, s), lo = funnel64(w1, w2, s);
        return Win{(u32)(hi >> 32), (u32)hi, (u32)(lo >> 32)};
    }
    // makes the compiler wait for the loaded words here
    __device__ __forceinline__ void touch() const { __asm__ volatile("" :: "v"(w0), "v"(w1), "v"(w2)); }
    // explicit wait for words requested N vector-memory operations before the most recent one (memory operations of a
    // wave retire in order); every later use of the words depends on this statement
    template <int N> __device__ __forceinline__ void wait_all_but() {
        __asm__ volatile("s_waitcnt vmcnt(%3)" : "+v"(w0), "+v"(w1), "+v"(w2) : "n"(N) : "memory");
    }
};
template <> struct Roll<K128> {
    u64 w0, w1, w2, w3;
    u32 s;
    __device__ __forceinline__ void load(const u64* __restrict__ w, u64 p) {
        const u64 i = p >> 5;
        w0 = w[i]; w1 = w[i + 1]; w2 = w[i + 2]; w3 = w[i + 3];
        s = (u32)(p & 31) << 1;
    }
    // the 160 bits (80 bases) from the load position: 8 starts + 63 bases of the longest k-mer
    struct Win {
        u32 d5, d4, d3, d2, d1;
        template <u32 J> __device__ __forceinline__ u32 top_hi() const { return word_at<J>(d5, d4); }
        template <u32 J> __device__ __forceinline__ K128 key(int k) const {
            const K128 win{((u64)word_at<J>(d5, d4) << 32) | word_at<J>(d4, d3), ((u64)word_at<J>(d3, d2) << 32) | word_at<J>(d2, d1)};
            return kshr(win, 128 - 2 * k);
        }
    };
    __device__ __forceinline__ Win prep() const {
        const u64 a = funnel64(w0, w1, s), b = funnel64(w1, w2, s), c = funnel64(w2, w3, s);
        return Win{(u32)(a >> 32), (u32)a, (u32)(b >> 32), (u32)b, (u32)(c >> 32)};
    }
    __device__ __forceinline__ void touch() const { __asm__ volatile("" :: "v"(w0), "v"(w1), "v"(w2), "v"(w3)); }
    template <int N> __device__ __forceinline__ void wait_all_but() {
        __asm__ volatile("s_waitcnt vmcnt(%4)" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "n"(N) : "memory");
    }
};

// words per key and k-mers per thread and round of the tile kernels (LDS staging is 8 KB per wave either way)
template <class K> struct KeyTraits;
// NFL: 16-byte store passes of k_bucket_scatter's flush = the tile + room for the padding of the staged runs (72 KB of LDS)
template <> struct KeyTraits<u64> { static constexpr int WORDS = 1; static constexpr int KT = 16; static constexpr int NFL = 9; };
template <> struct KeyTraits<K128> { static constexpr int WORDS = 2; static constexpr int KT = 8; static constexpr int NFL = 9; };
