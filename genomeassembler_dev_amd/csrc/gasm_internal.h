// gasm_internal.h — shared host-side plumbing of libgasm (context, device buffers, launch + profiling helpers).
// Device code lives in the .hip files; nothing here is visible through include/gasm.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/gasm.h"

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint16_t u16;
typedef uint8_t u8;

void gasm_set_error(const char* fmt, ...);
// roctx range around a stage (ctx.cpp); GasmRange: the same as a scope
void gasm_range_push(const char* name);
void gasm_range_pop();
struct GasmRange {
    explicit GasmRange(const char* name) { gasm_range_push(name); }
    ~GasmRange() { gasm_range_pop(); }
};

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) {                                                                        \
            gasm_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return GASM_ERR_HIP;                                                                       \
        }                                                                                              \
    } while (0)

#define GCHK(expr)                \
    do {                          \
        int _s = (expr);          \
        if (_s != GASM_OK) return _s; \
    } while (0)

// Grow-only device allocation.  Steady-state steps of the same shape never call hipMalloc.
struct DBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return GASM_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 16 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            gasm_set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
            p = nullptr;
            return GASM_ERR_HIP;
        }
        cap = want;
        return GASM_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct ProfStage {
    std::string name;
    double ms = 0;
    u64 launches = 0;
};

struct gasm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int n_cu = 256;
    bool lds_attrs_set = false;           // the > 64 KB dynamic-LDS opt-ins of this device's kernels (pipeline.hip)
    // lanes: further streams of the same device on which a batch runs its sub-batches side by side (capi.hip).  A lane is
    // a gasm_ctx of its own (stream, pinned area, profiler) owned by this one; sync / profile calls on the owner cover them.
    std::vector<gasm_ctx*> lanes;
    gasm_ctx* lane(size_t i);             // created on first use; nullptr on failure
    // tail lanes: lanes whose stream is served first by the device's dispatcher — the many small kernels behind a build's
    // streaming pair, while the next build's streaming pair fills the chip from this context's own stream (capi.hip)
    std::vector<gasm_ctx*> tails;
    gasm_ctx* tail_lane(size_t i);
    std::vector<gasm_ctx*> all_lanes() const;
    // small pinned host area for read-backs of counters/flags
    u64* h_pin = nullptr;
    size_t h_pin_words = 0;
    // profiling
    bool prof = false;
    std::vector<std::string> prof_only;   // empty = every kernel
    bool prof_wants(const char* name) const {
        if (!prof) return false;
        if (prof_only.empty()) return true;
        for (auto& s : prof_only) if (s == name) return true;
        return false;
    }
    std::vector<ProfStage> stages;
    std::map<std::string, int> stage_ix;
    struct Pending { int stage; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> ev_pool;
    // read-out storage for gasm_profile_read
    std::vector<ProfStage> out_stage_copy;
    std::vector<const char*> out_names;
    std::vector<double> out_ms;
    std::vector<u64> out_launches;

    hipEvent_t ev_get();
    void prof_begin(const char* name, hipEvent_t* a, hipEvent_t* b, int* stage);
    void prof_end(int stage, hipEvent_t a, hipEvent_t b);
    int prof_collect();  // syncs and folds pending pairs into stages
};

// Launch helper: optional HIP-event bracket per launch (stage = kernel name), launch error check.
#define GLAUNCH(ctx, name, kern, grid, block, shmem, ...)                                    \
    do {                                                                                     \
        hipEvent_t _a = nullptr, _b = nullptr;                                               \
        int _st = -1;                                                                        \
        const bool _p = (ctx)->prof_wants(name);                                             \
        if (_p) (ctx)->prof_begin(name, &_a, &_b, &_st);                                     \
        hipLaunchKernelGGL(kern, grid, block, shmem, (ctx)->stream, __VA_ARGS__);            \
        if (_p) (ctx)->prof_end(_st, _a, _b);                                                \
        HIPCHK(hipGetLastError());                                                           \
    } while (0)

// environment knobs (diagnostics and the documented switches): an integer / a flag ("0" = off), with a default
static inline int env_int(const char* name, int dflt) { const char* v = getenv(name); return v && *v ? atoi(v) : dflt; }
static inline bool env_flag(const char* name, bool dflt) { const char* v = getenv(name); return v && *v ? *v != '0' : dflt; }

static inline u32 ceil_div_u64(u64 a, u64 b) { return (u32)((a + b - 1) / b); }
static inline u32 next_pow2_u32(u32 x) {
    u32 p = 1;
    while (p < x) p <<= 1;
    return p;
}

// ---- host algorithms (host_algos.cpp) ------------------------------------------------------------------------
namespace gasm_host {
// std::shuffle permutations exactly as lib/DeNovoAssembler.cpp:195-203 draws them (indices instead of strings).
void shuffle_perm(u64 n, int seed, u64 rows, std::vector<u32>& perm);
// greedy merge + ordering of lib/DeNovoAssembler.cpp:228-304; returns GASM_ERR_RANGE where substr would throw.
int assemble(const std::vector<std::string>& contigs, const u32* perm, u64 rows, u64 row_len, int k, std::vector<std::string>& out);
// the index-form merge alone: sorted distinct chain signatures (u32 contig, u32 overlap, u32 contig, ...); false = not applicable
bool assemble_signatures(const std::vector<std::string>& contigs, const u32* perm, u64 rows, u64 row_len, int k, std::vector<std::string>& sigs);
// Myers bit-parallel edit distance; infix = edlib HW mode, else NW.
int levenshtein(const char* q, u64 nq, const char* t, u64 nt, bool infix);
// reads of sequence files, packed 2-bit back to back (seqio.cpp)
struct PackedReads {
    std::vector<u64> words;        // 32 bases per word, first base most significant
    std::vector<u64> read_off;     // n_reads + 1 base offsets (first entry 0)
    u64 total_bases = 0;
    void append(const std::string& seq);
};
int read_sequence_file(const char* path, bool error_on_non_acgt, PackedReads& out, u64* n_kept, u64* n_dropped);
}  // namespace gasm_host
