// ctx.cpp — context, error text, per-kernel HIP-event profiling.
#include "gasm_internal.h"

#include <dlfcn.h>

static thread_local std::string g_err;

void gasm_set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

// roctx ranges around the stages (the reference brackets its phases with Sys.time()): rocprofv3 --marker-trace shows
// them.  libroctx64 is resolved at first use; without it (or with GASM_ROCTX=0) the calls are no-ops.
namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char* off = getenv("GASM_ROCTX");
        if (off && *off == '0') return;
        for (const char* n : {"libroctx64.so.4", "libroctx64.so", "/opt/rocm/lib/libroctx64.so.4"}) {
            if (void* h = dlopen(n, RTLD_NOW | RTLD_LOCAL)) {
                push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
                pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (push && pop) return;
                push = nullptr; pop = nullptr;
            }
        }
    }
};
const Roctx& roctx() { static const Roctx r; return r; }
}  // namespace
void gasm_range_push(const char* name) { if (roctx().push) roctx().push(name); }
void gasm_range_pop() { if (roctx().pop) roctx().pop(); }

extern "C" const char* gasm_last_error(void) { return g_err.c_str(); }
extern "C" const char* gasm_version(void) { return "libgasm 0.1 (gfx950)"; }

hipEvent_t gasm_ctx::ev_get() {
    if (!ev_pool.empty()) {
        hipEvent_t e = ev_pool.back();
        ev_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

void gasm_ctx::prof_begin(const char* name, hipEvent_t* a, hipEvent_t* b, int* stage) {
    auto it = stage_ix.find(name);
    if (it == stage_ix.end()) {
        ProfStage s;
        s.name = name;
        stages.push_back(s);
        it = stage_ix.emplace(name, (int)stages.size() - 1).first;
    }
    *stage = it->second;
    *a = ev_get();
    *b = ev_get();
    (void)hipEventRecord(*a, stream);
}

void gasm_ctx::prof_end(int stage, hipEvent_t a, hipEvent_t b) {
    (void)hipEventRecord(b, stream);
    pending.push_back({stage, a, b});
}

// GASM_PROF_TIMELINE=<file> (diagnostic): every profiled launch as "stream name start_ms end_ms" relative to the first event
// this process collected — the steps' real timeline across the streams of the step slots, without a tracer that spaces the
// dispatches out (tools/slot_timeline.py)
static hipEvent_t g_tl_base = nullptr;
static FILE* timeline_file() {
    static FILE* f = [] { const char* v = getenv("GASM_PROF_TIMELINE"); return v && *v ? fopen(v, "w") : (FILE*)nullptr; }();
    return f;
}

int gasm_ctx::prof_collect() {
    HIPCHK(hipStreamSynchronize(stream));
    FILE* const tl = pending.empty() ? nullptr : timeline_file();
    if (tl && !g_tl_base) g_tl_base = pending.front().a;          // (kept: never returned to the pool)
    for (auto& p : pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            stages[p.stage].ms += ms;
            stages[p.stage].launches += 1;
        }
        if (tl) {
            float t0 = 0, t1 = 0;
            if (hipEventElapsedTime(&t0, g_tl_base, p.a) == hipSuccess && hipEventElapsedTime(&t1, g_tl_base, p.b) == hipSuccess)
                fprintf(tl, "%p %s %.4f %.4f\n", (void*)stream, stages[p.stage].name.c_str(), t0, t1);
        }
        if (p.a == g_tl_base) { ev_pool.push_back(p.b); continue; }
        ev_pool.push_back(p.a);
        ev_pool.push_back(p.b);
    }
    if (tl) fflush(tl);
    pending.clear();
    return GASM_OK;
}

static gasm_ctx* ctx_new(int device, int n_cu, bool high_priority = false) {
    gasm_ctx* c = new gasm_ctx();
    c->device = device;
    c->n_cu = n_cu;
    int lo = 0, hi = 0;                    // (numerically lower = served first)
    if (high_priority && hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) lo = hi = 0;
    if ((high_priority && hi != lo ? hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi)
                                   : hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
        delete c;
        gasm_set_error("hipStreamCreate failed");
        return nullptr;
    }
    c->h_pin_words = 1 << 17;      // 1 MB: staging of small copies
    if (hipHostMalloc((void**)&c->h_pin, c->h_pin_words * sizeof(u64), hipHostMallocCoherent) != hipSuccess) {
        (void)hipStreamDestroy(c->stream);
        delete c;
        gasm_set_error("hipHostMalloc failed");
        return nullptr;
    }
    return c;
}

gasm_ctx* gasm_ctx::lane(size_t i) {
    while (lanes.size() <= i) {
        if (hipSetDevice(device) != hipSuccess) return nullptr;
        gasm_ctx* l = ctx_new(device, n_cu);
        if (!l) return nullptr;
        l->prof = prof;
        l->prof_only = prof_only;
        lanes.push_back(l);
    }
    return lanes[i];
}

gasm_ctx* gasm_ctx::tail_lane(size_t i) {
    while (tails.size() <= i) {
        if (hipSetDevice(device) != hipSuccess) return nullptr;
        gasm_ctx* l = ctx_new(device, n_cu, true);
        if (!l) return nullptr;
        l->prof = prof;
        l->prof_only = prof_only;
        tails.push_back(l);
    }
    return tails[i];
}

std::vector<gasm_ctx*> gasm_ctx::all_lanes() const {
    std::vector<gasm_ctx*> v = lanes;
    v.insert(v.end(), tails.begin(), tails.end());
    return v;
}

extern "C" int gasm_ctx_create(int device, gasm_ctx** out) {
    if (!out) { gasm_set_error("gasm_ctx_create: out is null"); return GASM_ERR_INVALID; }
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        gasm_set_error("no HIP device available (%s); libgasm has no CPU fallback", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        return GASM_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) { gasm_set_error("device %d out of range (have %d)", device, n); return GASM_ERR_NO_DEVICE; }
    if (hipSetDevice(device) != hipSuccess) { gasm_set_error("hipSetDevice(%d) failed", device); return GASM_ERR_NO_DEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { gasm_set_error("hipGetDeviceProperties failed"); return GASM_ERR_NO_DEVICE; }
    if (!strstr(prop.gcnArchName, "gfx950")) {
        gasm_set_error("device %d is %s; libgasm is built for gfx950 only", device, prop.gcnArchName);
        return GASM_ERR_NO_DEVICE;
    }
    gasm_ctx* c = ctx_new(device, prop.multiProcessorCount);
    if (!c) return GASM_ERR_NO_DEVICE;
    *out = c;
    return GASM_OK;
}

extern "C" void gasm_ctx_destroy(gasm_ctx* c) {
    if (!c) return;
    for (gasm_ctx* l : c->all_lanes()) gasm_ctx_destroy(l);
    c->lanes.clear();
    c->tails.clear();
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& p : c->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int gasm_ctx_sync(gasm_ctx* c) {
    if (!c) { gasm_set_error("ctx is null"); return GASM_ERR_INVALID; }
    HIPCHK(hipStreamSynchronize(c->stream));
    for (gasm_ctx* l : c->all_lanes()) HIPCHK(hipStreamSynchronize(l->stream));
    return GASM_OK;
}

extern "C" void* gasm_ctx_stream(gasm_ctx* c) { return c ? (void*)c->stream : nullptr; }

extern "C" int gasm_profile_enable(gasm_ctx* c, int on) {
    if (!c) { gasm_set_error("ctx is null"); return GASM_ERR_INVALID; }
    if (!on && c->prof) GCHK(c->prof_collect());
    c->prof = on != 0;
    for (gasm_ctx* l : c->all_lanes()) GCHK(gasm_profile_enable(l, on));
    return GASM_OK;
}

extern "C" int gasm_profile_filter(gasm_ctx* c, const char* names) {
    if (!c) { gasm_set_error("ctx is null"); return GASM_ERR_INVALID; }
    c->prof_only.clear();
    if (!names) return GASM_OK;
    std::string cur;
    for (const char* p = names;; ++p) {
        if (*p == ',' || *p == 0) {
            if (!cur.empty()) c->prof_only.push_back(cur);
            cur.clear();
            if (*p == 0) break;
        } else cur.push_back(*p);
    }
    for (gasm_ctx* l : c->all_lanes()) l->prof_only = c->prof_only;
    return GASM_OK;
}

extern "C" int gasm_profile_reset(gasm_ctx* c) {
    if (!c) { gasm_set_error("ctx is null"); return GASM_ERR_INVALID; }
    GCHK(c->prof_collect());
    for (auto& s : c->stages) { s.ms = 0; s.launches = 0; }
    for (gasm_ctx* l : c->all_lanes()) GCHK(gasm_profile_reset(l));
    return GASM_OK;
}

extern "C" int gasm_profile_read(gasm_ctx* c, int* n, const char* const** names, const double** ms, const uint64_t** launches) {
    if (!c || !n || !names || !ms || !launches) { gasm_set_error("null argument"); return GASM_ERR_INVALID; }
    GCHK(c->prof_collect());
    // the lanes' stages are folded into this context's (same kernel names)
    std::vector<ProfStage> all = c->stages;
    for (gasm_ctx* l : c->all_lanes()) {
        GCHK(l->prof_collect());
        for (auto& ls : l->stages) {
            bool found = false;
            for (auto& s : all) if (s.name == ls.name) { s.ms += ls.ms; s.launches += ls.launches; found = true; break; }
            if (!found) all.push_back(ls);
        }
    }
    c->out_stage_copy = all;
    c->out_names.clear(); c->out_ms.clear(); c->out_launches.clear();
    for (auto& s : c->out_stage_copy) { c->out_names.push_back(s.name.c_str()); c->out_ms.push_back(s.ms); c->out_launches.push_back(s.launches); }
    *n = (int)c->out_stage_copy.size();
    *names = c->out_names.data();
    *ms = c->out_ms.data();
    *launches = c->out_launches.data();
    return GASM_OK;
}
