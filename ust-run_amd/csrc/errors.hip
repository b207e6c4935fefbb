#include "common.h"
#include <stdarg.h>
namespace ustrun {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
const char* get_error() { return g_err; }
}
// ---- dynamic LDS above the 64 KB default: the attribute is per kernel AND per device.  The largest size asked for so far is
// kept per (kernel, device) and the attribute is only ever RAISED (a later, smaller request must not lower the limit a
// larger launch of the same kernel still needs: ADVICE r3); a failed set is reported here instead of as an opaque launch
// error later ----
#include <map>
#include <mutex>
#include <utility>
namespace ustrun {
int ensure_dynamic_lds(const void* fn, int bytes, const char* who) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, int> granted;      // (kernel, device) -> bytes the attribute stands at
    if (bytes <= 64 * 1024) return 0;                                 // the default limit covers it
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    USTRUN_CHECK(e == hipSuccess, "%s: hipGetDevice failed: %s", who, hipGetErrorString(e));
    const std::pair<const void*, int> key(fn, dev);
    std::lock_guard<std::mutex> lk(mu);
    auto it = granted.find(key);
    if (it != granted.end() && it->second >= bytes) return 0;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    USTRUN_CHECK(e == hipSuccess, "%s: cannot raise the dynamic LDS limit to %d bytes on device %d: %s", who, bytes, dev, hipGetErrorString(e));
    granted[key] = bytes;
    return 0;
}
}
extern "C" int ustrun_version(void) { return USTRUN_VERSION; }
extern "C" const char* ustrun_last_error(void) { return ustrun::get_error(); }

// ---- launch profiler -----------------------------------------------------------------------
#include <vector>
namespace ustrun {
namespace {
struct Slot { hipEvent_t a, b; double flops, bytes; int kind, tag, n; };
bool g_prof_on = false;
bool g_prof_any_stream = true;          // false: only launches on g_prof_stream are timed
hipStream_t g_prof_stream = nullptr;
bool g_open = false;                    // the last prof_begin recorded (its prof_end must too)
int g_tag = -1, g_tag_n = 0;
std::vector<Slot> g_slots;
std::vector<hipEvent_t> g_pool;
hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    // no system-scope fence at the record: the default event flushes caches and costs ~10 us per record
    hipEvent_t e; (void)hipEventCreateWithFlags(&e, hipEventDisableSystemFence); return e;
}
}  // namespace
void prof_set_tag(int tag, int n) { g_tag = tag; g_tag_n = n; }
void prof_begin(int kind, double flops, double bytes, hipStream_t st) {
    g_open = false;
    // a launch on another stream (the batch-1 forward on its side stream) overlaps the profiled stream's kernels: its
    // event pair would measure contention, not the kernel
    if (!g_prof_on || (!g_prof_any_stream && st != g_prof_stream)) return;
    Slot s; s.a = get_event(); s.b = get_event(); s.flops = flops; s.bytes = bytes; s.kind = kind; s.tag = g_tag; s.n = g_tag_n;
    (void)hipEventRecord(s.a, st);
    g_slots.push_back(s);
    g_open = true;
}
void prof_end(hipStream_t st) {
    if (!g_open || g_slots.empty()) return;
    (void)hipEventRecord(g_slots.back().b, st);
    g_open = false;
}
}  // namespace ustrun

extern "C" int ustrun_profile_enable(int on) { ustrun::g_prof_on = on != 0; return 0; }

extern "C" int ustrun_profile_stream(ustrun_stream_t s, int only) {
    ustrun::g_prof_stream = (hipStream_t)s; ustrun::g_prof_any_stream = only == 0;
    return 0;
}

extern "C" int64_t ustrun_profile_records(ustrun_prof_rec_t* out, int64_t max) {
    using namespace ustrun;
    int64_t n = 0;
    for (auto& s : g_slots) {
        if (n >= max) break;
        if (hipEventSynchronize(s.b) != hipSuccess) { set_error("profile_records: event sync failed"); return -1; }
        float e = 0.f; (void)hipEventElapsedTime(&e, s.a, s.b);
        out[n].kind = s.kind; out[n].tag = s.tag; out[n].n = s.n; out[n].pad = 0;
        out[n].ms = e; out[n].flops = s.flops; out[n].bytes = s.bytes;
        ++n;
    }
    return n;
}

extern "C" int ustrun_profile_collect(int kind, double* ms, double* flops, double* bytes, int64_t* launches) {
    using namespace ustrun;
    double t = 0, f = 0, b = 0; int64_t n = 0;
    std::vector<Slot> keep;
    for (auto& s : g_slots) {
        if (s.kind != kind) { keep.push_back(s); continue; }
        if (hipEventSynchronize(s.b) != hipSuccess) { set_error("profile_collect: event sync failed"); return 2; }
        float e = 0.f; (void)hipEventElapsedTime(&e, s.a, s.b);
        t += e; f += s.flops; b += s.bytes; ++n;
        g_pool.push_back(s.a); g_pool.push_back(s.b);
    }
    g_slots.swap(keep);
    if (ms) *ms = t; if (flops) *flops = f; if (bytes) *bytes = b; if (launches) *launches = n;
    return 0;
}
