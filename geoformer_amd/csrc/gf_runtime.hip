// Error reporting + ABI version for libgeoformer_hip.so
#include <stdarg.h>

#include "gf_common.h"

static thread_local char g_err[512] = "";

void gf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int gf_abi_version(void) { return 1; }
extern "C" const char* gf_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------------------------
// Optional per-kernel timing with HIP events on the launch stream (bench.py's `roofline` object).
// Off by default: when off, gf_prof_begin/end are a single predictable branch.
// ---------------------------------------------------------------------------------------------
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {
struct ProfSpan { hipEvent_t a, b; double work; };
std::mutex g_prof_mu;
bool g_prof_on = false;
std::map<std::string, std::vector<ProfSpan>> g_prof;
}   // namespace

bool gf_prof_enabled() { return g_prof_on; }

void* gf_prof_begin(const char* tag, hipStream_t st, double work) {
    if (!g_prof_on) return nullptr;
    ProfSpan sp;
    sp.work = work;
    if (hipEventCreate(&sp.a) != hipSuccess || hipEventCreate(&sp.b) != hipSuccess) return nullptr;
    (void)hipEventRecord(sp.a, st);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    auto& v = g_prof[tag];
    v.push_back(sp);
    return (void*)(uintptr_t)v.size();
}

void gf_prof_end(const char* tag, void* token, hipStream_t st) {
    if (!token) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    auto& v = g_prof[tag];
    (void)hipEventRecord(v[(uintptr_t)token - 1].b, st);
}

extern "C" void gf_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
}

// Synchronises on the recorded events, returns the summed milliseconds, the span count and the summed
// work units (bytes or flops declared by the launch site) of `tag`, and clears it.
// Returns 0 on success, GF_ERR_INVALID_ARGUMENT when the tag has no spans.
extern "C" int gf_profile_collect(const char* tag, double* total_ms, int* count, double* work) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    auto it = g_prof.find(tag);
    if (it == g_prof.end() || it->second.empty()) {
        if (total_ms) *total_ms = 0.0;
        if (count) *count = 0;
        if (work) *work = 0.0;
        return GF_ERR_INVALID_ARGUMENT;
    }
    double tot = 0.0, wk = 0.0;
    int n = 0;
    for (auto& sp : it->second) {
        float ms = 0.f;
        if (hipEventSynchronize(sp.b) == hipSuccess && hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
            tot += ms;
            wk += sp.work;
            ++n;
        }
        (void)hipEventDestroy(sp.a);
        (void)hipEventDestroy(sp.b);
    }
    it->second.clear();
    if (total_ms) *total_ms = tot;
    if (count) *count = n;
    if (work) *work = wk;
    return GF_OK;
}
