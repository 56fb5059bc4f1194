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

extern "C" int gf_abi_version(void) { return GF_ABI_VERSION; }
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
std::string g_prof_filter;                              // empty = every tag
std::map<std::string, std::vector<ProfSpan>> g_prof;
uint32_t g_prof_gen = 1;                                // bumped by every collect: tokens of older spans go stale
std::vector<hipEvent_t> g_prof_pool;                    // recycled events (creating two per launch is not free)

hipEvent_t prof_event() {
    if (!g_prof_pool.empty()) {
        hipEvent_t e = g_prof_pool.back();
        g_prof_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    return hipEventCreate(&e) == hipSuccess ? e : nullptr;
}
}   // namespace

bool gf_prof_enabled() { return g_prof_on; }

void* gf_prof_begin(const char* tag, hipStream_t st, double work) {
    if (!g_prof_on) return nullptr;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_filter.empty() && g_prof_filter != tag) return nullptr;
    ProfSpan sp;
    sp.work = work;
    sp.a = prof_event();
    sp.b = prof_event();
    if (!sp.a || !sp.b) return nullptr;
    (void)hipEventRecord(sp.a, st);
    auto& v = g_prof[tag];
    v.push_back(sp);
    return (void*)(((uintptr_t)g_prof_gen << 32) | (uintptr_t)v.size());      // generation | 1-based index
}

void gf_prof_end(const char* tag, void* token, hipStream_t st) {
    if (!token) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    // a collect between begin and end (another host thread) emptied the vector: the token's generation no longer
    // matches and the span is dropped instead of indexing past the end
    const uintptr_t t = (uintptr_t)token, idx = t & 0xffffffffu;
    if ((uint32_t)(t >> 32) != g_prof_gen) return;
    auto& v = g_prof[tag];
    if (idx == 0 || idx > v.size()) return;
    (void)hipEventRecord(v[idx - 1].b, st);
}

extern "C" void gf_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
}

extern "C" void gf_profile_filter(const char* tag) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_filter = tag ? tag : "";
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
        g_prof_pool.push_back(sp.a);
        g_prof_pool.push_back(sp.b);
    }
    it->second.clear();
    ++g_prof_gen;
    if (total_ms) *total_ms = tot;
    if (count) *count = n;
    if (work) *work = wk;
    return GF_OK;
}
