// common.cpp — pfa_last_error / pfa_version and the optional per-kernel HIP-event timing registry
// (bench.py's roofline leg: events are recorded on the SAME stream the kernel is launched on).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/pufferlib_amd.h"

namespace pfa {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

struct TimedKernel {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pairs;
    size_t used = 0;
};
static int g_timing = 0;  // 0 off, 1 the selected (dominant) kernel only, 2 every instrumented kernel
static int g_stride = 1;  // mode 1: bracket every g_stride-th launch of the selected kernel (the event packets serialise the queue)
static long long g_seen = 0;
static std::string g_selected = "ppo_mlp_grad";
static std::map<std::string, TimedKernel> g_timed;

bool timing_enabled() { return g_timing != 0; }

// Returns the stop event (to be recorded after the launch) or nullptr when timing is off.
void *timing_begin(const char *name, hipStream_t stream) {
    if (!name || g_timing == 0 || (g_timing == 1 && g_selected != name)) return nullptr;
    if (g_timing == 1 && g_stride > 1 && (g_seen++ % g_stride) != 0) return nullptr;
    TimedKernel &k = g_timed[name];
    if (k.used == k.pairs.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return nullptr;
        k.pairs.emplace_back(a, b);
    }
    auto &p = k.pairs[k.used++];
    (void)hipEventRecord(p.first, stream);
    return (void *)p.second;
}
void timing_end(void *stop, hipStream_t stream) {
    if (stop) (void)hipEventRecord((hipEvent_t)stop, stream);
}
// An event pair for a launch that carries its own events (hipExtLaunchKernelGGL: the runtime stamps them with the dispatch's begin
// and end, the same two timestamps rocprofv3's kernel trace reports) — nothing is recorded here.  false: timing is off for this
// launch.
bool timing_ext_mode() { return true; }
bool timing_pair(const char *name, hipEvent_t *start, hipEvent_t *stop) {
    const bool ext = timing_ext_mode();
    if (!ext || g_timing == 0 || (g_timing == 1 && g_selected != name)) return false;
    if (g_timing == 1 && g_stride > 1 && (g_seen++ % g_stride) != 0) return false;
    TimedKernel &k = g_timed[name];
    if (k.used == k.pairs.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return false;
        k.pairs.emplace_back(a, b);
    }
    auto &p = k.pairs[k.used++];
    *start = p.first;
    *stop = p.second;
    return true;
}
}  // namespace pfa

extern "C" int pfa_version(void) { return 1; }
extern "C" const char *pfa_last_error(void) { return pfa::g_err; }

extern "C" int pfa_timing_enable(int on) {
    pfa::g_timing = on < 0 ? 0 : (on > 2 ? 2 : on);
    return 0;
}
extern "C" int pfa_timing_stride(int every) {
    pfa::g_stride = every < 1 ? 1 : every;
    pfa::g_seen = 0;
    return 0;
}
extern "C" int pfa_timing_select(const char *kernel) {
    if (!kernel || !*kernel) return -2;
    pfa::g_selected = kernel;
    return 0;
}
extern "C" int pfa_timing_reset(void) {
    for (auto &kv : pfa::g_timed) kv.second.used = 0;
    return 0;
}
extern "C" int pfa_timing_read(const char *kernel, int64_t *launches_host, double *total_ms_host) {
    if (!kernel || !launches_host || !total_ms_host) return -2;
    *launches_host = 0;
    *total_ms_host = 0.0;
    auto it = pfa::g_timed.find(kernel);
    if (it == pfa::g_timed.end()) return 0;
    double total = 0.0;
    for (size_t i = 0; i < it->second.used; ++i) {
        auto &p = it->second.pairs[i];
        if (hipEventSynchronize(p.second) != hipSuccess) return -1;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.first, p.second) != hipSuccess) return -1;
        total += ms;
    }
    *launches_host = (int64_t)it->second.used;
    *total_ms_host = total;
    return 0;
}
