// Library identification and the optional launch profiler (HIP events around the dominant kernels).
#include "common.h"
#include <mutex>
#include <vector>

#include <string.h>
extern "C" const char* mtd_version(void) { return "mtdgan_hip 0.1.0 (gfx950)"; }

// ---- run-time options: the ONE documented way to change the library's behaviour at run time (include/mtdgan_hip.h lists the
// names).  Everything else that used to be an environment variable is a lab switch of an -DMTD_LAB build (common.h).
namespace {
int g_options[MTD_OPT_COUNT] = {0, 0};
const char* const g_option_names[MTD_OPT_COUNT] = {"c32f_safe_wait", "wino_split"};
int option_id(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < MTD_OPT_COUNT; ++i)
        if (!strcmp(name, g_option_names[i])) return i;
    return -1;
}
}  // namespace
int mtd_option(int id) { return g_options[id]; }
extern "C" int mtd_set_option(const char* name, int value) {
    const int id = option_id(name);
    if (id < 0) return MTD_EINVAL;
    g_options[id] = value;
    return MTD_OK;
}
extern "C" int mtd_get_option(const char* name, int* value) {
    const int id = option_id(name);
    if (id < 0 || !value) return MTD_EINVAL;
    *value = g_options[id];
    return MTD_OK;
}
extern "C" int mtd_lab_build(void) {
#ifdef MTD_LAB
    return 1;
#else
    return 0;
#endif
}

namespace {
struct ProfSlot {
    mtd_prof_record rec;
    hipEvent_t e0, e1;
};
std::mutex g_prof_mu;
std::vector<ProfSlot> g_prof;
int g_prof_cap = 0;
// Timing mode.  1 (default): the two events ride on the kernel's own dispatch packet (hipExtLaunchKernelGGL start / stop
// events), so their difference is the dispatch's begin / end timestamps -- the duration a kernel trace (rocprofv3
// --kernel-trace) reports, with nothing of the bracketing in it.  0 (MTD_PROF_MODE=bracket): hipEventRecord before and
// after the launch, which adds the marker packets' own processing (2.8 us on a 35 us kernel in round 1).
int g_prof_attach = 1;
thread_local int t_pending_slot = -1;
}  // namespace

MtdProfLaunch mtd_prof_launch_events() {
    MtdProfLaunch r{nullptr, nullptr, false};
    if (t_pending_slot < 0) return r;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (t_pending_slot < (int)g_prof.size()) { r.e0 = g_prof[t_pending_slot].e0; r.e1 = g_prof[t_pending_slot].e1; r.on = true; }
    t_pending_slot = -1;
    return r;
}

// Internal hooks (common.h).  begin returns a slot index or -1 when profiling is off / full.
int mtd_prof_begin(int kernel, int cfg, int splitk, long long M, int N, int C, int taps, hipStream_t s, double bytes) {
    if (g_prof_cap <= 0) return -1;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if ((int)g_prof.size() >= g_prof_cap) return -1;
    ProfSlot sl;
    sl.rec.kernel = kernel; sl.rec.cfg = cfg; sl.rec.splitk = splitk;
    sl.rec.M = M; sl.rec.N = N; sl.rec.C = C; sl.rec.taps = taps;
    sl.rec.flops = 2.0 * (double)M * N * C * taps;      // (0 for the byte-moving kernels of class 2: taps == 0)
    sl.rec.ms = 0.f;
    sl.rec.bytes = bytes;
    if (hipEventCreate(&sl.e0) != hipSuccess) return -1;
    if (hipEventCreate(&sl.e1) != hipSuccess) { (void)hipEventDestroy(sl.e0); return -1; }
    if (!g_prof_attach) (void)hipEventRecord(sl.e0, s);
    g_prof.push_back(sl);
    if (g_prof_attach) t_pending_slot = (int)g_prof.size() - 1;      // consumed by the MTD_LAUNCH that follows
    return (int)g_prof.size() - 1;
}

void mtd_prof_end(int slot, hipStream_t s) {
    if (slot < 0) return;
    t_pending_slot = -1;
    if (g_prof_attach) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (slot < (int)g_prof.size()) (void)hipEventRecord(g_prof[slot].e1, s);
}

extern "C" int mtd_prof_mode(int attach) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof.empty()) return MTD_EINVAL;              // not while records are pending
    if (attach >= 0) g_prof_attach = attach ? 1 : 0;
    return g_prof_attach;
}

extern "C" int mtd_prof_enable(int capacity) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& sl : g_prof) { (void)hipEventDestroy(sl.e0); (void)hipEventDestroy(sl.e1); }
    g_prof.clear();
    g_prof_cap = capacity > 0 ? capacity : 0;
    if (g_prof_cap) g_prof.reserve(g_prof_cap);
    static const bool env_bracket = [] { const char* e = mtd_lab_env("MTD_PROF_MODE"); return e && e[0] == 'b'; }();
    if (env_bracket) g_prof_attach = 0;
    return MTD_OK;
}

extern "C" int mtd_prof_collect(mtd_prof_record* out, int max_records) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int n = 0;
    for (auto& sl : g_prof) {
        if (hipEventSynchronize(sl.e1) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, sl.e0, sl.e1) != hipSuccess) continue;
        sl.rec.ms = ms;
        if (out && n < max_records) out[n] = sl.rec;
        ++n;
    }
    return n;
}
