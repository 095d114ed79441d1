// gp_profile.hip -- optional per-kernel timing with hipEvent pairs recorded on the launch stream
// (torch.cuda.Event only sees torch's current stream; these see exactly the kernels they bracket).
#include <mutex>
#include <string>
#include <vector>

#include "gp_common.h"

struct ProfRec { const char* name; hipEvent_t a, b; };
static std::mutex g_mu;
static int g_level = 0;   // 0 off, 1 = only the kernels tagged level 1 (roofline kernel), 2 = every kernel
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;

static hipEvent_t take_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

bool gp_prof_on() { return g_level > 0; }

void* gp_prof_begin(const char* name, hipStream_t s, int level) {
    if (g_level < level) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    ProfRec r{name, take_event(), take_event()};
    if (!r.a || !r.b) return nullptr;
    hipEventRecord(r.a, s);
    g_recs.push_back(r);
    return (void*)(uintptr_t)g_recs.size();  // 1-based index
}
void gp_prof_end(void* h, hipStream_t s) {
    if (!h) return;
    std::lock_guard<std::mutex> lk(g_mu);
    size_t i = (size_t)(uintptr_t)h - 1;
    if (i < g_recs.size()) hipEventRecord(g_recs[i].b, s);
}

extern "C" int gp_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_level = on < 0 ? 0 : on;
    return 0;
}

extern "C" int gp_profile_collect(gp_profile_entry* out, int max_entries, int* n_out) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!n_out) GP_FAIL("null n_out");
    int n = 0;
    for (auto& r : g_recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            int k = 0;
            for (; k < n; ++k)
                if (strncmp(out[k].name, r.name, sizeof(out[k].name)) == 0) break;
            if (k == n && n < max_entries && out) {
                memset(&out[n], 0, sizeof(out[n]));
                strncpy(out[n].name, r.name, sizeof(out[n].name) - 1);
                ++n;
            }
            if (k < n) { out[k].launches += 1; out[k].total_ms += ms; }
        }
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_recs.clear();
    *n_out = n;
    return 0;
}
