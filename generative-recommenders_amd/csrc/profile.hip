// Optional in-library kernel timer used by the benchmark: when enabled, the launch code brackets the
// dominant kernel of an entry point with HIP events on the caller's stream.  Off by default; a
// disabled timer costs one predictable branch per launch and creates nothing.
#include <mutex>
#include <string>
#include <vector>

#include "hmx_common.h"
#include "profile.h"

namespace {
struct Span {
    std::string tag;
    hipEvent_t start, stop;
};
std::mutex g_mu;
bool g_on = false;
std::string g_only;          // non-empty: only spans with this tag are recorded
std::vector<Span> g_spans;
}  // namespace

bool hmx_profile_on() { return g_on; }

void* hmx_profile_begin(const char* tag, hipStream_t s) {
    if (!g_on) return nullptr;
    if (!g_only.empty() && g_only != tag) return nullptr;
    Span sp{tag, nullptr, nullptr};
    if (hipEventCreate(&sp.start) != hipSuccess || hipEventCreate(&sp.stop) != hipSuccess) return nullptr;
    (void)hipEventRecord(sp.start, s);
    std::lock_guard<std::mutex> lk(g_mu);
    g_spans.push_back(sp);
    return reinterpret_cast<void*>(g_spans.size());  // 1-based index
}

void hmx_profile_end(void* token, hipStream_t s) {
    if (!token) return;
    std::lock_guard<std::mutex> lk(g_mu);
    (void)hipEventRecord(g_spans[reinterpret_cast<size_t>(token) - 1].stop, s);
}

extern "C" int hmx_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = on != 0;
    return HMX_OK;
}

extern "C" int hmx_profile_filter(const char* tag) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_only = tag ? tag : "";
    return HMX_OK;
}

// Sums (and forgets) the recorded spans whose tag equals `tag`; blocks until their events completed.
extern "C" int hmx_profile_collect(const char* tag, double* total_ms, int64_t* launches) {
    if (!tag || !total_ms || !launches) return HMX_EINVAL;
    std::lock_guard<std::mutex> lk(g_mu);
    *total_ms = 0.0;
    *launches = 0;
    std::vector<Span> keep;
    for (auto& sp : g_spans) {
        if (sp.tag != tag) {
            keep.push_back(sp);
            continue;
        }
        float ms = 0.f;
        if (hipEventSynchronize(sp.stop) == hipSuccess && hipEventElapsedTime(&ms, sp.start, sp.stop) == hipSuccess) {
            *total_ms += ms;
            *launches += 1;
        }
        (void)hipEventDestroy(sp.start);
        (void)hipEventDestroy(sp.stop);
    }
    g_spans.swap(keep);
    return HMX_OK;
}
