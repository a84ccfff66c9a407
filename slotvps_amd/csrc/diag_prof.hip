// DIAGNOSTICS library (libslotvps_hip_diag.so, include/slotvps_hip_diag.h): per-kernel device timing with HIP events on the launch
// stream (bench.py's roofline leg). svps_diag_launch_hook is installed into the product library with svps_set_launch_hook.
#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

#include "../../include/slotvps_hip.h"
#include "../../include/slotvps_hip_diag.h"

namespace {

struct Span {
    hipEvent_t begin = nullptr;
    hipEvent_t end = nullptr;
};

struct Prof {
    bool on = false;
    std::mutex mu;
    std::vector<Span> spans[SVPS_KERNEL_COUNT];
    std::vector<hipEvent_t> pool;  // recycled events

    hipEvent_t get() {
        if (!pool.empty()) {
            hipEvent_t e = pool.back();
            pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
};

Prof& prof() {
    static Prof p;
    return p;
}

}  // namespace

extern "C" void svps_prof_enable(int on) { prof().on = on != 0; }

extern "C" void svps_prof_reset(void) {
    Prof& p = prof();
    std::lock_guard<std::mutex> g(p.mu);
    for (auto& v : p.spans) {
        for (Span& s : v) {
            if (s.begin) p.pool.push_back(s.begin);
            if (s.end) p.pool.push_back(s.end);
        }
        v.clear();
    }
}

extern "C" void svps_diag_launch_hook(int kernel_id, int is_end, void* stream) {
    Prof& p = prof();
    if (!p.on) return;
    if (kernel_id < 0 || kernel_id >= SVPS_KERNEL_COUNT) return;
    std::lock_guard<std::mutex> g(p.mu);
    auto& v = p.spans[kernel_id];
    if (!is_end) {
        Span s;
        s.begin = p.get();
        if (!s.begin) return;
        (void)hipEventRecord(s.begin, static_cast<hipStream_t>(stream));
        v.push_back(s);
    } else if (!v.empty() && v.back().begin && !v.back().end) {
        hipEvent_t e = p.get();
        if (!e) return;
        (void)hipEventRecord(e, static_cast<hipStream_t>(stream));
        v.back().end = e;
    }
}

extern "C" int svps_prof_collect(int kernel_id, double* total_ms, int* launches) {
    if (kernel_id < 0 || kernel_id >= SVPS_KERNEL_COUNT || !total_ms || !launches) return SVPS_ERR_BAD_ARG;
    Prof& p = prof();
    std::lock_guard<std::mutex> g(p.mu);
    double sum = 0.0;
    int n = 0;
    for (Span& s : p.spans[kernel_id]) {
        if (!s.begin || !s.end) continue;
        hipError_t e = hipEventSynchronize(s.end);
        if (e != hipSuccess) return (int)e;
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, s.begin, s.end);
        if (e != hipSuccess) return (int)e;
        sum += ms;
        ++n;
    }
    *total_ms = sum;
    *launches = n;
    return 0;
}
