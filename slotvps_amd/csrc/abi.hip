// ABI version and the launch hook of the product library (include/slotvps_hip.h). svps_prof_mark is the one trace point of every
// launcher: with no hook installed it is a relaxed load and a branch. The event bookkeeping that used to live here (rounds 1 - 5) is in
// the diagnostics library (diag_prof.hip -> libslotvps_hip_diag.so), which installs itself through svps_set_launch_hook.
#include <atomic>

#include "../../include/slotvps_hip.h"

namespace {
std::atomic<svps_launch_hook_t> g_hook{nullptr};
}

extern "C" int svps_abi_version(void) { return SVPS_ABI_VERSION; }

extern "C" void svps_set_launch_hook(svps_launch_hook_t hook) { g_hook.store(hook, std::memory_order_release); }

extern "C" void svps_prof_mark(int kernel_id, int is_end, void* stream) {
    const svps_launch_hook_t fn = g_hook.load(std::memory_order_acquire);
    if (fn) fn(kernel_id, is_end, stream);
}
