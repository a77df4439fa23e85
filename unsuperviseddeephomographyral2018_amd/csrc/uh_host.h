// uh_host.h -- host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "../../include/uh_hotpath.h"

namespace uh {

// Optional per-launch timing (uh_profile_enable).  Events are recorded on the SAME stream as the
// launch they bracket, so the figure is the kernel's device-side duration plus the event overhead,
// not host wall time.  Disabled -> zero cost beyond one branch.
extern bool g_prof_on;
extern unsigned g_prof_mask;      // bit k set = kernel k is timed
void prof_begin(int kernel, hipStream_t s);
void prof_end(int kernel, hipStream_t s);

// Event pair for ONE kernel launch, handed to hipExtLaunchKernelGGL: the runtime stamps them with the dispatch's
// own begin/end timestamps, so the figure is the kernel's duration on its stream without the ~3 us that a pair
// of hipEventRecord() calls adds around a short kernel.
void prof_pair(int kernel, hipEvent_t* a, hipEvent_t* b);

// launch `kernel`; when the profiler is on, time exactly this dispatch
template <typename K, typename... Args>
inline void launch_timed(int kid, K kernel, dim3 grid, dim3 block, hipStream_t s, Args... args) {
    if (g_prof_on && ((g_prof_mask >> kid) & 1u)) {
        hipEvent_t a, b;
        prof_pair(kid, &a, &b);
        hipExtLaunchKernelGGL(kernel, grid, block, 0, s, a, b, 0, args...);
    } else {
        hipLaunchKernelGGL(kernel, grid, block, 0, s, args...);
    }
}

// same, with `shm` bytes of dynamic LDS
template <typename K, typename... Args>
inline void launch_timed_shm(int kid, K kernel, dim3 grid, dim3 block, unsigned shm, hipStream_t s, Args... args) {
    if (g_prof_on && ((g_prof_mask >> kid) & 1u)) {
        hipEvent_t a, b;
        prof_pair(kid, &a, &b);
        hipExtLaunchKernelGGL(kernel, grid, block, shm, s, a, b, 0, args...);
    } else {
        hipLaunchKernelGGL(kernel, grid, block, shm, s, args...);
    }
}

struct ProfScope {
    int k; hipStream_t s; bool on;
    ProfScope(int kernel, hipStream_t stream) : k(kernel), s(stream), on(g_prof_on && ((g_prof_mask >> kernel) & 1u)) { if (on) prof_begin(k, s); }
    ~ProfScope() { if (on) prof_end(k, s); }
};

// uh_gather_patch_losses_forward with a second destination for l1_loss (uh_tail.hip writes the caller's loss scalar
// from the loss kernel itself instead of a 4-byte copy node)
int gather_patch_losses(const float* warped, const int* patch_idx, const float* target, const float* h4p, const float* gt,
                        float* pred, float* out16, float* l1_out, void* workspace, size_t workspace_bytes, int B, int H, int W,
                        int C, int P, hipStream_t stream);

}  // namespace uh
