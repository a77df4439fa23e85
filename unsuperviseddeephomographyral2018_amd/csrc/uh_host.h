// uh_host.h -- host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/uh_hotpath.h"

namespace uh {

// Optional per-launch timing (uh_profile_enable).  Events are recorded on the SAME stream as the
// launch they bracket, so the figure is the kernel's device-side duration plus the event overhead,
// not host wall time.  Disabled -> zero cost beyond one branch.
void prof_begin(int kernel, hipStream_t s);
void prof_end(int kernel, hipStream_t s);
extern bool g_prof_on;

struct ProfScope {
    int k; hipStream_t s; bool on;
    ProfScope(int kernel, hipStream_t stream) : k(kernel), s(stream), on(g_prof_on) { if (on) prof_begin(k, s); }
    ~ProfScope() { if (on) prof_end(k, s); }
};

}  // namespace uh
