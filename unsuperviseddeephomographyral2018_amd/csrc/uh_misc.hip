// uh_misc.hip -- the glue ops either side of the warp on the reference's (un-fused) path, plus the
// library-level services (error strings, launch profiler).
//
//   gray + patch gather   <- /root/reference/code/homography_model.py:74-76,263-269
//   photometric L1        <- /root/reference/code/homography_model.py:328
//
// These are plain streaming kernels (HBM-bound, a few MB); they exist so that the whole
// solve_DLT -> transform -> l1_loss chain runs in this library without a torch op in between.
#include "uh_device.h"
#include "uh_host.h"
#include <cstdlib>
#include <mutex>
#include <vector>

namespace uh {

// pred[k,i] = (w0 + w1 + ...)/C at flat pixel patch_idx[k,i] of image k
template <int C>
__global__ __launch_bounds__(256) void gray_patch_forward_kernel(const float* __restrict__ warped,
                                                                 const int* __restrict__ patch_idx,
                                                                 float* __restrict__ pred, int HW, int PP, int B) {
    const size_t n = (size_t)B * PP;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int k = (int)(i / PP);
        const float* p = warped + ((size_t)k * HW + patch_idx[i]) * C;
        float s = p[0];
#pragma unroll
        for (int c = 1; c < C; ++c) s = s + p[c];
        pred[i] = s / (float)C;
    }
}

// Backward of the gather = scatter-add of dPred/C into a zero frame (duplicates sum, like tf.gather's gradient).
// The dataloader's indices are a P x P rectangle, (v + y0)*W + (u + x0): memset + 3 float atomics per patch pixel
// (43 us at B=64) is a poor way to write a mostly-zero 59 MB frame.  Two kernels, no memset, still fully general:
//   dense  : every frame pixel is written exactly once -- the value of the patch entry that SHOULD sit there if the
//            patch is the rectangle anchored at patch_idx[k,0] (and whose stored index confirms it), else 0;
//   fix-up : every patch entry whose stored index is NOT its rectangle position (arbitrary gathers, duplicates,
//            PP not a square -> P = 0 -> every entry) is added atomically on top.  For rectangles it adds nothing.
template <int C>
__global__ __launch_bounds__(256) void gray_patch_backward_dense_kernel(const float* __restrict__ dPred,
                                                                        const int* __restrict__ patch_idx,
                                                                        float* __restrict__ dWarped, int HW, int W, int P,
                                                                        int PP) {
    const int k = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int* idx = patch_idx + (size_t)k * PP;
    const int o = idx[0];                                    // uniform -> scalar load
    const int y0 = o / W, x0 = o - y0 * W;
    const int y = p / W, x = p - y * W;
    const int u = x - x0, v = y - y0;
    float g = 0.f;
    if (u >= 0 && u < P && v >= 0 && v < P) {
        const int e = v * P + u;
        if (idx[e] == p) g = dPred[(size_t)k * PP + e] / (float)C;
    }
    float* q = dWarped + ((size_t)k * HW + p) * C;
#pragma unroll
    for (int c = 0; c < C; ++c) q[c] = g;
}

template <int C>
__global__ __launch_bounds__(256) void gray_patch_backward_fixup_kernel(const float* __restrict__ dPred,
                                                                        const int* __restrict__ patch_idx,
                                                                        float* __restrict__ dWarped, int HW, int W, int H,
                                                                        int P, int PP, int B) {
    const size_t n = (size_t)B * PP;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int k = (int)(i / PP), e = (int)(i - (size_t)k * PP);
        const int t = patch_idx[i];
        bool direct = false;
        if (P > 0) {
            const int o = patch_idx[(size_t)k * PP];
            const int y0 = o / W, x0 = o - y0 * W;
            const int v = e / P, u = e - v * P;
            direct = (x0 + u < W) && (y0 + v < H) && t == (y0 + v) * W + (x0 + u);
        }
        if (!direct && t >= 0 && t < HW) {                  // (an index outside the frame -- tf.gather would raise -- adds nothing)
            float* q = dWarped + ((size_t)k * HW + t) * C;
            const float g = dPred[i] / (float)C;
#pragma unroll
            for (int c = 0; c < C; ++c) atomicAdd(q + c, g);
        }
    }
}

constexpr int L1_BLOCKS = 1024;

// the grid's partial sums -> loss, by one wave, f64, fixed order
__global__ __launch_bounds__(256) void l1_partial_kernel(const float* __restrict__ pred,
                                                         const float* __restrict__ target,
                                                         float* __restrict__ partial, size_t n) {
    __shared__ float red[4];
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        acc += fabsf(pred[i] - target[i]);
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(64) void l1_finish_kernel(const float* __restrict__ partial, float* __restrict__ loss,
                                                       int nblk, double inv_n) {
    const int lane = (int)threadIdx.x;
    double acc = 0.0;
    for (int i = lane; i < nblk; i += 64) acc += (double)partial[i];
    acc = wave_sum(acc);
    if (lane == 0) loss[0] = (float)(acc * inv_n);
}

__global__ __launch_bounds__(256) void l1_backward_kernel(const float* __restrict__ pred,
                                                          const float* __restrict__ target,
                                                          const float* __restrict__ dLoss,
                                                          float* __restrict__ dPred, size_t n) {
    const float g = dLoss[0] / (float)n;          // reduce_mean grad: dLoss / n, then * sign
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float d = pred[i] - target[i];
        dPred[i] = d > 0.f ? g : (d < 0.f ? -g : 0.f * g);
    }
}

// ---- launch profiler -----------------------------------------------------------------------------
bool g_prof_on = false;
unsigned g_prof_mask = 0xffffffffu;
namespace {
struct Rec { int k; hipEvent_t a, b; };
std::mutex g_mu;
std::vector<Rec> g_open, g_done;
std::vector<hipEvent_t> g_pool;
double g_ms[UH_K_COUNT];
long long g_n[UH_K_COUNT];
hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    // Timing-only events: without the system-scope fence a default event performs when it completes (a cache writeback +
    // invalidate that perturbs the very kernels being timed -- hip_runtime_api.h on hipEventDisableSystemFence).
    // UH_PROF_FENCE=1 restores default events (rounds 1-3) for an A/B.
    static const bool fence = [] { const char* v = getenv("UH_PROF_FENCE"); return v && v[0] == '1'; }();
    hipEvent_t e; (void)hipEventCreateWithFlags(&e, fence ? hipEventDefault : hipEventDisableSystemFence); return e;
}
}  // namespace

void prof_begin(int kernel, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    Rec r{kernel, get_event(), get_event()};
    (void)hipEventRecord(r.a, s);
    g_open.push_back(r);
}
void prof_pair(int kernel, hipEvent_t* a, hipEvent_t* b) {
    std::lock_guard<std::mutex> lk(g_mu);
    Rec r{kernel, get_event(), get_event()};
    *a = r.a; *b = r.b;
    g_done.push_back(r);
}
void prof_end(int kernel, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (size_t i = g_open.size(); i-- > 0;) {
        if (g_open[i].k == kernel) {
            (void)hipEventRecord(g_open[i].b, s);
            g_done.push_back(g_open[i]);
            g_open.erase(g_open.begin() + i);
            return;
        }
    }
}

}  // namespace uh

using namespace uh;

extern "C" int uh_abi_version(void) { return UH_ABI_VERSION; }

extern "C" const char* uh_error_string(int code) {
    switch (code) {
        case 0: return "ok";
        case UH_E_NULL: return "UH_E_NULL: a required pointer is NULL";
        case UH_E_SHAPE: return "UH_E_SHAPE: non-positive or inconsistent dimension";
        case UH_E_CHANNELS: return "UH_E_CHANNELS: C must be 1..4";
        case UH_E_WORKSPACE: return "UH_E_WORKSPACE: workspace missing or too small";
        case UH_E_TOO_LARGE: return "UH_E_TOO_LARGE: image or index range exceeds 32-bit addressing";
        case UH_E_CAPTURING: return "UH_E_CAPTURING: a synchronising call on a stream that is being captured";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown uh error";
    }
}

extern "C" const char* uh_kernel_name(int k) {
    static const char* names[UH_K_COUNT] = {"dlt_forward", "dlt_backward", "warp_forward", "warp_backward",
                                            "warp_backward_finish", "gray_patch_forward", "gray_patch_backward",
                                            "l1_forward", "l1_backward", "warp_patch_l1_fused", "warp_patch_l1_finish",
                                            "patch_losses", "patch_losses_finish", "prepare_inputs",
                                            "bias_relu_forward", "bias_relu_backward", "patch_loss_backward"};
    return (k >= 0 && k < UH_K_COUNT) ? names[k] : "?";
}

extern "C" int uh_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    int prev = g_prof_on ? 1 : 0;
    for (auto& r : g_done) { g_pool.push_back(r.a); g_pool.push_back(r.b); }
    for (auto& r : g_open) { g_pool.push_back(r.a); g_pool.push_back(r.b); }
    g_done.clear(); g_open.clear();
    for (int i = 0; i < UH_K_COUNT; ++i) { g_ms[i] = 0; g_n[i] = 0; }
    g_prof_on = on != 0;
    // on == 1: every kernel; otherwise bit (k + 1) of `on` selects kernel k (UH_K_*), e.g. (1 << (UH_K_WARP_FWD + 1))
    g_prof_mask = (on == 1 || on == 0) ? 0xffffffffu : ((unsigned)on >> 1);
    return prev;
}

extern "C" int uh_profile_read(double* total_ms, long long* launches) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& r : g_done) {
        hipError_t e = hipEventSynchronize(r.b);
        if (e != hipSuccess) return (int)e;
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, r.a, r.b);
        if (e != hipSuccess) return (int)e;
        g_ms[r.k] += ms; g_n[r.k] += 1;
        g_pool.push_back(r.a); g_pool.push_back(r.b);
    }
    g_done.clear();
    for (int i = 0; i < UH_K_COUNT; ++i) {
        if (total_ms) total_ms[i] = g_ms[i];
        if (launches) launches[i] = g_n[i];
    }
    return 0;
}

// ---- gray + patch gather -----------------------------------------------------------------------------
static int check_gray(const void* a, const void* b, const void* c, int B, int H, int W, int C, int PP) {
    if (!a || !b || !c) return UH_E_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || PP <= 0) return UH_E_SHAPE;
    if (C < 1 || C > 4) return UH_E_CHANNELS;
    if ((uint64_t)H * W >= (1ull << 31)) return UH_E_TOO_LARGE;
    return 0;
}
static unsigned grid_for(size_t n) {
    size_t g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

extern "C" int uh_gray_patch_forward(const float* warped, const int* patch_idx, float* pred, int B, int H, int W,
                                     int C, int PP, uh_stream_t stream) {
    if (int e = check_gray(warped, patch_idx, pred, B, H, W, C, PP)) return e;
    hipStream_t s = (hipStream_t)stream;
    const unsigned g = grid_for((size_t)B * PP);
    ProfScope prof(UH_K_GRAY_FWD, s);
    switch (C) {
        case 1: hipLaunchKernelGGL(gray_patch_forward_kernel<1>, dim3(g), dim3(256), 0, s, warped, patch_idx, pred, H * W, PP, B); break;
        case 2: hipLaunchKernelGGL(gray_patch_forward_kernel<2>, dim3(g), dim3(256), 0, s, warped, patch_idx, pred, H * W, PP, B); break;
        case 3: hipLaunchKernelGGL(gray_patch_forward_kernel<3>, dim3(g), dim3(256), 0, s, warped, patch_idx, pred, H * W, PP, B); break;
        default: hipLaunchKernelGGL(gray_patch_forward_kernel<4>, dim3(g), dim3(256), 0, s, warped, patch_idx, pred, H * W, PP, B); break;
    }
    return (int)hipGetLastError();
}

extern "C" int uh_gray_patch_backward(const float* dPred, const int* patch_idx, float* dWarped, int B, int H, int W,
                                      int C, int PP, uh_stream_t stream) {
    if (int e = check_gray(dPred, patch_idx, dWarped, B, H, W, C, PP)) return e;
    if (B > 65535) return UH_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(UH_K_GRAY_BWD, s);
    int P = 0;
    for (int r = 1; (long long)r * r <= PP; ++r) if (r * r == PP) P = r;          // PP not a square: P = 0 (all entries via fix-up)
    const dim3 gd((H * W + 255) / 256, B);
    const unsigned gf = grid_for((size_t)B * PP);
#define UH_GPB(CC) do { \
        hipLaunchKernelGGL(gray_patch_backward_dense_kernel<CC>, gd, dim3(256), 0, s, dPred, patch_idx, dWarped, H * W, W, P, PP); \
        hipLaunchKernelGGL(gray_patch_backward_fixup_kernel<CC>, dim3(gf), dim3(256), 0, s, dPred, patch_idx, dWarped, H * W, W, H, P, PP, B); \
    } while (0)
    switch (C) {
        case 1: UH_GPB(1); break;
        case 2: UH_GPB(2); break;
        case 3: UH_GPB(3); break;
        default: UH_GPB(4); break;
    }
#undef UH_GPB
    return (int)hipGetLastError();
}

// ---- photometric L1 ------------------------------------------------------------------------------------
extern "C" size_t uh_l1_loss_workspace_bytes(size_t n) { (void)n; return L1_BLOCKS * sizeof(float); }

extern "C" int uh_l1_loss_forward(const float* pred, const float* target, float* loss, void* workspace,
                                  size_t workspace_bytes, size_t n, uh_stream_t stream) {
    if (!pred || !target || !loss) return UH_E_NULL;
    if (n == 0) return UH_E_SHAPE;
    if (!workspace || workspace_bytes < uh_l1_loss_workspace_bytes(n)) return UH_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    size_t gb = (n + 255) / 256;
    const int nblk = (int)(gb > (size_t)L1_BLOCKS ? (size_t)L1_BLOCKS : gb);
    ProfScope prof(UH_K_L1_FWD, s);
    hipLaunchKernelGGL(l1_partial_kernel, dim3(nblk), dim3(256), 0, s, pred, target, (float*)workspace, n);
    hipLaunchKernelGGL(l1_finish_kernel, dim3(1), dim3(64), 0, s, (const float*)workspace, loss, nblk, 1.0 / (double)n);
    return (int)hipGetLastError();
}

extern "C" int uh_l1_loss_backward(const float* pred, const float* target, const float* dLoss, float* dPred,
                                   size_t n, uh_stream_t stream) {
    if (!pred || !target || !dLoss || !dPred) return UH_E_NULL;
    if (n == 0) return UH_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(UH_K_L1_BWD, s);
    hipLaunchKernelGGL(l1_backward_kernel, dim3(grid_for(n)), dim3(256), 0, s, pred, target, dLoss, dPred, n);
    return (int)hipGetLastError();
}
