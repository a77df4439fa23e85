// uh_losses.hip -- all photometric losses of build_losses() in ONE pass over (pred_I2, I2_aug).
//
// The reference evaluates every loss on every step: the active one carries the gradient, the other five are
// stop_gradient monitors fetched for logging (/root/reference/code/homography_model.py:286-352,
// homography_CNN_synthetic.py:345).  As torch ops that is ~30 small launches per step (avg-pools, squares,
// reductions ...); here one kernel produces
//   rec_loss       = sqrt(mean((x - y)^2))                                   homography_model.py:303
//   ssim_loss      = mean(clip((1 - SSIM_3x3(x, y)) / 2, 0, 1)), VALID 3x3   :141-158
//   l1_loss        = mean(|x - y|)                                           :328
//   l1_smooth_loss = mean(|d| < 1 ? 0.5 d^2 : |d| - 0.5)                     :136-139
//   ncc_loss       = sqrt(sum((y/|y| - x/|x|)^2)) = sqrt(2 - 2 <x,y>/(|x||y|))   :161-166 (called as _NCC_loss(I2, pred))
// with x = pred_I2, y = I2_aug, both [B,P,P] (one channel).  Forward values only: the gradient-carrying loss of
// the hot path (l1_loss) has its own kernels; the other losses keep a torch-autograd path when they are trained on.
// HBM-bound and tiny (2*B*P*P*4 bytes = 8.4 MB at B=64): the point is the launch count.
#include "uh_device.h"
#include "uh_host.h"

namespace uh {

constexpr int NLS = 7;      // |d|, d^2, smooth-l1, x^2, y^2, x*y, ssim

// one thread per patch pixel (i, j): its point-wise terms, plus the SSIM of the 3x3 window whose top-left
// corner it is (i, j < P-2).  The 2 x 9 window taps come from L1/L2 (the patch is 64 KiB per image).
__global__ __launch_bounds__(256) void patch_losses_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                           float* __restrict__ partial, int P, int blocks_per_image) {
    __shared__ float red[NLS][16];
    const int b = blockIdx.x / blocks_per_image, chunk = blockIdx.x - b * blocks_per_image;
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int n = P * P;
    const float* __restrict__ x = X + (size_t)b * n;
    const float* __restrict__ y = Y + (size_t)b * n;
    float acc[NLS];
#pragma unroll
    for (int k = 0; k < NLS; ++k) acc[k] = 0.f;
    const int e = chunk * 256 + (int)threadIdx.x;
    if (e < n) {
        const int i = e / P, j = e - i * P;
        const float xv = x[e], yv = y[e];
        const float d = xv - yv, ad = fabsf(d);
        acc[0] = ad; acc[1] = d * d;
        acc[2] = ad < 1.0f ? 0.5f * (ad * ad) : ad - 0.5f;
        acc[3] = xv * xv; acc[4] = yv * yv; acc[5] = xv * yv;
        if (i < P - 2 && j < P - 2) {
            float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    const float a = x[e + u * P + v], c = y[e + u * P + v];
                    sx += a; sy += c; sxx += a * a; syy += c * c; sxy += a * c;
                }
            const float inv9 = 1.0f / 9.0f;
            const float mux = sx * inv9, muy = sy * inv9;
            const float sgx = sxx * inv9 - mux * mux, sgy = syy * inv9 - muy * muy, sgxy = sxy * inv9 - mux * muy;
            const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
            const float ssim = ((2.f * mux * muy + C1) * (2.f * sgxy + C2)) / ((mux * mux + muy * muy + C1) * (sgx + sgy + C2));
            acc[6] = fminf(fmaxf((1.f - ssim) * 0.5f, 0.f), 1.f);
        }
    }
#pragma unroll
    for (int k = 0; k < NLS; ++k) {
        const float r = row16_sum(acc[k]);
        if ((lane & 15) == 0) red[k][wave * 4 + (lane >> 4)] = r;
    }
    __syncthreads();
    if (threadIdx.x < NLS) {
        const float* r = red[threadIdx.x];
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += r[k];
        partial[(size_t)blockIdx.x * NLS + threadIdx.x] = t;
    }
}

// out[0..4] = rec, ssim, l1, l1_smooth, ncc;  out[5] = h_loss = sqrt(mean((h4p - gt)^2)) when h4p != NULL.
__global__ __launch_bounds__(256) void patch_losses_finish_kernel(const float* __restrict__ partial, int nblk,
                                                                  const float* __restrict__ h4p,
                                                                  const float* __restrict__ gt, int nh,
                                                                  float* __restrict__ out, double inv_n, double inv_ns) {
    __shared__ double red[NLS + 1][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double a[NLS + 1];
#pragma unroll
    for (int k = 0; k <= NLS; ++k) a[k] = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) {
#pragma unroll
        for (int k = 0; k < NLS; ++k) a[k] += (double)partial[(size_t)i * NLS + k];
    }
    if (h4p) for (int i = threadIdx.x; i < nh; i += 256) { const double d = (double)h4p[i] - (double)gt[i]; a[NLS] += d * d; }
#pragma unroll
    for (int k = 0; k <= NLS; ++k) {
        const double s = wave_sum(a[k]);
        if (lane == 0) red[k][wave] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s[NLS + 1];
#pragma unroll
        for (int k = 0; k <= NLS; ++k) s[k] = (red[k][0] + red[k][1]) + (red[k][2] + red[k][3]);
        out[0] = (float)sqrt(s[1] * inv_n);
        out[1] = (float)(s[6] * inv_ns);
        out[2] = (float)(s[0] * inv_n);
        out[3] = (float)(s[2] * inv_n);
        const double den = sqrt(s[3] * s[4]);
        const double c = den > 0.0 ? s[5] / den : 0.0;          // x or y identically 0: the reference yields NaN; monitor only
        out[4] = (float)sqrt(fmax(0.0, 2.0 - 2.0 * c));
        out[5] = h4p ? (float)sqrt(s[NLS] / (double)nh) : 0.f;
    }
}

}  // namespace uh

using namespace uh;

extern "C" size_t uh_patch_losses_workspace_bytes(int B, int P) {
    if (B <= 0 || P <= 0) return 0;
    return (size_t)B * ((P * P + 255) / 256) * NLS * sizeof(float);
}

extern "C" int uh_patch_losses_forward(const float* pred, const float* target, const float* h4p, const float* gt,
                                       float* out6, void* workspace, size_t workspace_bytes, int B, int P,
                                       uh_stream_t stream) {
    if (!pred || !target || !out6) return UH_E_NULL;
    if ((h4p == nullptr) != (gt == nullptr)) return UH_E_NULL;
    if (B <= 0 || P < 3) return UH_E_SHAPE;
    if ((uint64_t)B * P * P >= (1ull << 31)) return UH_E_TOO_LARGE;
    if (!workspace || workspace_bytes < uh_patch_losses_workspace_bytes(B, P)) return UH_E_WORKSPACE;
    const int bpi = (P * P + 255) / 256;
    hipStream_t s = (hipStream_t)stream;
    launch_timed(UH_K_LOSSES, patch_losses_kernel, dim3((unsigned)B * bpi), dim3(256), s, pred, target, (float*)workspace, P, bpi);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    launch_timed(UH_K_LOSSES_FIN, patch_losses_finish_kernel, dim3(1), dim3(256), s, (const float*)workspace, B * bpi, h4p, gt,
                 B * 8, out6, 1.0 / ((double)B * P * P), 1.0 / ((double)B * (P - 2) * (P - 2)));
    return (int)hipGetLastError();
}
