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
// with x = pred_I2, y = I2_aug, both [B,P,P] (one channel), and -- for whichever of them is being trained on --
// d loss / d pred_I2 in one more pass (uh_patch_loss_backward; the reference gets it from TF autodiff of
// homography_model.py:136-166,298-352).  HBM-bound and tiny (2*B*P*P*4 bytes = 8.4 MB at B=64): the point is the
// launch count.
#include "uh_device.h"
#include "uh_host.h"

namespace uh {

constexpr int NLS = 7;      // |d|, d^2, smooth-l1, x^2, y^2, x*y, ssim
constexpr int LPT = 4;      // patch pixels per thread of patch_losses_kernel

// out[0..4] = rec, ssim, l1, l1_smooth, ncc;  out[5] = h_loss = sqrt(mean((h4p - gt)^2)) when h4p != NULL.
// One block of 256 threads.
struct LossFinish { const float* h4p; const float* gt; int nh; float* out; double inv_n, inv_ns;
                    float* l1_out; };      // optional second destination of l1_loss (uh_tail: the caller's loss scalar)
__device__ __forceinline__ void patch_losses_finish(const float* __restrict__ partial, int nblk, const LossFinish& f,
                                                    double (*red)[4]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double a[NLS + 1];
#pragma unroll
    for (int k = 0; k <= NLS; ++k) a[k] = 0.0;
    for (int i0 = threadIdx.x; i0 < nblk; i0 += 256 * 4) {          // 4 rows x 7 loads in flight, added in row order
        float v[4][NLS];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < NLS; ++k) { const bool ok = i0 + 256 * q < nblk; const float t = partial[ok ? (i0 + 256 * q) * NLS + k : 0]; v[q][k] = ok ? t : 0.f; }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < NLS; ++k) a[k] += (double)v[q][k];
    }
    if (f.h4p) for (int i = threadIdx.x; i < f.nh; i += 256) { const double d = (double)f.h4p[i] - (double)f.gt[i]; a[NLS] += d * d; }
#pragma unroll
    for (int k = 0; k <= NLS; ++k) {
        const double s = wave_sum(a[k]);
        if (lane == 0) red[k][wave] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* out = f.out;
        double s[NLS + 1];
#pragma unroll
        for (int k = 0; k <= NLS; ++k) s[k] = (red[k][0] + red[k][1]) + (red[k][2] + red[k][3]);
        out[0] = (float)sqrt(s[1] * f.inv_n);
        out[1] = (float)(s[6] * f.inv_ns);
        out[2] = (float)(s[0] * f.inv_n);
        if (f.l1_out) *f.l1_out = out[2];
        out[3] = (float)(s[2] * f.inv_n);
        const double den = sqrt(s[3] * s[4]);
        const double c = den > 0.0 ? s[5] / den : 0.0;          // x or y identically 0: the reference yields NaN; monitor only
        out[4] = (float)sqrt(fmax(0.0, 2.0 - 2.0 * c));
        out[5] = f.h4p ? (float)sqrt(s[NLS] / (double)f.nh) : 0.f;
        // raw sums for uh_patch_loss_backward (the global norms of rec / ncc): |d|, d^2, smooth, x^2, y^2, xy, ssim
#pragma unroll
        for (int k = 0; k < NLS; ++k) out[6 + k] = (float)s[k];
        out[13] = 0.f; out[14] = 0.f; out[15] = 0.f;
    }
}
__global__ __launch_bounds__(256) void patch_losses_finish_kernel(const float* __restrict__ partial, int nblk, LossFinish f) {
    __shared__ double red[NLS + 1][4];
    patch_losses_finish(partial, nblk, f, red);
}

// one thread per patch pixel (i, j): its point-wise terms, plus the SSIM of the 3x3 window whose top-left
// corner it is (i, j < P-2).  The 2 x 9 window taps come from L1/L2 (the patch is 64 KiB per image).
// GC > 0 (uh_gather_patch_losses_forward): X is not pred_I2 but the warped FRAME [B,H,W,GC]; the block first forms its
// pixels' pred = gray(frame[patch_idx]) (reduce_mean(axis=3) + gather, homography_model.py:263-269 -- the arithmetic of
// gray_patch_forward_kernel), writes them to `pred`, and keeps them -- with the two rows + two pixels after its chunk
// that its SSIM windows reach into -- in LDS.  Same statistics, bit for bit, as gather kernel + this kernel on pred.
struct GatherArgs { const int* idx; float* pred; int HW; };
template <int GC>
__global__ __launch_bounds__(256) void patch_losses_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                           float* __restrict__ partial, int P, int blocks_per_image,
                                                           GatherArgs ga) {
    __shared__ float red[NLS][16];
    extern __shared__ float xt[];                 // GC > 0: LPT*256 + 2P + 3 floats
    const int b = blockIdx.x / blocks_per_image, chunk = blockIdx.x - b * blocks_per_image;
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int n = P * P;
    const int c0 = chunk * LPT * 256;
    const float* __restrict__ x = X + (size_t)b * n;
    const float* __restrict__ y = Y + (size_t)b * n;
    if constexpr (GC > 0) {
        const int m = min(n - c0, LPT * 256 + 2 * P + 3);
        const int* __restrict__ idx = ga.idx + (size_t)b * n + c0;
        const float* __restrict__ frame = X + (size_t)b * ga.HW * GC;
        float* __restrict__ pred = ga.pred + (size_t)b * n + c0;
        // batches of GB pixels per thread: all index loads, then all frame loads, in flight together (one block has only
        // ~5 pixels per thread, and a rolled loop would pay two memory latencies for each of them)
        constexpr int GB = 6;
        for (int t0 = threadIdx.x; t0 < m; t0 += 256 * GB) {
            int id[GB];
#pragma unroll
            for (int q = 0; q < GB; ++q) id[q] = idx[min(t0 + 256 * q, m - 1)];
            float g[GB][GC];
#pragma unroll
            for (int q = 0; q < GB; ++q)
#pragma unroll
                for (int c = 0; c < GC; ++c) g[q][c] = frame[(size_t)id[q] * GC + c];
#pragma unroll
            for (int q = 0; q < GB; ++q) {
                const int t = t0 + 256 * q;
                float s = g[q][0];
#pragma unroll
                for (int c = 1; c < GC; ++c) s = s + g[q][c];
                const float v = s / (float)GC;
                if (t < m) {
                    xt[t] = v;
                    if (t < LPT * 256) pred[t] = v;
                }
            }
        }
        __syncthreads();
    }
    auto xat = [&](int e) { if constexpr (GC > 0) return xt[e - c0]; else return x[e]; };
    float acc[NLS];
#pragma unroll
    for (int k = 0; k < NLS; ++k) acc[k] = 0.f;
    // LPT pixels per thread: 4x fewer partial rows for the one-block finish to walk (it was the slowest launch of the tail)
#pragma unroll
    for (int q = 0; q < LPT; ++q) {
    const int e = (chunk * LPT + q) * 256 + (int)threadIdx.x;
    if (e < n) {
        const int i = e / P, j = e - i * P;
        const float xv = xat(e), yv = y[e];
        const float d = xv - yv, ad = fabsf(d);
        acc[0] += ad; acc[1] += d * d;
        acc[2] += ad < 1.0f ? 0.5f * (ad * ad) : ad - 0.5f;
        acc[3] += xv * xv; acc[4] += yv * yv; acc[5] += xv * yv;
        if (i < P - 2 && j < P - 2) {
            float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    const float a = xat(e + u * P + v), c = y[e + u * P + v];
                    sx += a; sy += c; sxx += a * a; syy += c * c; sxy += a * c;
                }
            const float inv9 = 1.0f / 9.0f;
            const float mux = sx * inv9, muy = sy * inv9;
            const float sgx = sxx * inv9 - mux * mux, sgy = syy * inv9 - muy * muy, sgxy = sxy * inv9 - mux * muy;
            const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
            const float ssim = ((2.f * mux * muy + C1) * (2.f * sgxy + C2)) / ((mux * mux + muy * muy + C1) * (sgx + sgy + C2));
            acc[6] += fminf(fmaxf((1.f - ssim) * 0.5f, 0.f), 1.f);
        }
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

// ---- gradients -------------------------------------------------------------------------------------------------
// Point-wise losses (loss_coef / loss_grad_point in uh_device.h).
template <int KIND>
__global__ __launch_bounds__(256) void patch_loss_backward_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                                  const float* __restrict__ stats,
                                                                  const float* __restrict__ dLoss,
                                                                  float* __restrict__ dX, size_t n) {
    const LossCoef lc = loss_coef(KIND, stats, dLoss[0], n);
    auto grad = [&](float x, float y) { return loss_grad_point(KIND, x, y, lc); };
    // 16 bytes per lane (the patch tensors are 16-byte aligned whenever P*P % 4 == 0; hipMalloc / torch allocations are)
    const size_t n4 = ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(dX)) & 15) ? 0 : n / 4;
    const float4* X4 = reinterpret_cast<const float4*>(X);
    const float4* Y4 = reinterpret_cast<const float4*>(Y);
    float4* D4 = reinterpret_cast<float4*>(dX);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 x = X4[i], y = Y4[i];
        D4[i] = make_float4(grad(x.x, y.x), grad(x.y, y.y), grad(x.z, y.z), grad(x.w, y.w));
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dX[i] = grad(X[i], Y[i]);
}

// SSIM: loss = (1/Ns) sum_w clip((1 - SSIM_w)/2, 0, 1) over the VALID 3x3 windows w.  With the window sums,
//   dSSIM_w/dx_p = alpha_w + beta_w x_p + gamma_w y_p     for every pixel p of w        (derivation in DESIGN.md 3.5)
//   gamma = (2/9) A1/(B1 B2)    beta = -(2/9) SSIM/B2
//   alpha = (2/9) [ mu_y (A2 - A1)/(B1 B2) - SSIM mu_x (1/B1 - 1/B2) ]
// and a pixel belongs to up to 9 windows.  One block = one 16x16 pixel tile of one image: the 20x20 pixel halo and
// the 18x18 window coefficients (already scaled by -dLoss/(2 Ns) and zeroed where the clip is inactive or the window
// does not exist) live in LDS.
constexpr int ST = 16;
__global__ __launch_bounds__(256) void patch_ssim_backward_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                                  const float* __restrict__ dLoss,
                                                                  float* __restrict__ dX, int P, int tiles_1d, float scale) {
    __shared__ float xs[ST + 4][ST + 5], ys[ST + 4][ST + 5];
    __shared__ float ca[ST + 2][ST + 3], cb[ST + 2][ST + 3], cg[ST + 2][ST + 3];
    const int b = blockIdx.x / (tiles_1d * tiles_1d), t = blockIdx.x - b * tiles_1d * tiles_1d;
    const int ti = (t / tiles_1d) * ST, tj = (t - (t / tiles_1d) * tiles_1d) * ST;
    const float* __restrict__ x = X + (size_t)b * P * P;
    const float* __restrict__ y = Y + (size_t)b * P * P;
    for (int e = threadIdx.x; e < (ST + 4) * (ST + 4); e += 256) {
        const int r = e / (ST + 4), c = e - r * (ST + 4);
        const int i = ti - 2 + r, j = tj - 2 + c;
        const bool in = i >= 0 && i < P && j >= 0 && j < P;
        xs[r][c] = in ? x[i * P + j] : 0.f;
        ys[r][c] = in ? y[i * P + j] : 0.f;
    }
    __syncthreads();
    const float k = -0.5f * scale * dLoss[0];
    for (int e = threadIdx.x; e < (ST + 2) * (ST + 2); e += 256) {
        const int r = e / (ST + 2), c = e - r * (ST + 2);
        const int wi = ti - 2 + r, wj = tj - 2 + c;             // top-left pixel of window (r, c)
        float a = 0.f, bt = 0.f, gm = 0.f;
        if (wi >= 0 && wi <= P - 3 && wj >= 0 && wj <= P - 3) {
            float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    const float p = xs[r + u][c + v], q = ys[r + u][c + v];
                    sx += p; sy += q; sxx += p * p; syy += q * q; sxy += p * q;
                }
            const float inv9 = 1.0f / 9.0f;
            const float mux = sx * inv9, muy = sy * inv9;
            const float sgx = sxx * inv9 - mux * mux, sgy = syy * inv9 - muy * muy, sgxy = sxy * inv9 - mux * muy;
            const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
            const float A1 = 2.f * mux * muy + C1, A2 = 2.f * sgxy + C2;
            const float B1 = mux * mux + muy * muy + C1, B2 = sgx + sgy + C2;
            const float ssim = (A1 * A2) / (B1 * B2);
            const float val = (1.f - ssim) * 0.5f;
            if (val >= 0.f && val <= 1.f) {                       // clip_by_value passes the gradient inside [0, 1]
                const float k9 = k * (2.0f / 9.0f);
                gm = k9 * A1 / (B1 * B2);
                bt = -k9 * ssim / B2;
                a = k9 * (muy * (A2 - A1) / (B1 * B2) - ssim * mux * (1.0f / B1 - 1.0f / B2));
            }
        }
        ca[r][c] = a; cb[r][c] = bt; cg[r][c] = gm;
    }
    __syncthreads();
    const int li = threadIdx.x / ST, lj = threadIdx.x - li * ST;
    const int i = ti + li, j = tj + lj;
    if (i < P && j < P) {
        float sa = 0.f, sb = 0.f, sg = 0.f;                        // windows with top-left (i-2..i, j-2..j) = coefficient rows li..li+2
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int v = 0; v < 3; ++v) { sa += ca[li + u][lj + v]; sb += cb[li + u][lj + v]; sg += cg[li + u][lj + v]; }
        dX[(size_t)b * P * P + i * P + j] = sa + sb * xs[li + 2][lj + 2] + sg * ys[li + 2][lj + 2];
    }
}

}  // namespace uh

using namespace uh;

extern "C" size_t uh_patch_losses_workspace_bytes(int B, int P) {
    if (B <= 0 || P <= 0) return 0;
    return (size_t)B * ((P * P + 256 * LPT - 1) / (256 * LPT)) * NLS * sizeof(float);
}

extern "C" int uh_patch_loss_backward(int kind, const float* pred, const float* target, const float* stats16,
                                      const float* dLoss, float* dPred, int B, int P, uh_stream_t stream) {
    if (!pred || !target || !stats16 || !dLoss || !dPred) return UH_E_NULL;
    if (B <= 0 || P < 3) return UH_E_SHAPE;
    if (kind < UH_LOSS_REC || kind > UH_LOSS_NCC) return UH_E_SHAPE;
    if ((uint64_t)B * P * P >= (1ull << 31)) return UH_E_TOO_LARGE;
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)B * P * P;
    const size_t nt = (n + 3) / 4;                                     // threads needed at 4 elements each
    const unsigned grid = (unsigned)((nt + 255) / 256 < 2048 ? (nt + 255) / 256 : 2048);
    switch (kind) {
        case UH_LOSS_REC: launch_timed(UH_K_LOSS_BWD, patch_loss_backward_kernel<UH_LOSS_REC>, dim3(grid), dim3(256), s, pred, target, stats16, dLoss, dPred, n); break;
        case UH_LOSS_L1: launch_timed(UH_K_LOSS_BWD, patch_loss_backward_kernel<UH_LOSS_L1>, dim3(grid), dim3(256), s, pred, target, stats16, dLoss, dPred, n); break;
        case UH_LOSS_L1_SMOOTH: launch_timed(UH_K_LOSS_BWD, patch_loss_backward_kernel<UH_LOSS_L1_SMOOTH>, dim3(grid), dim3(256), s, pred, target, stats16, dLoss, dPred, n); break;
        case UH_LOSS_NCC: launch_timed(UH_K_LOSS_BWD, patch_loss_backward_kernel<UH_LOSS_NCC>, dim3(grid), dim3(256), s, pred, target, stats16, dLoss, dPred, n); break;
        default: {
            const int t1 = (P + ST - 1) / ST;
            launch_timed(UH_K_LOSS_BWD, patch_ssim_backward_kernel, dim3((unsigned)B * t1 * t1), dim3(256), s, pred, target, dLoss,
                         dPred, P, t1, (float)(1.0 / ((double)B * (P - 2) * (P - 2))));
        }
    }
    return (int)hipGetLastError();
}

static int launch_losses(const float* X, const float* target, const float* h4p, const float* gt, float* out16, void* workspace,
                         int B, int P, hipStream_t s, int GC, GatherArgs ga, float* l1_out = nullptr) {
    const int bpi = (P * P + 256 * LPT - 1) / (256 * LPT);
    const LossFinish fin{h4p, gt, B * 8, out16, 1.0 / ((double)B * P * P), 1.0 / ((double)B * (P - 2) * (P - 2)), l1_out};
    const unsigned shm = GC ? (unsigned)((LPT * 256 + 2 * P + 3) * sizeof(float)) : 0u;
#define UH_LOSSES(GCV) launch_timed_shm(UH_K_LOSSES, patch_losses_kernel<GCV>, dim3((unsigned)B * bpi), dim3(256), shm, s, X, target, \
                                        (float*)workspace, P, bpi, ga)
    switch (GC) {
        case 0: UH_LOSSES(0); break;
        case 1: UH_LOSSES(1); break;
        case 2: UH_LOSSES(2); break;
        case 3: UH_LOSSES(3); break;
        default: UH_LOSSES(4); break;
    }
#undef UH_LOSSES
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    launch_timed(UH_K_LOSSES_FIN, patch_losses_finish_kernel, dim3(1), dim3(256), s, (const float*)workspace, B * bpi, fin);
    return (int)hipGetLastError();
}

extern "C" int uh_patch_losses_forward(const float* pred, const float* target, const float* h4p, const float* gt,
                                       float* out16, void* workspace, size_t workspace_bytes, int B, int P,
                                       uh_stream_t stream) {
    if (!pred || !target || !out16) return UH_E_NULL;
    if ((h4p == nullptr) != (gt == nullptr)) return UH_E_NULL;
    if (B <= 0 || P < 3) return UH_E_SHAPE;
    if ((uint64_t)B * P * P >= (1ull << 31)) return UH_E_TOO_LARGE;
    if (!workspace || workspace_bytes < uh_patch_losses_workspace_bytes(B, P)) return UH_E_WORKSPACE;
    return launch_losses(pred, target, h4p, gt, out16, workspace, B, P, (hipStream_t)stream, 0, GatherArgs{nullptr, nullptr, 0});
}

namespace uh {
int gather_patch_losses(const float* warped, const int* patch_idx, const float* target, const float* h4p, const float* gt,
                        float* pred, float* out16, float* l1_out, void* workspace, size_t workspace_bytes, int B, int H, int W,
                        int C, int P, hipStream_t stream) {
    if (!warped || !patch_idx || !target || !pred || !out16) return UH_E_NULL;
    if ((h4p == nullptr) != (gt == nullptr)) return UH_E_NULL;
    if (B <= 0 || P < 3 || H <= 0 || W <= 0) return UH_E_SHAPE;
    if (C < 1 || C > 4) return UH_E_CHANNELS;
    if ((uint64_t)B * P * P >= (1ull << 31) || (uint64_t)H * W >= (1ull << 31)) return UH_E_TOO_LARGE;
    if ((size_t)(LPT * 256 + 2 * P + 3) * sizeof(float) > 60 * 1024) return UH_E_TOO_LARGE;       // the chunk + its SSIM halo in LDS
    if (!workspace || workspace_bytes < uh_patch_losses_workspace_bytes(B, P)) return UH_E_WORKSPACE;
    return launch_losses(warped, target, h4p, gt, out16, workspace, B, P, stream, C, GatherArgs{patch_idx, pred, H * W}, l1_out);
}
}  // namespace uh

extern "C" int uh_gather_patch_losses_forward(const float* warped, const int* patch_idx, const float* target, const float* h4p,
                                              const float* gt, float* pred, float* out16, void* workspace,
                                              size_t workspace_bytes, int B, int H, int W, int C, int P, uh_stream_t stream) {
    return gather_patch_losses(warped, patch_idx, target, h4p, gt, pred, out16, nullptr, workspace, workspace_bytes, B, H, W, C, P,
                               (hipStream_t)stream);
}
