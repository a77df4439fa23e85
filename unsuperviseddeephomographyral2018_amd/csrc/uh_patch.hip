// uh_patch.hip -- fused photometric patch path:
//   sample(theta) -> gray -> pred_I2 -> |pred - I2| -> mean      and      d mean / d theta
// restricted to the P x P loss patch, in ONE pass that never materialises the warped frame.
//
// It is the composition of transformer() (/root/reference/code/utils/tf_spatial_transformer.py:18),
// reduce_mean(axis=3) + gather (homography_model.py:263-269) and the l1 branch of build_losses()
// (homography_model.py:328), and of their autodiff.  Forward values use the exact op order of the
// un-fused kernels (same make_sample / blend, same sequential channel sum), so pred is bit-identical
// to uh_warp_forward -> uh_gray_patch_forward.  Because d|.|/dpred = sign(.), the backward needs no
// second pass: each sample contributes sign/(C) * [...] to dTheta, scaled by 1/(B*PP) at the end.
//
// Algorithmic bytes per pair: PP * (4 neighbours * C * 4 + 4 (I2) + 4 (idx) + 4 (pred)).
#include "uh_device.h"
#include "uh_host.h"

namespace uh {

constexpr int PPT = 4;             // patch pixels per thread (amortises the block reduction)
constexpr int PB = 256 * PPT;      // patch pixels per block
constexpr int NACC = 10;           // 9 dTheta sums + 1 |diff| sum

template <int C, bool WANT_GRAD>
__global__ __launch_bounds__(256) void warp_patch_l1_kernel(
        const float* __restrict__ U, const float* __restrict__ theta, const float* __restrict__ I2,
        const int* __restrict__ patch_idx, float* __restrict__ pred, float* __restrict__ partial,
        int H, int W, int PP, int blocks_per_image, unsigned nblk) {
    __shared__ float red[4][NACC];
    const unsigned v = xcd_remap(blockIdx.x, nblk);
    const int b = v / blocks_per_image, chunk = v - b * blocks_per_image;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    Theta th;
#pragma unroll
    for (int j = 0; j < 9; ++j) th.a[j] = theta[(size_t)b * 9 + j];
    const float* __restrict__ Ub = U + (size_t)b * H * W * C;
    const float sx = lin_step(W), sy = lin_step(H);
    const float halfW = (float)W * 0.5f, halfH = (float)H * 0.5f;
    float acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = 0.f;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int i = chunk * PB + k * 256 + threadIdx.x;
        if (i >= PP) break;
        const size_t e = (size_t)b * PP + i;
        const int idx = patch_idx[e];
        const int row = idx / W, col = idx - row * W;
        const float gx = lin_at(sx, col), gy = lin_at(sy, row);
        const Sample s = make_sample(th, gx, gy, W, H);
        const int ra = s.y0 * W, rb = s.y1 * W;
        const float* pa = Ub + (ra + s.x0) * C; const float* pb = Ub + (rb + s.x0) * C;
        const float* pc = Ub + (ra + s.x1) * C; const float* pd = Ub + (rb + s.x1) * C;
        float Ia[C], Ib[C], Ic[C], Id[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { Ia[c] = pa[c]; Ib[c] = pb[c]; Ic[c] = pc[c]; Id[c] = pd[c]; }
        float gsum = blend(s, Ia[0], Ib[0], Ic[0], Id[0]);
#pragma unroll
        for (int c = 1; c < C; ++c) gsum = gsum + blend(s, Ia[c], Ib[c], Ic[c], Id[c]);
        const float p = gsum / (float)C;
        pred[e] = p;
        const float d = p - I2[e];
        acc[9] += fabsf(d);
        if (WANT_GRAD) {
            const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            float ex = 0.f, ey = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                ex += fmaf(s.ay1, Ic[c] - Ia[c], s.ay0 * (Id[c] - Ib[c]));
                ey += fmaf(s.ax1, Ib[c] - Ia[c], s.ax0 * (Id[c] - Ic[c]));
            }
            const float g = sgn / (float)C;
            const float rt = 1.0f / s.t;
            const float dxs = g * ex * halfW * rt, dys = g * ey * halfH * rt;
            const float dt = -(dxs * s.xs + dys * s.ys) * rt;
            acc[0] = fmaf(dxs, gx, acc[0]); acc[1] = fmaf(dxs, gy, acc[1]); acc[2] += dxs;
            acc[3] = fmaf(dys, gx, acc[3]); acc[4] = fmaf(dys, gy, acc[4]); acc[5] += dys;
            acc[6] = fmaf(dt,  gx, acc[6]); acc[7] = fmaf(dt,  gy, acc[7]); acc[8] += dt;
        }
    }
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
        const float r = wave_sum(acc[j]);
        if (lane == 0) red[wave][j] = r;
    }
    __syncthreads();
    if (threadIdx.x < NACC) {
        const int j = threadIdx.x;
        partial[(size_t)v * NACC + j] = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
    }
}

// One block finishes everything deterministically.  Thread q owns one (image b, accumulator j) pair
// and adds that image's per-block partials in f64 in fixed order (adjacent threads read adjacent
// floats); the |diff| sums (j == 9) then meet in LDS for the scalar loss.
__global__ __launch_bounds__(1024) void warp_patch_l1_finish_kernel(const float* __restrict__ partial,
                                                                    float* __restrict__ loss,
                                                                    float* __restrict__ dTheta, int B,
                                                                    int blocks_per_image, double inv_n) {
    __shared__ double lsum[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double lacc = 0.0;
    for (int q = threadIdx.x; q < B * NACC; q += 1024) {
        const int b = q / NACC, j = q - b * NACC;
        const float* p = partial + (size_t)b * blocks_per_image * NACC + j;
        double a = 0.0;
#pragma unroll 8
        for (int t = 0; t < blocks_per_image; ++t) a += (double)p[(size_t)t * NACC];
        if (j < 9) { if (dTheta) dTheta[(size_t)b * 9 + j] = (float)(a * inv_n); }
        else lacc += a;
    }
    lacc = wave_sum(lacc);
    if (lane == 0) lsum[wave] = lacc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += lsum[w];
        loss[0] = (float)(t * inv_n);
    }
}

}  // namespace uh

using namespace uh;

extern "C" size_t uh_warp_patch_l1_workspace_bytes(int B, int PP) {
    if (B <= 0 || PP <= 0) return 0;
    return (size_t)B * ((PP + PB - 1) / PB) * NACC * sizeof(float);
}

template <int C>
static void launch_patch(const float* U, const float* theta, const float* I2, const int* idx, float* pred,
                         float* partial, bool grad, int H, int W, int PP, int bpi, unsigned nblk, hipStream_t s) {
    if (grad)
        hipLaunchKernelGGL((warp_patch_l1_kernel<C, true>), dim3(nblk), dim3(256), 0, s, U, theta, I2, idx, pred,
                           partial, H, W, PP, bpi, nblk);
    else
        hipLaunchKernelGGL((warp_patch_l1_kernel<C, false>), dim3(nblk), dim3(256), 0, s, U, theta, I2, idx, pred,
                           partial, H, W, PP, bpi, nblk);
}

extern "C" int uh_warp_patch_l1_fwdbwd(const float* U, const float* theta, const float* I2, const int* patch_idx,
                                       float* pred, float* loss, float* dTheta, void* workspace,
                                       size_t workspace_bytes, int B, int H, int W, int C, int PP,
                                       uh_stream_t stream) {
    if (!U || !theta || !I2 || !patch_idx || !pred || !loss) return UH_E_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || PP <= 0) return UH_E_SHAPE;
    if (C < 1 || C > 4) return UH_E_CHANNELS;
    if ((uint64_t)H * W * C * 4 >= (1ull << 31)) return UH_E_TOO_LARGE;
    if (!workspace || workspace_bytes < uh_warp_patch_l1_workspace_bytes(B, PP)) return UH_E_WORKSPACE;
    const int bpi = (PP + PB - 1) / PB;
    if ((uint64_t)B * bpi >= (1ull << 31)) return UH_E_TOO_LARGE;
    const unsigned nblk = (unsigned)B * bpi;
    hipStream_t s = (hipStream_t)stream;
    float* partial = (float*)workspace;
    {
        ProfScope prof(UH_K_PATCH_FUSED, s);
        const bool grad = dTheta != nullptr;
        switch (C) {
            case 1: launch_patch<1>(U, theta, I2, patch_idx, pred, partial, grad, H, W, PP, bpi, nblk, s); break;
            case 2: launch_patch<2>(U, theta, I2, patch_idx, pred, partial, grad, H, W, PP, bpi, nblk, s); break;
            case 3: launch_patch<3>(U, theta, I2, patch_idx, pred, partial, grad, H, W, PP, bpi, nblk, s); break;
            default: launch_patch<4>(U, theta, I2, patch_idx, pred, partial, grad, H, W, PP, bpi, nblk, s); break;
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    {
        ProfScope prof(UH_K_PATCH_FIN, s);
        hipLaunchKernelGGL(warp_patch_l1_finish_kernel, dim3(1), dim3(1024), 0, s, (const float*)partial, loss,
                           dTheta, B, bpi, 1.0 / ((double)B * (double)PP));
    }
    return (int)hipGetLastError();
}
