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

// flat index -> (row, col) without an integer division: q = trunc(idx * (1/W)) is within 1 of the true
// quotient for idx < 2^24; one fix-up step makes it exact.  (Large images take the '/' path.)
template <bool SMALL>
__device__ __forceinline__ void split_index(int idx, int W, float invW, int& row, int& col) {
    if constexpr (SMALL) {
        int q = (int)((float)idx * invW);
        int r = idx - q * W;
        if (r < 0) { q -= 1; r += W; }
        if (r >= W) { q += 1; r -= W; }
        row = q; col = r;
    } else {
        row = idx / W; col = idx - row * W;
    }
}

template <int C, bool WANT_GRAD, bool SMALL>
__global__ __launch_bounds__(256) void warp_patch_l1_kernel(
        const float* __restrict__ U, const float* __restrict__ theta, const float* __restrict__ I2,
        const int* __restrict__ patch_idx, float* __restrict__ pred, float* __restrict__ partial,
        int H, int W, float sx, float sy, float invW, int PP, int blocks_per_image, unsigned nblk) {
    __shared__ float red[NACC][16];
    const unsigned v = xcd_remap(blockIdx.x, nblk);
    const int b = v / blocks_per_image, chunk = v - b * blocks_per_image;
    const int lane = threadIdx.x & 63, wave = wave_id();
    Theta th;
#pragma unroll
    for (int j = 0; j < 9; ++j) th.a[j] = theta[(size_t)b * 9 + j];
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(U + (size_t)b * H * W * C, (unsigned)(H * W * C * 4));
    const SrcGeom g = make_geom<C>(W, H);
    const float halfW = (float)W * 0.5f, halfH = (float)H * 0.5f;
    float acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = 0.f;
    constexpr int BT = 2;
#pragma unroll
    for (int k0 = 0; k0 < PPT; k0 += BT) {
        Tap s[BT];
        Pix<C> Ia[BT], Ib[BT], Ic[BT], Id[BT];
        float gx[BT], gy[BT], tgt[BT];
        bool ok[BT];
        size_t e[BT];
#pragma unroll
        for (int k = 0; k < BT; ++k) {
            const int i = chunk * PB + (k0 + k) * 256 + (int)threadIdx.x;
            ok[k] = i < PP;
            e[k] = (size_t)b * PP + (ok[k] ? i : PP - 1);
            const int idx = patch_idx[e[k]];
            tgt[k] = I2[e[k]];
            int row, col;
            split_index<SMALL>(idx, W, invW, row, col);
            gx[k] = lin_at(sx, col); gy[k] = lin_at(sy, row);
            s[k] = make_tap<C, SMALL>(th, th.a[0] * gx[k], th.a[3] * gx[k], th.a[6] * gx[k], gy[k], g);
            Ia[k] = buf_load<C>(rin, s[k].oa, 0);
            Ib[k] = buf_load<C>(rin, s[k].ob, 0);
            Ic[k] = buf_load<C>(rin, s[k].oc, 0);
            Id[k] = buf_load<C>(rin, s[k].od, 0);
        }
#pragma unroll
        for (int k = 0; k < BT; ++k) {
            const float wa = s[k].ax1 * s[k].ay1, wb = s[k].ax1 * s[k].ay0;
            const float wc = s[k].ax0 * s[k].ay1, wd = s[k].ax0 * s[k].ay0;
            float gsum = blend4(wa, wb, wc, wd, Ia[k].v[0], Ib[k].v[0], Ic[k].v[0], Id[k].v[0]);
#pragma unroll
            for (int c = 1; c < C; ++c)
                gsum = gsum + blend4(wa, wb, wc, wd, Ia[k].v[c], Ib[k].v[c], Ic[k].v[c], Id[k].v[c]);
            const float p = gsum / (float)C;
            if (ok[k]) pred[e[k]] = p;
            const float m = ok[k] ? 1.f : 0.f;
            const float d = p - tgt[k];
            acc[9] += m * fabsf(d);
            if (WANT_GRAD) {
                const float sgn = d > 0.f ? m : (d < 0.f ? -m : 0.f);
                // lerp form of ay1 (Ic-Ia) + ay0 (Id-Ib) with the shared second difference u: exact 0 where the clip
                // collapsed a pair (see accumulate() in uh_warp.hip)
                float s1 = 0.f, sb = 0.f, sc = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float ddb = Id[k].v[c] - Ib[k].v[c], ddc = Id[k].v[c] - Ic[k].v[c];
                    s1 += (Ic[k].v[c] - Ia[k].v[c]) - ddb; sb += ddb; sc += ddc;
                }
                const float ex = fmaf(s[k].ay1, s1, s[k].hy * sb), ey = fmaf(s[k].ax1, s1, s[k].hx * sc);
                const float gg = sgn / (float)C;
                const float rt = s[k].rt;
                const float dxs = gg * ex * halfW * rt, dys = gg * ey * halfH * rt;
                const float dt = -(dxs * s[k].xs + dys * s[k].ys) * rt;
                acc[0] = fmaf(dxs, gx[k], acc[0]); acc[1] = fmaf(dxs, gy[k], acc[1]); acc[2] += dxs;
                acc[3] = fmaf(dys, gx[k], acc[3]); acc[4] = fmaf(dys, gy[k], acc[4]); acc[5] += dys;
                acc[6] = fmaf(dt,  gx[k], acc[6]); acc[7] = fmaf(dt,  gy[k], acc[7]); acc[8] += dt;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
        if (!WANT_GRAD && j < 9) continue;
        const float r = row16_sum(acc[j]);
        if ((lane & 15) == 0) red[j][wave * 4 + (lane >> 4)] = r;
    }
    __syncthreads();
    if (threadIdx.x < NACC) {
        const float* r = red[threadIdx.x];
        float t = 0.f;
        if (WANT_GRAD || threadIdx.x == 9) {
#pragma unroll
            for (int k = 0; k < 16; ++k) t += r[k];
        }
        partial[(size_t)v * NACC + threadIdx.x] = t;
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
    const bool sm = (uint64_t)H * W * C * 4 <= (1ull << 24);
    const float sx = lin_step(W), sy = lin_step(H), invW = 1.0f / (float)W;
#define UH_PATCH(GRAD, SM) launch_timed(UH_K_PATCH_FUSED, warp_patch_l1_kernel<C, GRAD, SM>, dim3(nblk), dim3(256), s, U, theta, \
                                       I2, idx, pred, partial, H, W, sx, sy, invW, PP, bpi, nblk)
    if (grad) { if (sm) UH_PATCH(true, true); else UH_PATCH(true, false); }
    else      { if (sm) UH_PATCH(false, true); else UH_PATCH(false, false); }
#undef UH_PATCH
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
        launch_timed(UH_K_PATCH_FIN, warp_patch_l1_finish_kernel, dim3(1), dim3(1024), s, (const float*)partial, loss,
                     dTheta, B, bpi, 1.0 / ((double)B * (double)PP));
    }
    return (int)hipGetLastError();
}
