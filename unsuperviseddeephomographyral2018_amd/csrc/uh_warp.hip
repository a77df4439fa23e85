// uh_warp.hip -- projective Spatial-Transformer bilinear warp, forward and backward (gfx950).
//
// Replaces transformer(U, theta, out_size) of /root/reference/code/utils/tf_spatial_transformer.py
// (:18-251) and TF's autodiff of it w.r.t. theta.  The TF graph materialises ~30 [B*N]-sized f32
// intermediates (~600 B/pixel); here each output pixel is produced in registers: algorithmic traffic
// is read U + write out (forward) and read dOut + read U (backward), 2*B*H*W*C*4 bytes each.
//
// Mapping: one 256-thread block = one TX x TY tile of ONE image (TX = 64 lanes along x so a wave
// reads/writes 64 consecutive NHWC pixels = 768 contiguous bytes at C=3; each wave walks ROWS
// consecutive rows so the 2x2 neighbourhoods of row r+1 re-hit the lines row r pulled into L1).
// Tiles of one image are consecutive virtual block ids and xcd_remap() keeps them on one XCD, so the
// source rows shared between vertically adjacent tiles are served by that XCD's L2, not re-fetched.
// theta is wave-uniform -> SGPRs; per-image base pointer is scalar, per-pixel offsets are 32-bit.
#include "uh_device.h"
#include "uh_host.h"

namespace uh {

constexpr int TX = 64;        // tile width  = one wavefront
constexpr int NWAVE = 4;      // waves per block

template <int C> struct Pix { float v[C]; };

template <int C>
__device__ __forceinline__ Pix<C> load_pix(const float* __restrict__ base, int off) {
    Pix<C> p;
    if constexpr (C == 3) {
        struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };
        F3 t = *reinterpret_cast<const F3*>(base + off);
        p.v[0] = t.x; p.v[1] = t.y; p.v[2] = t.z;
    } else if constexpr (C == 4) {
        float4 t = *reinterpret_cast<const float4*>(base + off);
        p.v[0] = t.x; p.v[1] = t.y; p.v[2] = t.z; p.v[3] = t.w;
    } else if constexpr (C == 2) {
        float2 t = *reinterpret_cast<const float2*>(base + off);
        p.v[0] = t.x; p.v[1] = t.y;
    } else {
        p.v[0] = base[off];
    }
    return p;
}

template <int C>
__device__ __forceinline__ void store_pix(float* __restrict__ base, int off, const Pix<C>& p) {
    if constexpr (C == 3) {
        struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };
        F3 t{p.v[0], p.v[1], p.v[2]};
        *reinterpret_cast<F3*>(base + off) = t;
    } else if constexpr (C == 4) {
        *reinterpret_cast<float4*>(base + off) = make_float4(p.v[0], p.v[1], p.v[2], p.v[3]);
    } else if constexpr (C == 2) {
        *reinterpret_cast<float2*>(base + off) = make_float2(p.v[0], p.v[1]);
    } else {
        base[off] = p.v[0];
    }
}

struct TileGeom { int tiles_x, tiles_y, tiles; };
static inline TileGeom tile_geom(int oh, int ow, int rows) {
    TileGeom g;
    g.tiles_x = (ow + TX - 1) / TX;
    g.tiles_y = (oh + NWAVE * rows - 1) / (NWAVE * rows);
    g.tiles = g.tiles_x * g.tiles_y;
    return g;
}

__device__ __forceinline__ Theta load_theta(const float* __restrict__ theta, int b) {
    Theta th;
#pragma unroll
    for (int j = 0; j < 9; ++j) th.a[j] = theta[(size_t)b * 9 + j];   // uniform address -> s_load
    return th;
}

// ------------------------------------------------------------------------------------------------
template <int C, int ROWS, bool COND>
__global__ __launch_bounds__(256) void warp_forward_kernel(
        const float* __restrict__ U, const float* __restrict__ theta, float* __restrict__ out,
        float* __restrict__ condition, int H, int W, int oh, int ow, int tiles_x, int tiles, unsigned nblk) {
    const unsigned v = xcd_remap(blockIdx.x, nblk);
    const int b = v / tiles, tile = v - b * tiles;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = tx * TX + lane;
    const int row0 = (ty * NWAVE + wave) * ROWS;
    const Theta th = load_theta(theta, b);
    const float* __restrict__ Ub = U + (size_t)b * H * W * C;
    float* __restrict__ Ob = out + (size_t)b * oh * ow * C;
    const float sx = lin_step(ow), sy = lin_step(oh);
    const float gx = lin_at(sx, col);
    float cnt = 0.f;
    if (col < ow) {
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            const int row = row0 + i;
            if (row >= oh) break;
            const Sample s = make_sample(th, gx, lin_at(sy, row), W, H);
            const int ra = s.y0 * W, rb = s.y1 * W;
            const Pix<C> Ia = load_pix<C>(Ub, (ra + s.x0) * C);
            const Pix<C> Ib = load_pix<C>(Ub, (rb + s.x0) * C);
            const Pix<C> Ic = load_pix<C>(Ub, (ra + s.x1) * C);
            const Pix<C> Id = load_pix<C>(Ub, (rb + s.x1) * C);
            Pix<C> o;
#pragma unroll
            for (int c = 0; c < C; ++c) o.v[c] = blend(s, Ia.v[c], Ib.v[c], Ic.v[c], Id.v[c]);
            store_pix<C>(Ob, (row * ow + col) * C, o);
            if (COND) cnt += (fabsf(s.t) > 1e-7f) ? 1.f : 0.f;                  // (:235)
        }
    }
    if (COND) {
        cnt = wave_sum(cnt);
        if (lane == 0) atomicAdd(condition, cnt);
    }
}

// ------------------------------------------------------------------------------------------------
// Backward w.r.t. theta (and optionally U).  Closed form of TF's autodiff (floor/clip/cast carry no
// gradient; weights use the clipped corners):
//   dx  = sum_c g_c [ ay1 (Ic - Ia) + ay0 (Id - Ib) ]      dy = sum_c g_c [ ax1 (Ib - Ia) + ax0 (Id - Ic) ]
//   dxn = dx W/2, dyn = dy H/2, dxs = dxn/t, dys = dyn/t, dt = -(dxn xs + dyn ys)/t^2
//   dTheta = [dxs; dys; dt] (3xN) . grid^T (Nx3)
// The differences (Ic - Ia) ... are exactly 0 where the clip collapsed a corner pair, so far-field
// samples contribute exactly nothing (the TF op order leaves f32 cancellation noise there).
template <int C, int ROWS, bool WANT_DU>
__global__ __launch_bounds__(256) void warp_backward_kernel(
        const float* __restrict__ U, const float* __restrict__ theta, const float* __restrict__ dOut,
        float* __restrict__ partial, float* __restrict__ dU,
        int H, int W, int oh, int ow, int tiles_x, int tiles, unsigned nblk) {
    __shared__ float red[NWAVE][9];
    const unsigned v = xcd_remap(blockIdx.x, nblk);
    const int b = v / tiles, tile = v - b * tiles;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = tx * TX + lane;
    const int row0 = (ty * NWAVE + wave) * ROWS;
    const Theta th = load_theta(theta, b);
    const float* __restrict__ Ub = U + (size_t)b * H * W * C;
    const float* __restrict__ Gb = dOut + (size_t)b * oh * ow * C;
    float* __restrict__ dUb = WANT_DU ? dU + (size_t)b * H * W * C : nullptr;
    const float sx = lin_step(ow), sy = lin_step(oh);
    const float gx = lin_at(sx, col);
    const float halfW = (float)W * 0.5f, halfH = (float)H * 0.5f;
    float acc[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[j] = 0.f;
    if (col < ow) {
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            const int row = row0 + i;
            if (row >= oh) break;
            const float gy = lin_at(sy, row);
            const Sample s = make_sample(th, gx, gy, W, H);
            const int ra = s.y0 * W, rb = s.y1 * W;
            const Pix<C> g = load_pix<C>(Gb, (row * ow + col) * C);
            const Pix<C> Ia = load_pix<C>(Ub, (ra + s.x0) * C);
            const Pix<C> Ib = load_pix<C>(Ub, (rb + s.x0) * C);
            const Pix<C> Ic = load_pix<C>(Ub, (ra + s.x1) * C);
            const Pix<C> Id = load_pix<C>(Ub, (rb + s.x1) * C);
            float dx = 0.f, dy = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float ex = fmaf(s.ay1, Ic.v[c] - Ia.v[c], s.ay0 * (Id.v[c] - Ib.v[c]));
                const float ey = fmaf(s.ax1, Ib.v[c] - Ia.v[c], s.ax0 * (Id.v[c] - Ic.v[c]));
                dx = fmaf(g.v[c], ex, dx);
                dy = fmaf(g.v[c], ey, dy);
            }
            const float rt = 1.0f / s.t;
            const float dxs = dx * halfW * rt, dys = dy * halfH * rt;
            const float dt = -(dxs * s.xs + dys * s.ys) * rt;
            acc[0] = fmaf(dxs, gx, acc[0]); acc[1] = fmaf(dxs, gy, acc[1]); acc[2] += dxs;
            acc[3] = fmaf(dys, gx, acc[3]); acc[4] = fmaf(dys, gy, acc[4]); acc[5] += dys;
            acc[6] = fmaf(dt,  gx, acc[6]); acc[7] = fmaf(dt,  gy, acc[7]); acc[8] += dt;
            if (WANT_DU) {
                const float wa = s.ax1 * s.ay1, wb = s.ax1 * s.ay0, wc = s.ax0 * s.ay1, wd = s.ax0 * s.ay0;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    atomicAdd(dUb + (ra + s.x0) * C + c, wa * g.v[c]);
                    atomicAdd(dUb + (rb + s.x0) * C + c, wb * g.v[c]);
                    atomicAdd(dUb + (ra + s.x1) * C + c, wc * g.v[c]);
                    atomicAdd(dUb + (rb + s.x1) * C + c, wd * g.v[c]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const float r = wave_sum(acc[j]);
        if (lane == 0) red[wave][j] = r;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        const int j = threadIdx.x;
        partial[(size_t)v * 9 + j] = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
    }
}

// Deterministic finish: dTheta[b][j] = sum over the image's tiles, accumulated in f64, fixed order.
// One wave per image.  The image's partials are `tiles*9` contiguous floats; lane l < 63 walks them
// with stride 63 (= 7*9), so its accumulator index j = l % 9 never changes and every load instruction
// is coalesced and independent of the others; the 7 lanes that share a j then meet in LDS.
__global__ __launch_bounds__(256) void warp_backward_finish_kernel(const float* __restrict__ partial,
                                                                   float* __restrict__ dTheta, int tiles, int B) {
    __shared__ double red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    double a = 0.0;
    if (b < B && lane < 63) {
        const float* p = partial + (size_t)b * tiles * 9;
        const int n = tiles * 9;
#pragma unroll 4
        for (int e = lane; e < n; e += 63) a += (double)p[e];
    }
    red[wave][lane] = a;
    __syncthreads();
    if (b < B && lane < 9) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 7; ++k) t += red[wave][lane + 9 * k];
        dTheta[(size_t)b * 9 + lane] = (float)t;
    }
}

}  // namespace uh

// ---- C ABI ------------------------------------------------------------------------------------
using namespace uh;

#ifndef UH_WARP_ROWS
#define UH_WARP_ROWS 4
#endif

static int check_warp_args(int B, int H, int W, int C, int oh, int ow) {
    if (B <= 0 || H <= 0 || W <= 0 || oh <= 0 || ow <= 0) return UH_E_SHAPE;
    if (C < 1 || C > 4) return UH_E_CHANNELS;
    if ((uint64_t)H * W * C * 4 >= (1ull << 31) || (uint64_t)oh * ow * C * 4 >= (1ull << 31)) return UH_E_TOO_LARGE;
    const TileGeom g = tile_geom(oh, ow, UH_WARP_ROWS);
    if ((uint64_t)B * g.tiles >= (1ull << 31)) return UH_E_TOO_LARGE;
    return 0;
}

template <int C>
static void launch_fwd(const float* U, const float* theta, float* out, float* condition, int B, int H, int W,
                       int oh, int ow, hipStream_t s) {
    constexpr int R = UH_WARP_ROWS;
    const TileGeom g = tile_geom(oh, ow, R);
    const unsigned nblk = (unsigned)B * g.tiles;
    if (condition)
        hipLaunchKernelGGL((warp_forward_kernel<C, R, true>), dim3(nblk), dim3(256), 0, s, U, theta, out,
                           condition, H, W, oh, ow, g.tiles_x, g.tiles, nblk);
    else
        hipLaunchKernelGGL((warp_forward_kernel<C, R, false>), dim3(nblk), dim3(256), 0, s, U, theta, out,
                           condition, H, W, oh, ow, g.tiles_x, g.tiles, nblk);
}

extern "C" int uh_warp_forward(const float* U, const float* theta, float* out, float* condition, int B,
                               int H, int W, int C, int oh, int ow, uh_stream_t stream) {
    if (!U || !theta || !out) return UH_E_NULL;
    if (int e = check_warp_args(B, H, W, C, oh, ow)) return e;
    hipStream_t s = (hipStream_t)stream;
    if (condition) {
        hipError_t e = hipMemsetAsync(condition, 0, sizeof(float), s);
        if (e != hipSuccess) return (int)e;
    }
    ProfScope prof(UH_K_WARP_FWD, s);
    switch (C) {
        case 1: launch_fwd<1>(U, theta, out, condition, B, H, W, oh, ow, s); break;
        case 2: launch_fwd<2>(U, theta, out, condition, B, H, W, oh, ow, s); break;
        case 3: launch_fwd<3>(U, theta, out, condition, B, H, W, oh, ow, s); break;
        default: launch_fwd<4>(U, theta, out, condition, B, H, W, oh, ow, s); break;
    }
    return (int)hipGetLastError();
}

extern "C" size_t uh_warp_backward_workspace_bytes(int B, int H, int W, int C, int oh, int ow) {
    if (check_warp_args(B, H, W, C, oh, ow)) return 0;
    const TileGeom g = tile_geom(oh, ow, UH_WARP_ROWS);
    return (size_t)B * g.tiles * 9 * sizeof(float);
}

template <int C>
static void launch_bwd(const float* U, const float* theta, const float* dOut, float* partial, float* dU, int B,
                       int H, int W, int oh, int ow, hipStream_t s) {
    constexpr int R = UH_WARP_ROWS;
    const TileGeom g = tile_geom(oh, ow, R);
    const unsigned nblk = (unsigned)B * g.tiles;
    if (dU)
        hipLaunchKernelGGL((warp_backward_kernel<C, R, true>), dim3(nblk), dim3(256), 0, s, U, theta, dOut,
                           partial, dU, H, W, oh, ow, g.tiles_x, g.tiles, nblk);
    else
        hipLaunchKernelGGL((warp_backward_kernel<C, R, false>), dim3(nblk), dim3(256), 0, s, U, theta, dOut,
                           partial, dU, H, W, oh, ow, g.tiles_x, g.tiles, nblk);
}

extern "C" int uh_warp_backward(const float* U, const float* theta, const float* dOut, float* dTheta,
                                float* dU, void* workspace, size_t workspace_bytes, int B, int H, int W,
                                int C, int oh, int ow, uh_stream_t stream) {
    if (!U || !theta || !dOut || !dTheta) return UH_E_NULL;
    if (int e = check_warp_args(B, H, W, C, oh, ow)) return e;
    if (!workspace || workspace_bytes < uh_warp_backward_workspace_bytes(B, H, W, C, oh, ow)) return UH_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (dU) {
        hipError_t e = hipMemsetAsync(dU, 0, (size_t)B * H * W * C * sizeof(float), s);
        if (e != hipSuccess) return (int)e;
    }
    float* partial = (float*)workspace;
    {
        ProfScope prof(UH_K_WARP_BWD, s);
        switch (C) {
            case 1: launch_bwd<1>(U, theta, dOut, partial, dU, B, H, W, oh, ow, s); break;
            case 2: launch_bwd<2>(U, theta, dOut, partial, dU, B, H, W, oh, ow, s); break;
            case 3: launch_bwd<3>(U, theta, dOut, partial, dU, B, H, W, oh, ow, s); break;
            default: launch_bwd<4>(U, theta, dOut, partial, dU, B, H, W, oh, ow, s); break;
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    {
        ProfScope prof(UH_K_WARP_BWD_FIN, s);
        const TileGeom g = tile_geom(oh, ow, UH_WARP_ROWS);
        hipLaunchKernelGGL(warp_backward_finish_kernel, dim3((B + 3) / 4), dim3(256), 0, s, partial, dTheta, g.tiles, B);
    }
    return (int)hipGetLastError();
}
