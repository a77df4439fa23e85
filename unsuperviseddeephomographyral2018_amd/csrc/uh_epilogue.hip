// uh_epilogue.hip -- bias + ReLU (+ 2x2/2 max-pool) epilogue of the regressor's conv layers, forward and backward (gfx950).
//
// NOT part of the reference's hot path: the conv GEMMs stay stock MIOpen (north_star).  In eager PyTorch every
// `_conv2d` of /root/reference/code/homography_model.py:88-95 (conv + bias + ReLU) costs three extra full passes over
// the activation per step: bias add and ReLU in the forward, and in the backward a ReLU-mask pass plus a separate
// per-channel reduction for the bias gradient -- ~1.1 ms of a 6.8 ms step at batch 64.  These kernels do
//   forward : y <- max(y + b[c], 0) in place + ONE BIT per element (y > 0) for the backward      8.03 B/element
//   backward: g = bit ? gy : 0,  db[c] = sum g   (+ a tiny deterministic finishing reduction)     8.03 B/element
// on the NHWC (channels_last) activation, 16 bytes per lane per access.  HBM-bound.
//
// What the backward needs is kept as BITS, not as the activation (round 6; rounds 1-5 re-read y: 12 B/element in the backward,
// and the pooled variant wrote relu(y + b) back at full resolution in the forward only to read it again: 9 + 9 B per conv-output
// element).  bias+ReLU needs one bit per element; bias+ReLU+pool needs, per pooled element, which of its four window elements
// received the gradient (none when the maximum is not positive) -- nothing downstream reads the full-resolution relu(y + b): the
// next conv consumes `pooled`.  Measured on the batch-64 step: the four kernels 568 -> 428 us, +2.7 % pairs/s
// (profiles/r06_epi_ab.jsonl, three alternating pairs in one session).
// Mask layout (opaque to the caller; produced by the forward, consumed by the backward of the same shape):
//   bias+ReLU       the float4 stream is cut into chunks of 128; lane l of the wave that owns chunk c holds float4 128c + l and
//                   128c + 64 + l and stores ONE byte mask[64c + l] = bits(first) | bits(second) << 4 (64 B per wave, coalesced)
//   pool            one uint16 per (pooled pixel, channel quad): bit 4k + j = "window element k of component j gets the gradient"
//                   (k = 0..3: (2i,2j) (2i,2j+1) (2i+1,2j) (2i+1,2j+1); first maximum wins a tie, as max_pool2d; live = max > 0)
#include "uh_device.h"
#include "uh_host.h"

namespace uh {

constexpr int EPI_BLOCKS = 1024;     // upper bound of the grid; every thread keeps ONE channel quad (1024 % C == 0)

// db[c] = sum over blocks of partial[blk][c] in fixed order.  Block b of the grid owns FIN_CH consecutive channels; its 1024
// threads are (row r = tid / FIN_CH, channel j = tid % FIN_CH): a thread adds rows r, r + R, r + 2R ... (R = 1024 / FIN_CH rows, at
// most EPI_BLOCKS / R = 8 of them, all loads in flight at once), the R row sums of a channel then meet in LDS and are added in
// f64 in a fixed two-level order.  (Rounds 2-5 ran this as ONE block walking 64 - 128 rows per thread: 6.1 us per launch, eight launches
// per step; C / 8 blocks walk 8 rows each.)
constexpr int FIN_CH = 8;
__global__ __launch_bounds__(1024) void bias_grad_finish_kernel(const float* __restrict__ partial, float* __restrict__ db,
                                                                int nblk, int C) {
    __shared__ float sm[1024];
    const int cb = C < FIN_CH ? C : FIN_CH;                 // channels of this block (C = 4: one block of four)
    const int j = threadIdx.x % cb, r = threadIdx.x / cb, R = 1024 / cb;
    const int c = blockIdx.x * cb + j;
    float v[EPI_BLOCKS / (1024 / FIN_CH)];
#pragma unroll
    for (int k = 0; k < EPI_BLOCKS / (1024 / FIN_CH); ++k) {
        const int row = r + k * R;
        v[k] = row < nblk ? partial[(size_t)row * C + c] : 0.f;
    }
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < EPI_BLOCKS / (1024 / FIN_CH); ++k) a += v[k];
    sm[threadIdx.x] = a;
    __syncthreads();
    // R row sums per channel -> R / 8 group sums (f64, rows in order) -> one sum (groups in order): two short serial chains
    __shared__ double sg[1024 / 8];
    const int G = R / 8;
    if ((int)threadIdx.x < G * cb) {
        const int g = threadIdx.x / cb;
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += (double)sm[(g * 8 + k) * cb + j];
        sg[threadIdx.x] = t;
    }
    __syncthreads();
    if ((int)threadIdx.x < cb) {
        double t = 0.0;
        for (int g = 0; g < G; ++g) t += sg[g * cb + threadIdx.x];
        db[c] = (float)t;
    }
}

__device__ __forceinline__ float4 relu_bias4(float4 v, float4 b) {
    return make_float4(fmaxf(v.x + b.x, 0.f), fmaxf(v.y + b.y, 0.f), fmaxf(v.z + b.z, 0.f), fmaxf(v.w + b.w, 0.f));
}
__device__ __forceinline__ unsigned pos4(float4 v) {
    return (v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u);
}
__device__ __forceinline__ float4 keep4(float4 d, unsigned bits) {
    return make_float4((bits & 1u) ? d.x : 0.f, (bits & 2u) ? d.y : 0.f, (bits & 4u) ? d.z : 0.f, (bits & 8u) ? d.w : 0.f);
}

__global__ __launch_bounds__(256) void bias_relu_forward_kernel(float4* __restrict__ y, const float* __restrict__ bias,
                                                                     unsigned char* __restrict__ mask, size_t n4, int q /* = C/4 */) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (size_t)gridDim.x * 4;
    // q divides 256 and nwaves is a multiple of 4: the channel quads of a lane's two float4s never change along the loop
    const float4 b0 = *reinterpret_cast<const float4*>(bias + (int)((wave * 128 + lane) & (size_t)(q - 1)) * 4);
    const float4 b1 = *reinterpret_cast<const float4*>(bias + (int)((wave * 128 + 64 + lane) & (size_t)(q - 1)) * 4);
    const size_t nchunk = (n4 + 127) / 128;
    for (size_t c = wave; c < nchunk; c += nwaves) {
        const size_t i0 = c * 128 + lane, i1 = i0 + 64;
        const bool ok0 = i0 < n4, ok1 = i1 < n4;
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
        if (ok0) v0 = y[i0];
        if (ok1) v1 = y[i1];
        v0 = relu_bias4(v0, b0); v1 = relu_bias4(v1, b1);
        if (ok0) { y[i0] = v0; if (mask) mask[c * 64 + lane] = (unsigned char)(pos4(v0) | (ok1 ? pos4(v1) << 4 : 0u)); }
        if (ok1) y[i1] = v1;
    }
}

__global__ __launch_bounds__(256) void bias_relu_backward_kernel(const unsigned char* __restrict__ mask,
                                                                      const float4* __restrict__ gy, float4* __restrict__ g,
                                                                      float* __restrict__ partial, size_t n4, int q) {
    __shared__ float4 sm[512];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (size_t)gridDim.x * 4;
    const size_t nchunk = (n4 + 127) / 128;
    float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
    for (size_t c = wave; c < nchunk; c += nwaves) {
        const size_t i0 = c * 128 + lane, i1 = i0 + 64;
        const bool ok0 = i0 < n4, ok1 = i1 < n4;
        unsigned bits = 0;
        float4 d0 = make_float4(0.f, 0.f, 0.f, 0.f), d1 = d0;
        if (ok0) { bits = mask[c * 64 + lane]; d0 = gy[i0]; }
        if (ok1) d1 = gy[i1];
        d0 = keep4(d0, bits); d1 = keep4(d1, bits >> 4);
        if (ok0) g[i0] = d0;
        if (ok1) g[i1] = d1;
        acc0.x += d0.x; acc0.y += d0.y; acc0.z += d0.z; acc0.w += d0.w;
        acc1.x += d1.x; acc1.y += d1.y; acc1.z += d1.z; acc1.w += d1.w;
    }
    // virtual slot v = 128 w + 64 h + lane holds the sum of channel quad v & (q - 1) (block-invariant: 512 % q == 0)
    sm[128 * w + lane] = acc0;
    sm[128 * w + 64 + lane] = acc1;
    __syncthreads();
    if ((int)threadIdx.x < q) {
        float4 s = sm[threadIdx.x];
        for (int k = threadIdx.x + q; k < 512; k += q) { const float4 v = sm[k]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        *reinterpret_cast<float4*>(partial + (size_t)blockIdx.x * q * 4 + threadIdx.x * 4) = s;
    }
}

// which window element gets the gradient, as 4 one-hot bits (0 when the maximum is not positive)
__device__ __forceinline__ unsigned route_bits(float y0, float y1, float y2, float y3, float& m) {
    m = fmaxf(fmaxf(y0, y1), fmaxf(y2, y3));
    const unsigned sel = y0 == m ? 0x1u : (y1 == m ? 0x10u : (y2 == m ? 0x100u : 0x1000u));
    return m > 0.f ? sel : 0u;
}

__global__ __launch_bounds__(256) void bias_relu_pool_forward_kernel(const float4* __restrict__ y, const float* __restrict__ bias,
                                                                          float4* __restrict__ p, unsigned short* __restrict__ mask,
                                                                          size_t nq, int Hp, int Wp, int q) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const float4 b = *reinterpret_cast<const float4*>(bias + (int)(t % (size_t)q) * 4);
    const size_t stride = (size_t)gridDim.x * 256;
    const size_t rowq = (size_t)2 * Wp * q;
    for (size_t i = t; i < nq; i += stride) {
        const size_t pix = i / q; const int cq = (int)(i - pix * q);
        const size_t n_i = pix / Wp; const int j = (int)(pix - n_i * Wp);
        const size_t base = (n_i * 2) * rowq + (size_t)(2 * j) * q + cq;
        const float4 a0 = relu_bias4(y[base], b), a1 = relu_bias4(y[base + q], b);
        const float4 a2 = relu_bias4(y[base + rowq], b), a3 = relu_bias4(y[base + rowq + q], b);
        float4 m;
        const unsigned bits = route_bits(a0.x, a1.x, a2.x, a3.x, m.x) | route_bits(a0.y, a1.y, a2.y, a3.y, m.y) << 1 |
                              route_bits(a0.z, a1.z, a2.z, a3.z, m.z) << 2 | route_bits(a0.w, a1.w, a2.w, a3.w, m.w) << 3;
        p[i] = m;
        if (mask) mask[i] = (unsigned short)bits;
    }
}

__global__ __launch_bounds__(256) void bias_relu_pool_backward_kernel(const unsigned short* __restrict__ mask,
                                                                           const float4* __restrict__ gp, float4* __restrict__ g,
                                                                           float* __restrict__ partial, size_t nq, int Hp, int Wp,
                                                                           int q) {
    __shared__ float4 sm[256];
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    const size_t rowq = (size_t)2 * Wp * q;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = t; i < nq; i += stride) {
        const size_t pix = i / q; const int cq = (int)(i - pix * q);
        const size_t n_i = pix / Wp; const int j = (int)(pix - n_i * Wp);
        const size_t base = (n_i * 2) * rowq + (size_t)(2 * j) * q + cq;
        const unsigned bits = mask[i];
        const float4 d = gp[i];
        g[base] = keep4(d, bits); g[base + q] = keep4(d, bits >> 4);
        g[base + rowq] = keep4(d, bits >> 8); g[base + rowq + q] = keep4(d, bits >> 12);
        // exactly one window element (or none) holds d: the window's contribution to db is d where any bit of the component is set
        const float4 live = keep4(d, (bits | bits >> 4 | bits >> 8 | bits >> 12) & 15u);
        acc.x += live.x; acc.y += live.y; acc.z += live.z; acc.w += live.w;
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    if ((int)threadIdx.x < q) {
        float4 s = sm[threadIdx.x];
        for (int k = threadIdx.x + q; k < 256; k += q) { const float4 v = sm[k]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        *reinterpret_cast<float4*>(partial + (size_t)blockIdx.x * q * 4 + threadIdx.x * 4) = s;
    }
}

// grid of the chunked (two float4 per lane) kernels: one wave per 128 float4, at most EPI_BLOCKS blocks
static unsigned epi_grid2(size_t n4) {
    const size_t g = (n4 + 511) / 512;
    return (unsigned)(g < 1 ? 1 : (g > (size_t)EPI_BLOCKS ? (size_t)EPI_BLOCKS : g));
}

// grid of the pooled kernels: one thread per (pooled pixel, channel quad)
static unsigned epi_grid(size_t nq) {
    const size_t g = (nq + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > (size_t)EPI_BLOCKS ? (size_t)EPI_BLOCKS : g));
}

}  // namespace uh

using namespace uh;

static int check_epi(size_t npix, int C) {
    if (npix == 0 || C <= 0) return UH_E_SHAPE;
    if (C % 4 != 0 || 1024 % C != 0) return UH_E_CHANNELS;       // C in {4, 8, ..., 1024} dividing 1024
    return 0;
}

extern "C" size_t uh_relu_mask_bytes(size_t npix, int C) {
    if (check_epi(npix, C)) return 0;
    return ((npix * (size_t)C / 4 + 127) / 128) * 64;
}

extern "C" int uh_bias_relu_forward(float* y, const float* bias, void* mask, size_t npix, int C, uh_stream_t stream) {
    if (!y || !bias) return UH_E_NULL;                            // mask may be NULL: forward only (no backward will follow)
    if (int e = check_epi(npix, C)) return e;
    const size_t n4 = npix * (size_t)C / 4;
    launch_timed(UH_K_EPI_FWD, bias_relu_forward_kernel, dim3(epi_grid2(n4)), dim3(256), (hipStream_t)stream,
                 reinterpret_cast<float4*>(y), bias, (unsigned char*)mask, n4, C / 4);
    return (int)hipGetLastError();
}

extern "C" size_t uh_bias_relu_backward_workspace_bytes(size_t npix, int C) {
    if (check_epi(npix, C)) return 0;
    return (size_t)epi_grid2(npix * (size_t)C / 4) * C * sizeof(float);
}

extern "C" int uh_bias_relu_backward(const void* mask, const float* gy, float* g, float* dbias, void* workspace,
                                     size_t workspace_bytes, size_t npix, int C, uh_stream_t stream) {
    if (!mask || !gy || !g || !dbias) return UH_E_NULL;
    if (int e = check_epi(npix, C)) return e;
    if (!workspace || workspace_bytes < uh_bias_relu_backward_workspace_bytes(npix, C)) return UH_E_WORKSPACE;
    const size_t n4 = npix * (size_t)C / 4;
    const unsigned grid = epi_grid2(n4);
    hipStream_t s = (hipStream_t)stream;
    launch_timed(UH_K_EPI_BWD, bias_relu_backward_kernel, dim3(grid), dim3(256), s, (const unsigned char*)mask,
                 reinterpret_cast<const float4*>(gy), reinterpret_cast<float4*>(g), (float*)workspace, n4, C / 4);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(bias_grad_finish_kernel, dim3(C < FIN_CH ? 1 : C / FIN_CH), dim3(1024), 0, s, (const float*)workspace, dbias, (int)grid, C);
    return (int)hipGetLastError();
}

// ---- with the 2x2/2 max-pool ---------------------------------------------------------------------------
// y [N,H,W,C] (H, W even), p / gp [N,H/2,W/2,C].
static int check_pool(int N, int H, int W, int C) {
    if (N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return UH_E_SHAPE;
    return check_epi((size_t)N * H * W, C);
}

extern "C" size_t uh_pool_mask_bytes(int N, int H, int W, int C) {
    if (check_pool(N, H, W, C)) return 0;
    return (size_t)N * (H / 2) * (W / 2) * (C / 4) * sizeof(unsigned short);
}

extern "C" int uh_bias_relu_pool_forward(const float* y, const float* bias, float* pooled, void* mask, int N, int H, int W,
                                         int C, uh_stream_t stream) {
    if (!y || !bias || !pooled) return UH_E_NULL;                 // mask may be NULL: forward only
    if (int e = check_pool(N, H, W, C)) return e;
    const int q = C / 4;
    const size_t nq = (size_t)N * (H / 2) * (W / 2) * q;
    launch_timed(UH_K_EPI_FWD, bias_relu_pool_forward_kernel, dim3(epi_grid(nq)), dim3(256), (hipStream_t)stream,
                 reinterpret_cast<const float4*>(y), bias, reinterpret_cast<float4*>(pooled), (unsigned short*)mask, nq, H / 2,
                 W / 2, q);
    return (int)hipGetLastError();
}

extern "C" size_t uh_bias_relu_pool_backward_workspace_bytes(int N, int H, int W, int C) {
    if (check_pool(N, H, W, C)) return 0;
    return (size_t)epi_grid((size_t)N * (H / 2) * (W / 2) * (C / 4)) * C * sizeof(float);
}

extern "C" int uh_bias_relu_pool_backward(const void* mask, const float* gpooled, float* g, float* dbias, void* workspace,
                                          size_t workspace_bytes, int N, int H, int W, int C, uh_stream_t stream) {
    if (!mask || !gpooled || !g || !dbias) return UH_E_NULL;
    if (int e = check_pool(N, H, W, C)) return e;
    if (!workspace || workspace_bytes < uh_bias_relu_pool_backward_workspace_bytes(N, H, W, C)) return UH_E_WORKSPACE;
    const int q = C / 4;
    const size_t nq = (size_t)N * (H / 2) * (W / 2) * q;
    const unsigned grid = epi_grid(nq);
    hipStream_t s = (hipStream_t)stream;
    launch_timed(UH_K_EPI_BWD, bias_relu_pool_backward_kernel, dim3(grid), dim3(256), s, (const unsigned short*)mask,
                 reinterpret_cast<const float4*>(gpooled), reinterpret_cast<float4*>(g), (float*)workspace, nq, H / 2, W / 2, q);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(bias_grad_finish_kernel, dim3(C < FIN_CH ? 1 : C / FIN_CH), dim3(1024), 0, s, (const float*)workspace, dbias, (int)grid, C);
    return (int)hipGetLastError();
}
