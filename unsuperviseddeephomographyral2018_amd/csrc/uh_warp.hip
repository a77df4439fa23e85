// uh_warp.hip -- projective Spatial-Transformer bilinear warp, forward and backward (gfx950).
//
// Replaces transformer(U, theta, out_size) of /root/reference/code/utils/tf_spatial_transformer.py
// (:18-251) and TF's autodiff of it w.r.t. theta.  The TF graph materialises ~30 [B*N]-sized f32
// intermediates (~600 B/pixel); here each output pixel is produced in registers: algorithmic traffic
// is read U + write out (forward) and read dOut + read U (backward), 2*B*H*W*C*4 bytes each.
//
// Design (what the measurements forced -- DESIGN.md "Warp kernels"):
//   * The first version gathered the 2x2 neighbourhood straight from global memory: 4 per-lane 12-byte
//     loads per pixel.  HBM traffic was already minimal (FETCH/WRITE counters = unique footprint) but the
//     kernel ran at 45-60 % of a same-bytes copy: the vector-memory pipe (TA -> TCP -> TD) processes a
//     >4-byte-per-lane access at 4 lanes/clk, so 4 gathers + 1 store per pixel kept TA/TCP/TD busy ~100 %
//     of the time (TCP_GATE_EN / TD_TD_BUSY = kernel duration; SQ_WAIT_INST_ANY = 69 % of wave cycles,
//     i.e. waves waiting to ISSUE the next VMEM instruction) while HBM idled.
//   * Now every wave owns a TW x (WY*STEPS) output tile (default 16 x 16: a square footprint keeps the taps of
//     a rotated tile on few source rows).  FORWARD: it computes the sample coordinates of its 4 pixels/lane,
//     reduces the bounding rectangle of all taps with DPP min/max (exact: no convexity assumption, no safety
//     margin), pulls that source rectangle into a wave-private 4 KiB slice of LDS with wide (16 B/lane)
//     row-contiguous loads -- each source byte crosses the TA once instead of up to 4 times -- and takes the
//     2x2 neighbourhoods from LDS (ds_read2_b32 + ds_read_b32 per tap).  No block barrier is involved: the
//     producer and the consumers of a slice are lanes of the same wave.
//   * A wave whose rectangle does not fit its LDS slice (far field of a strong perspective, where the
//     footprint of 256 output pixels can be the whole frame) falls back to the direct gather; the choice is
//     wave-uniform and both paths give bit-identical results.
//   * BACKWARD: direct gather with all 16 tap loads of a lane in flight (UH_WARP_STAGE_BWD 0).  Staging was
//     measured 15-30 % slower there in every variant tried.
//   * blockIdx -> (image, tile) keeps the tiles of one image on one XCD (xcd_remap) so that the overlap of
//     neighbouring rectangles is served by that XCD's L2.
#include "uh_device.h"
#include "uh_host.h"

namespace uh {

#ifndef UH_WARP_TW
#define UH_WARP_TW 16
#endif
#ifndef UH_WARP_STEPS
#define UH_WARP_STEPS 4
#endif
#ifndef UH_WARP_LDS_PER_WAVE
#define UH_WARP_LDS_PER_WAVE 4096
#endif
#ifndef UH_WARP_STAGE_FWD
#define UH_WARP_STAGE_FWD 1       // forward: stage through LDS when the rectangle fits (else gather)
#endif
#ifndef UH_WARP_STAGE_BWD
#define UH_WARP_STAGE_BWD 0       // backward: direct gather only (staging measured slower: profiles/r01g_*)
#endif
constexpr int TW = UH_WARP_TW;          // wave tile width (pixels)
constexpr int WY = 64 / TW;             // rows a wave covers per step
#ifndef UH_WARP_STORE_AUX
#define UH_WARP_STORE_AUX 0       // cache policy of the forward's `out` stores: 0 default, 2 = nt (non-temporal)
#endif
#ifndef UH_WARP_GLOAD_AUX
#define UH_WARP_GLOAD_AUX 0       // cache policy of the backward's dOut loads
#endif
#ifndef UH_WARP_STEPS_BWD
#define UH_WARP_STEPS_BWD UH_WARP_STEPS
#endif
constexpr int STEPS = UH_WARP_STEPS;    // pixels per lane (forward)
constexpr int TH = WY * STEPS;          // wave tile height (forward)
constexpr int STEPS_B = UH_WARP_STEPS_BWD;   // pixels per lane (backward)
constexpr int TH_B = WY * STEPS_B;           // wave tile height (backward)
constexpr int NWAVE = 4;                // waves per block, side by side in x: block tile = (4*TW) x TH
constexpr int LDS_PER_WAVE = UH_WARP_LDS_PER_WAVE;
constexpr bool STAGE_FWD = UH_WARP_STAGE_FWD != 0, STAGE_BWD = UH_WARP_STAGE_BWD != 0;
static_assert(TW * WY == 64, "UH_WARP_TW must divide 64");
static_assert(LDS_PER_WAVE % 1024 == 0 && LDS_PER_WAVE <= 16384, "LDS slice: multiple of 1 KiB, <= 16 KiB");

struct TileGeom { int tiles_x, tiles_y, tiles; };
static inline TileGeom tile_geom(int oh, int ow, int th = TH) {
    TileGeom g;
    g.tiles_x = (ow + NWAVE * TW - 1) / (NWAVE * TW);
    g.tiles_y = (oh + th - 1) / th;
    g.tiles = g.tiles_x * g.tiles_y;
    return g;
}

__device__ __forceinline__ Theta load_theta(const float* __restrict__ theta, int b) {
    Theta th;
#pragma unroll
    for (int j = 0; j < 9; ++j) th.a[j] = theta[(size_t)b * 9 + j];   // uniform address -> s_load
    return th;
}

// Rectangle of all taps of this wave (exact: min/max over the clipped corners of every pixel of the wave), the
// LDS layout chosen for it, and whether it fits the wave's LDS slice.
//   LDS layout: row r of the rectangle at r*pitch; pitch = cw*16 bytes with cw = 16, 32 or 64 chunks of 16 bytes
//   (the smallest that covers the rectangle's row), so that lane -> (row, chunk) is a shift and a mask.
struct Rect {
    int bx0, by0;        // top-left source pixel of the rectangle
    int cpr;             // 16-byte chunks that actually carry data in one row
    int sh;              // log2(cw)
    int iters;           // staging iterations, each moves 64 >> sh ... rows
    float pitch, nbase;  // pitch in bytes; nbase = -(by0*pitch + bx0*pixB)
    bool fits;
};
template <int C, int N>
__device__ __forceinline__ Rect wave_rect(const Coord (&c)[N]) {
    // the clipped corners are exact non-negative integers held in f32: their bit patterns order like integers
    int mnx = __float_as_int(c[0].x0f), mxx = __float_as_int(c[0].x1f);
    int mny = __float_as_int(c[0].y0f), mxy = __float_as_int(c[0].y1f);
#pragma unroll
    for (int k = 1; k < N; ++k) {
        mnx = min(mnx, __float_as_int(c[k].x0f)); mxx = max(mxx, __float_as_int(c[k].x1f));
        mny = min(mny, __float_as_int(c[k].y0f)); mxy = max(mxy, __float_as_int(c[k].y1f));
    }
    const float fx0 = __int_as_float(wave_min_nonneg(mnx)), fx1 = __int_as_float(wave_max_nonneg(mxx));
    const float fy0 = __int_as_float(wave_min_nonneg(mny)), fy1 = __int_as_float(wave_max_nonneg(mxy));
    Rect r;
    r.bx0 = (int)fx0; r.by0 = (int)fy0;
    const int rw = (int)fx1 - r.bx0 + 1, rh = (int)fy1 - r.by0 + 1;
    r.cpr = (rw * C * 4 + 15) >> 4;
    r.sh = r.cpr <= 16 ? 4 : (r.cpr <= 32 ? 5 : 6);
    const int rpi = 64 >> r.sh;                           // rows per staging iteration
    r.iters = (rh + rpi - 1) / rpi;
    const int pitch = 16 << r.sh;
    r.fits = r.cpr <= 64 && r.iters * rpi * pitch <= LDS_PER_WAVE;
    r.pitch = (float)pitch;
    r.nbase = -((float)r.by0 * r.pitch + (float)(r.bx0 * C * 4));
    return r;
}

// Stage the rectangle into this wave's LDS slice with 16-byte-per-lane loads: lane = (row r, chunk c) with
// r = lane >> sh, c = lane & (cw-1); iteration i moves rows i*rpi + r.  The global address of chunk c of row r,
// ((by0+r)*W + bx0)*C*4 + c*16, is only 4-byte aligned, which costs nothing on global loads (tools/ubench:
// misaligned b128 == aligned b128).  Chunks c >= cpr (padding of the power-of-two pitch) and rows past the end
// of the image get an out-of-range offset: the buffer unit returns zeros without touching memory; the LDS
// write needs no mask because `fits` accounted for the padded footprint.  Per iteration: one VALU add (global
// offset), one buffer_load_dwordx4, one ds_write_b128 (LDS step folded into the instruction offset).
template <int C>
__device__ __forceinline__ void stage_rect(__amdgpu_buffer_rsrc_t rin, unsigned char* lds, int lane, const Rect& r,
                                           int rowBi) {
    const int rr = lane >> r.sh, cc = lane & ((1 << r.sh) - 1);
    const unsigned goff = cc < r.cpr ? (unsigned)(r.by0 + rr) * (unsigned)rowBi + (unsigned)(r.bx0 * C * 4 + cc * 16)
                                     : 0x80000000u;
    unsigned char* lp = lds + lane * 16;                   // == rr*pitch + cc*16
    const unsigned gstep = (unsigned)(64 >> r.sh) * (unsigned)rowBi;
    // the row step goes into the VGPR offset (not soffset): the buffer unit range-checks the VGPR offset, which is
    // what turns rows past the end of the image into zeros instead of reads past the allocation
    int i = 0;
    for (; i + 4 <= r.iters; i += 4) {                      // wave-uniform trip counts
        u32x4_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rin, goff + (unsigned)(i + u) * gstep, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<u32x4_t*>(lp + (size_t)(i + u) * 1024) = v[u];
    }
    for (; i < r.iters; ++i) {
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rin, goff + (unsigned)i * gstep, 0, 0);
        *reinterpret_cast<u32x4_t*>(lp + (size_t)i * 1024) = v;
    }
}

// ------------------------------------------------------------------------------------------------
template <int C, bool COND, bool SMALL>
__global__ __launch_bounds__(256) void warp_forward_kernel(
        const float* __restrict__ U, const float* __restrict__ theta, float* __restrict__ out,
        float* __restrict__ condition, int H, int W, int oh, int ow, float sx, float sy, int tiles_x, int tiles,
        unsigned nblk) {
    __shared__ __attribute__((aligned(16))) unsigned char lds_all[STAGE_FWD ? NWAVE * LDS_PER_WAVE : 16];
    const int lane = threadIdx.x & 63, wave = wave_id();
    unsigned char* lds = lds_all + (STAGE_FWD ? wave * LDS_PER_WAVE : 0);
    const unsigned v = xcd_remap(blockIdx.x, nblk);
    const int b = v / tiles, tile = v - b * tiles;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int col = (tx * NWAVE + wave) * TW + (lane & (TW - 1));
    const int row0 = ty * TH + (lane / TW);                              // this lane's first row
    if ((tx * NWAVE + wave) * TW >= ow) return;                          // whole wave outside (wave-uniform)
    const Theta th = load_theta(theta, b);
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(U + (size_t)b * H * W * C, (unsigned)(H * W * C * 4));
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(out + (size_t)b * oh * ow * C, (unsigned)(oh * ow * C * 4));
    const SrcGeom g = make_geom<C>(W, H);
    const float gx = lin_at(sx, col);           // lanes past the right/bottom edge compute on; their store is dropped
    const float A0 = th.a[0] * gx, A3 = th.a[3] * gx, A6 = th.a[6] * gx;
    const bool col_ok = col < ow;
    const unsigned orow = (unsigned)(ow * C * 4);

    const float rowf0 = (float)row0;                                     // (float)(row0 + k*WY) == rowf0 + k*WY exactly
    Coord c[STEPS];
#pragma unroll
    for (int k = 0; k < STEPS; ++k) c[k] = make_coord(th, A0, A3, A6, -1.0f + sy * (rowf0 + (float)(k * WY)), g);
    Rect r;
    r.fits = false;
    if constexpr (STAGE_FWD) r = wave_rect<C, STEPS>(c);

    Pix<C> Ia[STEPS], Ib[STEPS], Ic[STEPS], Id[STEPS];
    if (STAGE_FWD && r.fits) {                                          // wave-uniform
        stage_rect<C>(rin, lds, lane, r, g.rowBi);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int k = 0; k < STEPS; ++k) {
            const TapOff o = staged_offsets(c[k], r.pitch, g.pixB, r.nbase);
            Ia[k] = lds_load<C>(lds, o.oa); Ib[k] = lds_load<C>(lds, o.ob);
            Ic[k] = lds_load<C>(lds, o.oc); Id[k] = lds_load<C>(lds, o.od);
        }
    } else {
#pragma unroll
        for (int k = 0; k < STEPS; ++k) {
            const TapOff o = global_offsets<SMALL>(c[k], g);
            Ia[k] = buf_load<C>(rin, o.oa, 0); Ib[k] = buf_load<C>(rin, o.ob, 0);
            Ic[k] = buf_load<C>(rin, o.oc, 0); Id[k] = buf_load<C>(rin, o.od, 0);
        }
    }
    float cnt = 0.f;
    // Store offsets: row*orow + col*pixB, advanced by WY rows per step.  A row >= oh gives an offset >= the
    // buffer's num_records (= oh*orow), which makes the buffer unit drop the store; lanes right of the image
    // start from 2^31 and stay out of range (an image is < 2^31 bytes, check_warp_args).
    unsigned voff = col_ok ? (unsigned)row0 * orow + (unsigned)col * (C * 4) : 0x80000000u;
#pragma unroll
    for (int k = 0; k < STEPS; ++k) {
        const float wa = c[k].ax1 * c[k].ay1, wb = c[k].ax1 * c[k].ay0;                         // (:134-137)
        const float wc = c[k].ax0 * c[k].ay1, wd = c[k].ax0 * c[k].ay0;
        Pix<C> o;
#pragma unroll
        for (int ch = 0; ch < C; ++ch) o.v[ch] = blend4(wa, wb, wc, wd, Ia[k].v[ch], Ib[k].v[ch], Ic[k].v[ch], Id[k].v[ch]);
#ifdef UH_DBG_NO_STORE            // developer A/B switch only: keeps the math alive, never stores
        buf_store<C, UH_WARP_STORE_AUX>(rout, o.v[0] != 12345.678f ? 0x80000000u : voff, 0, o);
#else
        buf_store<C, UH_WARP_STORE_AUX>(rout, voff, 0, o);
#endif
        if (COND) cnt += (col_ok && row0 + k * WY < oh && fabsf(c[k].t) > 1e-7f) ? 1.f : 0.f;   // (:235)
        voff += (unsigned)WY * orow;
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
// 1/t is the refined reciprocal the sampling already produced (<= 1 ulp; the gradient is checked
// against the f64 closed form, not bit-for-bit).
template <int C, bool WANT_DU, bool SMALL>
__global__ __launch_bounds__(256) void warp_backward_kernel(
        const float* __restrict__ U, const float* __restrict__ theta, const float* __restrict__ dOut,
        float* __restrict__ partial, float* __restrict__ dU,
        int H, int W, int oh, int ow, float sx, float sy, int tiles_x, int tiles, unsigned nblk) {
    __shared__ __attribute__((aligned(16))) unsigned char lds_all[STAGE_BWD ? NWAVE * LDS_PER_WAVE : 16];
    __shared__ float red[9][NWAVE * 4];
    const int lane = threadIdx.x & 63, wave = wave_id();
    const unsigned v = xcd_remap(blockIdx.x, nblk);
    const int b = v / tiles, tile = v - b * tiles;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int col = (tx * NWAVE + wave) * TW + (lane & (TW - 1));
    const int row0 = ty * TH_B + (lane / TW);                              // this lane's first row
    float acc[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[j] = 0.f;
    if ((tx * NWAVE + wave) * TW < ow) {                                 // wave-uniform; else the wave adds 0
        unsigned char* lds = lds_all + (STAGE_BWD ? wave * LDS_PER_WAVE : 0);
        const Theta th = load_theta(theta, b);
        const __amdgpu_buffer_rsrc_t rin = make_rsrc(U + (size_t)b * H * W * C, (unsigned)(H * W * C * 4));
        const __amdgpu_buffer_rsrc_t rg = make_rsrc(dOut + (size_t)b * oh * ow * C, (unsigned)(oh * ow * C * 4));
        float* __restrict__ dUb = WANT_DU ? dU + (size_t)b * H * W * C : nullptr;
        const SrcGeom g = make_geom<C>(W, H);
        const float gx = lin_at(sx, col);
        const float A0 = th.a[0] * gx, A3 = th.a[3] * gx, A6 = th.a[6] * gx;
        const float halfW = (float)W * 0.5f, halfH = (float)H * 0.5f;
        const bool col_ok = col < ow;
        const unsigned orow = (unsigned)(ow * C * 4);

        const float rowf0 = (float)row0;                                 // (float)(row0 + k*WY) == rowf0 + k*WY exactly
        Coord c[STEPS_B];
        Pix<C> G[STEPS_B];
        float gy[STEPS_B];
        // lanes past the right/bottom edge read dOut out of range (rows >= oh: offset >= num_records; columns
        // >= ow: offset 2^31) -> the buffer unit returns 0 -> they add exactly 0 to dTheta
        unsigned voff = col_ok ? (unsigned)row0 * orow + (unsigned)col * (C * 4) : 0x80000000u;
#pragma unroll
        for (int k = 0; k < STEPS_B; ++k) {
            gy[k] = -1.0f + sy * (rowf0 + (float)(k * WY));
            G[k] = buf_load<C, UH_WARP_GLOAD_AUX>(rg, voff, 0);
            voff += (unsigned)WY * orow;
            c[k] = make_coord(th, A0, A3, A6, gy[k], g);
        }
        // one pixel's contribution to the nine sums (and, optionally, to dU)
        auto accumulate = [&](int k, const Pix<C>& Ia, const Pix<C>& Ib, const Pix<C>& Ic, const Pix<C>& Id) {
            float dx = 0.f, dy = 0.f;
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                const float ex = fmaf(c[k].ay1, Ic.v[ch] - Ia.v[ch], c[k].ay0 * (Id.v[ch] - Ib.v[ch]));
                const float ey = fmaf(c[k].ax1, Ib.v[ch] - Ia.v[ch], c[k].ax0 * (Id.v[ch] - Ic.v[ch]));
                dx = fmaf(G[k].v[ch], ex, dx);
                dy = fmaf(G[k].v[ch], ey, dy);
            }
            const float rt = c[k].rt;
            const float dxs = dx * halfW * rt, dys = dy * halfH * rt;
            const float dt = -(dxs * c[k].xs + dys * c[k].ys) * rt;
            acc[0] = fmaf(dxs, gx, acc[0]); acc[1] = fmaf(dxs, gy[k], acc[1]); acc[2] += dxs;
            acc[3] = fmaf(dys, gx, acc[3]); acc[4] = fmaf(dys, gy[k], acc[4]); acc[5] += dys;
            acc[6] = fmaf(dt,  gx, acc[6]); acc[7] = fmaf(dt,  gy[k], acc[7]); acc[8] += dt;
            if (WANT_DU) {
                const TapOff o = global_offsets<SMALL>(c[k], g);
                const float wa = c[k].ax1 * c[k].ay1, wb = c[k].ax1 * c[k].ay0;
                const float wc = c[k].ax0 * c[k].ay1, wd = c[k].ax0 * c[k].ay0;
#pragma unroll
                for (int ch = 0; ch < C; ++ch) {          // G is 0 for masked lanes/rows: adds 0
                    atomicAdd(dUb + o.oa / 4 + ch, wa * G[k].v[ch]);
                    atomicAdd(dUb + o.ob / 4 + ch, wb * G[k].v[ch]);
                    atomicAdd(dUb + o.oc / 4 + ch, wc * G[k].v[ch]);
                    atomicAdd(dUb + o.od / 4 + ch, wd * G[k].v[ch]);
                }
            }
        };
        Rect r;
        r.fits = false;
        if constexpr (STAGE_BWD) r = wave_rect<C, STEPS_B>(c);
        if (STAGE_BWD && r.fits) {                                      // wave-uniform
            stage_rect<C>(rin, lds, lane, r, g.rowBi);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // one pixel at a time (LDS latency is short): only 4 taps live at once keeps the kernel under 96 VGPRs
#pragma unroll
            for (int k = 0; k < STEPS_B; ++k) {
                const TapOff o = staged_offsets(c[k], r.pitch, g.pixB, r.nbase);
                const Pix<C> Ia = lds_load<C>(lds, o.oa), Ib = lds_load<C>(lds, o.ob);
                const Pix<C> Ic = lds_load<C>(lds, o.oc), Id = lds_load<C>(lds, o.od);
                accumulate(k, Ia, Ib, Ic, Id);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            Pix<C> Ia[STEPS_B], Ib[STEPS_B], Ic[STEPS_B], Id[STEPS_B];         // all 16 gathers in flight before the first use
#pragma unroll
            for (int k = 0; k < STEPS_B; ++k) {
                const TapOff o = global_offsets<SMALL>(c[k], g);
                Ia[k] = buf_load<C>(rin, o.oa, 0); Ib[k] = buf_load<C>(rin, o.ob, 0);
                Ic[k] = buf_load<C>(rin, o.oc, 0); Id[k] = buf_load<C>(rin, o.od, 0);
            }
#pragma unroll
            for (int k = 0; k < STEPS_B; ++k) accumulate(k, Ia[k], Ib[k], Ic[k], Id[k]);
        }
    }
    // block reduction: DPP inside rows of 16 lanes, the 16 row sums of the block meet in LDS (fixed order)
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const float rs = row16_sum(acc[j]);
        if ((lane & 15) == 0) red[j][wave * 4 + (lane >> 4)] = rs;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        const float* rr = red[threadIdx.x];
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < NWAVE * 4; ++k) t += rr[k];
        partial[(size_t)v * 9 + threadIdx.x] = t;
    }
}

// ------------------------------------------------------------------------------------------------
// Literal forward (validation only): one pixel per thread, the op-for-op transcription make_sample()/blend() with
// the compiler's IEEE division, x86-style integer cast and integer clamps, plain global loads.  It exists so that the
// lean path above (shared-reciprocal division, float-domain clips, f32 offsets, LDS staging) can be checked BIT FOR
// BIT on the GPU at full sizes and over millions of thetas' worth of samples (tests/test_gpu_parity.py).
template <int C>
__global__ __launch_bounds__(256) void warp_forward_literal_kernel(const float* __restrict__ U, const float* __restrict__ theta,
                                                                   float* __restrict__ out, int H, int W, int oh, int ow) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= oh * ow) return;
    const int row = p / ow, col = p - row * ow;
    const Theta th = load_theta(theta, b);
    const Sample s = make_sample(th, lin_at(lin_step(ow), col), lin_at(lin_step(oh), row), W, H);
    const float* Ub = U + (size_t)b * H * W * C;
    float* Ob = out + ((size_t)b * oh * ow + p) * C;
#pragma unroll
    for (int c = 0; c < C; ++c)
        Ob[c] = blend(s, Ub[((size_t)s.y0 * W + s.x0) * C + c], Ub[((size_t)s.y1 * W + s.x0) * C + c],
                      Ub[((size_t)s.y0 * W + s.x1) * C + c], Ub[((size_t)s.y1 * W + s.x1) * C + c]);
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

static inline bool small_image(int H, int W, int C) { return (uint64_t)H * W * C * 4 <= (1ull << 24); }

static int check_warp_args(int B, int H, int W, int C, int oh, int ow) {
    if (B <= 0 || H <= 0 || W <= 0 || oh <= 0 || ow <= 0) return UH_E_SHAPE;
    if (C < 1 || C > 4) return UH_E_CHANNELS;
    if ((uint64_t)H * W * C * 4 >= (1ull << 31) || (uint64_t)oh * ow * C * 4 >= (1ull << 31)) return UH_E_TOO_LARGE;
    const TileGeom g = tile_geom(oh, ow);
    if ((uint64_t)B * g.tiles >= (1ull << 31)) return UH_E_TOO_LARGE;
    return 0;
}

template <int C>
static void launch_fwd(const float* U, const float* theta, float* out, float* condition, int B, int H, int W,
                       int oh, int ow, hipStream_t s) {
    const TileGeom g = tile_geom(oh, ow);
    const unsigned nblk = (unsigned)B * g.tiles;
    const bool sm = small_image(H, W, C);
#define UH_FWD(COND, SM) launch_timed(UH_K_WARP_FWD, warp_forward_kernel<C, COND, SM>, dim3(nblk), dim3(256), s, U, \
                                     theta, out, condition, H, W, oh, ow, lin_step(ow), lin_step(oh), g.tiles_x, g.tiles, nblk)
    if (condition) { if (sm) UH_FWD(true, true); else UH_FWD(true, false); }
    else           { if (sm) UH_FWD(false, true); else UH_FWD(false, false); }
#undef UH_FWD
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
    const TileGeom g = tile_geom(oh, ow, TH_B);
    return (size_t)B * g.tiles * 9 * sizeof(float);
}

template <int C>
static void launch_bwd(const float* U, const float* theta, const float* dOut, float* partial, float* dU, int B,
                       int H, int W, int oh, int ow, hipStream_t s) {
    const TileGeom g = tile_geom(oh, ow, TH_B);
    const unsigned nblk = (unsigned)B * g.tiles;
    const bool sm = small_image(H, W, C);
#define UH_BWD(DU, SM) launch_timed(UH_K_WARP_BWD, warp_backward_kernel<C, DU, SM>, dim3(nblk), dim3(256), s, U, \
                                   theta, dOut, partial, dU, H, W, oh, ow, lin_step(ow), lin_step(oh), g.tiles_x, g.tiles, nblk)
    if (dU) { if (sm) UH_BWD(true, true); else UH_BWD(true, false); }
    else    { if (sm) UH_BWD(false, true); else UH_BWD(false, false); }
#undef UH_BWD
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
        const TileGeom g = tile_geom(oh, ow, TH_B);
        launch_timed(UH_K_WARP_BWD_FIN, warp_backward_finish_kernel, dim3((B + 3) / 4), dim3(256), s,
                     (const float*)partial, dTheta, g.tiles, B);
    }
    return (int)hipGetLastError();
}

extern "C" int uh_warp_forward_literal(const float* U, const float* theta, float* out, int B, int H, int W, int C,
                                       int oh, int ow, uh_stream_t stream) {
    if (!U || !theta || !out) return UH_E_NULL;
    if (int e = check_warp_args(B, H, W, C, oh, ow)) return e;
    if (B > 65535) return UH_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((oh * ow + 255) / 256, B), block(256);
    switch (C) {
        case 1: hipLaunchKernelGGL(warp_forward_literal_kernel<1>, grid, block, 0, s, U, theta, out, H, W, oh, ow); break;
        case 2: hipLaunchKernelGGL(warp_forward_literal_kernel<2>, grid, block, 0, s, U, theta, out, H, W, oh, ow); break;
        case 3: hipLaunchKernelGGL(warp_forward_literal_kernel<3>, grid, block, 0, s, U, theta, out, H, W, oh, ow); break;
        default: hipLaunchKernelGGL(warp_forward_literal_kernel<4>, grid, block, 0, s, U, theta, out, H, W, oh, ow); break;
    }
    return (int)hipGetLastError();
}
