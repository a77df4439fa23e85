// uh_warp.hip -- projective Spatial-Transformer bilinear warp, forward and backward (gfx950).
//
// Replaces transformer(U, theta, out_size) of /root/reference/code/utils/tf_spatial_transformer.py
// (:18-251) and TF's autodiff of it w.r.t. theta.  The TF graph materialises ~30 [B*N]-sized f32
// intermediates (~600 B/pixel); here each output pixel is produced in registers: algorithmic traffic
// is read U + write out (forward) and read dOut + read U (backward), 2*B*H*W*C*4 bytes each.
//
// Design (what the measurements forced -- DESIGN.md 3.1):
//   * HBM traffic has been minimal since the first version (FETCH/WRITE counters = unique footprint); what the
//     kernels are short of is INSTRUCTION ISSUE: round 1 ended at ~140 VALU instructions per pixel-lane with the
//     VALU ~60-75 % busy and a 12-byte gather costing 16.5 TA cycles per wave.  Round 2 therefore splits the
//     sampling law (uh_device.h: project() / clip_coord()) and gives every wave one of three paths, chosen
//     wave-uniformly from an EXACT min/max reduction of floor(x), floor(y) over the wave's pixels (DPP):
//       A  interior + rectangle fits the wave's LDS slice: the source rectangle is pulled into LDS by LDS-DMA
//          (buffer_load_dwordx4 ... lds: no VGPR round trip, no ds_write), rows packed at pitch = 16*ceil(12*w/16);
//          no clipping, ONE tap offset per pixel, the 2x2x3 neighbourhood is 6 ds_read2_b32 at constant offsets;
//          the per-pixel weights are formed while the DMA is in flight;
//       B  interior, rectangle too large (minifying regions): same cheap arithmetic, taps gathered from global
//          memory with two offsets per pixel (the x+1 taps ride on the instruction's immediate offset);
//       C  some tap is clipped (27-42 % of the tiles at the benchmark laws): floor/clip per sample (clip_pair), then
//          C1 the exact rectangle of the CLIPPED taps through LDS like A (four tap offsets per pixel), or
//          C2 the direct gather when that rectangle does not fit (far field of a strong perspective).
//     All three produce identical bits (tests/test_gpu_parity.py compares them with the f32 oracle and with the
//     literal kernel below at full size).
//   * Every wave owns a TW x (WY*STEPS) output tile (default 16 x 16: a square footprint keeps the taps of
//     a rotated tile on few source rows).  No block barrier is involved in the forward: the producer and the
//     consumers of an LDS slice are lanes of the same wave.
//   * blockIdx -> (image, tile) keeps the tiles of one image on one XCD (xcd_remap) so that the overlap of
//     neighbouring rectangles is served by that XCD's L2.
#include "uh_device.h"
#include "uh_host.h"
#include <algorithm>

namespace uh {

#ifndef UH_WARP_TW
#define UH_WARP_TW 16
#endif
#ifndef UH_WARP_STEPS
#define UH_WARP_STEPS 4
#endif
#ifndef UH_WARP_LDS_PER_WAVE
#define UH_WARP_LDS_PER_WAVE 5120  // forward: LDS slice per wave (path A holds rectangles up to this size).  5 KiB, not 6: with
                                   // 80 VGPRs the register file admits 6 waves per SIMD, and 6 x 4 x 6 KiB = 144 KiB made LDS a second
                                   // limit of exactly 6 blocks per CU -- a block whose last (gather) wave is still running then holds
                                   // the LDS a new block needs although its other three waves' registers are free.  20 KiB blocks leave
                                   // room for 8: -1.7 % at config 4, -3 % at 240x320 (profiles/r03_microbench_variants.jsonl, `l5`),
                                   // although 3 % more tiles gather
#endif
#ifndef UH_WARP_LDS_PER_WAVE_BWD
#define UH_WARP_LDS_PER_WAVE_BWD 6144
#endif
#ifndef UH_WARP_STAGE_FWD
#define UH_WARP_STAGE_FWD 1       // 0: path A off (developer A/B switch)
#endif
#ifndef UH_WARP_STAGE_BWD
#define UH_WARP_STAGE_BWD 1
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
#ifndef UH_WARP_BWD_BATCH
#define UH_WARP_BWD_BATCH 2       // backward gather paths (B, C): pixels per lane whose taps are in flight together
#endif
constexpr int BT_B = UH_WARP_BWD_BATCH;
#ifndef UH_WARP_FWD_BATCH
#define UH_WARP_FWD_BATCH 2       // forward gather paths (B, C): the same
#endif
constexpr int BT_F = UH_WARP_FWD_BATCH;
static_assert(UH_WARP_STEPS % UH_WARP_FWD_BATCH == 0, "batch must divide the steps");
#ifndef UH_WARP_FWD_MINW
#define UH_WARP_FWD_MINW 6        // __launch_bounds__ second argument: minimum waves per SIMD the register allocator must allow
                                  // (80 VGPRs, no spill; 6 blocks x 4 waves x 5 KiB LDS slices = 120 KiB of the CU's 160 KiB)
#endif
#ifndef UH_WARP_BWD_MINW
#define UH_WARP_BWD_MINW 5        // 96 VGPRs without path C1 (no spill).  The instantiations that need more registers get one wave
                                  // less instead of spilling (C = 4: 5 forward / 4 backward waves; the forward with `condition`: 5; the
                                  // backward with dU, which holds four tap offsets per pixel for the scatter: 3) -- a spill is a
                                  // scratch round trip per tile in a kernel that is short of memory slots, not of waves
#endif
static_assert(UH_WARP_STEPS_BWD % UH_WARP_BWD_BATCH == 0, "batch must divide the steps");
constexpr int NWAVE = 4;                // waves per block, side by side in x: block tile = (4*TW) x TH
constexpr bool STAGE_FWD = UH_WARP_STAGE_FWD != 0, STAGE_BWD = UH_WARP_STAGE_BWD != 0;
static_assert(TW * WY == 64, "UH_WARP_TW must divide 64");
static_assert(UH_WARP_LDS_PER_WAVE % 1024 == 0 && UH_WARP_LDS_PER_WAVE <= 16384, "LDS slice: multiple of 1 KiB, <= 16 KiB");
static_assert(UH_WARP_LDS_PER_WAVE_BWD % 1024 == 0 && UH_WARP_LDS_PER_WAVE_BWD <= 16384, "LDS slice: multiple of 1 KiB, <= 16 KiB");

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

// ---- wave-uniform path decision ------------------------------------------------------------------------
// Exact extent of floor(x), floor(y) over every pixel of the wave.  The reduction runs on the f32 BIT PATTERNS as
// signed integers: for non-negative floats that order is the numeric order; a negative float (or -NaN) anywhere
// makes the minimum negative, a +NaN / huge value makes the maximum exceed any valid bound -- both fail the
// interior test below, and nothing else is read from the result in that case.
struct Extent {
    bool interior;          // every tap of every pixel is an un-clipped source pixel
    int bx0, by0;           // top-left source pixel of the tap rectangle (interior only)
    int rw, rh;             // its width / height in pixels
};
template <int N>
__device__ __forceinline__ Extent wave_extent(const float (&fx)[N], const float (&fy)[N], const SrcGeom& g) {
    int mnx = __float_as_int(fx[0]), mxx = mnx, mny = __float_as_int(fy[0]), mxy = mny;
#pragma unroll
    for (int k = 1; k < N; ++k) {
        mnx = min(mnx, __float_as_int(fx[k])); mxx = max(mxx, __float_as_int(fx[k]));
        mny = min(mny, __float_as_int(fy[k])); mxy = max(mxy, __float_as_int(fy[k]));
    }
    mnx = wave_min_nonneg(mnx); mxx = wave_max_nonneg(mxx);      // (valid for any sign at lane 63, see uh_device.h)
    mny = wave_min_nonneg(mny); mxy = wave_max_nonneg(mxy);
    Extent e;
    // floor(x) >= 0 and floor(x) + 1 <= W - 1   (bit patterns of non-negative floats compare like the floats)
    e.interior = mnx >= 0 && mny >= 0 && mxx <= __float_as_int(g.Wm1 - 1.0f) && mxy <= __float_as_int(g.Hm1 - 1.0f)
                 && g.Wm1 >= 1.0f && g.Hm1 >= 1.0f;
    e.bx0 = (int)__int_as_float(mnx); e.by0 = (int)__int_as_float(mny);
    e.rw = (int)__int_as_float(mxx) - e.bx0 + 2; e.rh = (int)__int_as_float(mxy) - e.by0 + 2;
    return e;
}

// The same for a wave with clipped taps: exact extent of the CLIPPED corners (non-negative integers held in f32),
// ... computed straight from the pixel coordinates: the clipped corners are formed, folded into the running min / max
// and dropped (the clipped paths re-derive them per pixel: 12 cheap VALU ops instead of 4 live VGPRs per pixel).
template <int N>
__device__ __forceinline__ Extent wave_extent_clipped_xy(const Proj (&p)[N], const SrcGeom& g) {
    int mnx = 0x7fffffff, mxx = 0, mny = 0x7fffffff, mxy = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        float a0, a1, b0, b1;
        clip_pair(p[k].x, g.Wm1, a0, a1);
        clip_pair(p[k].y, g.Hm1, b0, b1);
        mnx = min(mnx, __float_as_int(a0)); mxx = max(mxx, __float_as_int(a1));
        mny = min(mny, __float_as_int(b0)); mxy = max(mxy, __float_as_int(b1));
    }
    mnx = wave_min_nonneg(mnx); mxx = wave_max_nonneg(mxx);
    mny = wave_min_nonneg(mny); mxy = wave_max_nonneg(mxy);
    Extent e;
    e.interior = true;
    e.bx0 = (int)__int_as_float(mnx); e.by0 = (int)__int_as_float(mny);
    e.rw = (int)__int_as_float(mxx) - e.bx0 + 1; e.rh = (int)__int_as_float(mxy) - e.by0 + 1;
    return e;
}

// ---- path A staging: LDS-DMA of the tap rectangle ------------------------------------------------------------
// The rectangle's rows are cut into 16-byte chunks, cpr = ceil(rw*C*4/16) per row, packed in LDS at pitch 16*cpr.  One
// buffer_load_dwordx4 ... lds moves rpi = floor(64/cpr) WHOLE rows (the LDS address of lane l is M0 + 16*l by
// construction of the instruction, so lane l = r*cpr + c lands on row r, chunk c of the instruction's block of rows):
// the lane's (r, c) and its global offset are computed ONCE, every further instruction only moves two scalars (M0 by
// rpi rows of LDS, the buffer instruction's soffset by rpi rows of the image) and lanes past the last row are masked.
// The global address of a chunk, ((by0+r)*W + bx0)*C*4 + 16*c, is only 4-byte aligned; the last chunk of a row may
// run past the rectangle (never read).  The rectangle lies inside the image on both staged paths, so no range check
// is involved.  A slice of S bytes holds any rectangle with rh*cpr*16 <= S.
struct Stage { int cpr, rpi, nld; float pitch, nbase; bool fits; };
template <int C, int LDS_BYTES>
__device__ __forceinline__ Stage plan_stage(const Extent& e) {
    Stage s;
    s.cpr = (e.rw * C * 4 + 15) >> 4;
    s.fits = e.interior && s.cpr <= 64 && s.cpr * e.rh * 16 <= LDS_BYTES;
    s.rpi = 64 / max(s.cpr, 1);                      // wave-uniform operands: the quotient lives in an SGPR (the loop
                                                     // below must step scalars, or the compiler "waterfalls" the DMA)
    s.nld = 0;
    s.pitch = (float)(s.cpr * 16);
    s.nbase = -((float)e.by0 * s.pitch + (float)(e.bx0 * C * 4));
    return s;
}
template <int C, int LDS_BYTES>
__device__ __forceinline__ void stage_dma(__amdgpu_buffer_rsrc_t rin, unsigned char* lds, int lane, const Extent& e,
                                          Stage& s, int rowBi) {
    // lane -> (row r, chunk c) of one instruction's block of rows:  r = floor((lane + 0.5) / cpr)  (off integers by >= 1/128)
    const int r = (int)(((float)lane + 0.5f) * __builtin_amdgcn_rcpf((float)s.cpr));
    const int c = lane - r * s.cpr;
    const unsigned goff = (unsigned)(e.by0 + r) * (unsigned)rowBi + (unsigned)(e.bx0 * C * 4 + c * 16);
    const int rpi = s.rpi;
    const unsigned gstep = (unsigned)rpi * (unsigned)rowBi, lstep = (unsigned)(rpi * s.cpr * 16);
    int n = 0;
#pragma unroll 1
    for (int row = 0; row < e.rh; row += rpi, ++n) {                      // wave-uniform trip count (<= LDS_BYTES / 528)
        if (r < min(rpi, e.rh - row))
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(lds + (unsigned)n * lstep), 16,
                                                     goff, (int)((unsigned)n * gstep), 0, 0);
    }
    s.nld = n;
}
// the DMA's data is in LDS once the wave's vector-memory counter has drained; nothing else orders a ds_read behind it
__device__ __forceinline__ void stage_wait() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// the 2x2 neighbourhood of an interior sample out of the staged rectangle: taps at o, o+pixB (same row), o+pitch ...
template <int C>
__device__ __forceinline__ void lds_quad(const unsigned char* lds, unsigned oa, unsigned ob, Pix<C>& Ia, Pix<C>& Ib,
                                         Pix<C>& Ic, Pix<C>& Id) {
    const float* pa = reinterpret_cast<const float*>(lds + oa);
    const float* pb = reinterpret_cast<const float*>(lds + ob);
#pragma unroll
    for (int c = 0; c < C; ++c) { Ia.v[c] = pa[c]; Ic.v[c] = pa[C + c]; Ib.v[c] = pb[c]; Id.v[c] = pb[C + c]; }
}
// ... and straight from global memory (path B): two VGPR offsets, the x+1 taps use the immediate offset
template <int C>
__device__ __forceinline__ void global_quad(__amdgpu_buffer_rsrc_t rin, unsigned oa, unsigned ob, Pix<C>& Ia, Pix<C>& Ib,
                                            Pix<C>& Ic, Pix<C>& Id) {
    Ia = buf_load<C>(rin, oa, 0); Ic = buf_load<C>(rin, oa + C * 4, 0);
    Ib = buf_load<C>(rin, ob, 0); Id = buf_load<C>(rin, ob + C * 4, 0);
}

#ifdef UH_WARP_TRACE
// Developer instrumentation (tools/trace_waves.py; never in the shipped library): lane 0 of every wave (forward, and
// the backward with UH_TRACE_BWD=1; s_memtime orders with memory waits, not with VALU work, so the compute phases are
// approximate) records
// s_memtime at its phase boundaries and the path it took.  trace[w*16 + ..] = t0 entry, t1 decision made, t2 loads /
// DMA issued, t3 data landed, t4 stores issued, [5] path (0 A, 1 B, 2 C1, 3 C2), [6] DMA instructions, [7] t_end;
// [8] / [9] s_memrealtime (the chip-wide 100 MHz counter: comparable ACROSS XCDs, which the per-XCD shader clock behind
// s_memtime is not) at entry / end, [10] XCC_ID, [11] HW_ID (CU / SE of the wave).
__device__ unsigned long long* g_trace = nullptr;
#define UH_TRACE_STRIDE 16
#define UH_TR(i) do { if (trp) trp[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define UH_TRV(i, v) do { if (trp) trp[i] = (unsigned long long)(v); } while (0)
#define UH_TR_ENTRY() do { if (trp) { trp[8] = __builtin_amdgcn_s_memrealtime(); \
                                      trp[10] = (unsigned long long)__builtin_amdgcn_s_getreg(6164);  /* hwreg(HW_REG_XCC_ID, 0, 4) */ \
                                      trp[11] = (unsigned long long)__builtin_amdgcn_s_getreg(63492); /* hwreg(HW_REG_HW_ID, 0, 32) */ } } while (0)
#define UH_TR_EXIT() do { if (trp) trp[9] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define UH_TR(i) do {} while (0)
#define UH_TRV(i, v) do {} while (0)
#define UH_TR_ENTRY() do {} while (0)
#define UH_TR_EXIT() do {} while (0)
#endif

// ------------------------------------------------------------------------------------------------
template <int C, bool COND, bool SMALL>
__global__ __launch_bounds__(256, ((C == 4 || COND) ? UH_WARP_FWD_MINW - 1 : UH_WARP_FWD_MINW)) void warp_forward_kernel(
        const float* __restrict__ U, const float* __restrict__ theta, float* __restrict__ out,
        float* __restrict__ condition, int H, int W, int oh, int ow, float sx, float sy, int tiles_x, int tiles,
        unsigned nblk) {
    constexpr int LDSW = UH_WARP_LDS_PER_WAVE;
    __shared__ __attribute__((aligned(16))) unsigned char lds_all[STAGE_FWD ? NWAVE * LDSW : 16];
    const int lane = threadIdx.x & 63, wave = wave_id();
    unsigned char* lds = lds_all + (STAGE_FWD ? wave * LDSW : 0);
    float cnt = 0.f;
    const unsigned v = xcd_remap(blockIdx.x, nblk);
    const int b = v / tiles, tile = v - b * tiles;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int col = (tx * NWAVE + wave) * TW + (lane & (TW - 1));
    const int row0 = ty * TH + (lane / TW);                              // this lane's first row
    if ((tx * NWAVE + wave) * TW >= ow) return;                          // whole wave outside (wave-uniform)
#ifdef UH_WARP_TRACE
    unsigned long long* trp = (g_trace && lane == 0) ? g_trace + ((size_t)v * NWAVE + wave) * UH_TRACE_STRIDE : nullptr;
#endif
    UH_TR_ENTRY(); UH_TR(0);
    const Theta th = load_theta(theta, b);
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(U + (size_t)b * H * W * C, (unsigned)(H * W * C * 4));
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(out + (size_t)b * oh * ow * C, (unsigned)(oh * ow * C * 4));
    const SrcGeom g = make_geom<C>(W, H);
    const float gx = lin_at(sx, col);           // lanes past the right/bottom edge compute on; their store is dropped
    const float A0 = th.a[0] * gx, A3 = th.a[3] * gx, A6 = th.a[6] * gx;
    const bool col_ok = col < ow;
    const unsigned orow = (unsigned)(ow * C * 4);
    const float rowf0 = (float)row0;                                     // (float)(row0 + k*WY) == rowf0 + k*WY exactly

    Proj p[STEPS];
    float fx[STEPS], fy[STEPS];
    {
        float gyk[STEPS];
#pragma unroll
        for (int k = 0; k < STEPS; ++k) gyk[k] = -1.0f + sy * (rowf0 + (float)(k * WY));
        project_all<STEPS>(th, A0, A3, A6, gyk, g, p);          // one eps-guard test per wave instead of one per pixel
    }
#pragma unroll
    for (int k = 0; k < STEPS; ++k) { fx[k] = floorf(p[k].x); fy[k] = floorf(p[k].y); }         // (:101,103)
    const Extent e = wave_extent<STEPS>(fx, fy, g);
    UH_TR(1);

    // Store offsets: row*orow + col*pixB, advanced by WY rows per step.  A row >= oh gives an offset >= the
    // buffer's num_records (= oh*orow), which makes the buffer unit drop the store; lanes right of the image
    // start from 2^31 and stay out of range (an image is < 2^31 bytes, check_warp_args).
    unsigned voff = col_ok ? (unsigned)row0 * orow + (unsigned)col * (C * 4) : 0x80000000u;
    auto emit = [&](int k, float ax1, float ax0, float ay1, float ay0, const Pix<C>& Ia, const Pix<C>& Ib,
                    const Pix<C>& Ic, const Pix<C>& Id) {
        const float wa = ax1 * ay1, wb = ax1 * ay0, wc = ax0 * ay1, wd = ax0 * ay0;             // (:134-137)
        Pix<C> o;
#pragma unroll
        for (int ch = 0; ch < C; ++ch) o.v[ch] = blend4(wa, wb, wc, wd, Ia.v[ch], Ib.v[ch], Ic.v[ch], Id.v[ch]);
#ifdef UH_DBG_NO_STORE            // developer A/B switch only: keeps the math alive, never stores
        buf_store<C, UH_WARP_STORE_AUX>(rout, o.v[0] != 12345.678f ? 0x80000000u : voff, 0, o);
#else
        buf_store<C, UH_WARP_STORE_AUX>(rout, voff, 0, o);
#endif
        if (COND) cnt += (col_ok && row0 + k * WY < oh && fabsf(p[k].t) > 1e-7f) ? 1.f : 0.f;   // (:235)
        voff += (unsigned)WY * orow;
    };

    if (e.interior) {                                                   // wave-uniform
        Stage st;
        st.fits = false;
        if constexpr (STAGE_FWD) st = plan_stage<C, LDSW>(e);
        float ax0[STEPS], ay0[STEPS];
        if (STAGE_FWD && st.fits) {                                     // ---- path A
            stage_dma<C, LDSW>(rin, lds, lane, e, st, g.rowBi);
            unsigned oa[STEPS];
#pragma unroll
            for (int k = 0; k < STEPS; ++k) {                           // overlaps the DMA
                ax0[k] = p[k].x - fx[k]; ay0[k] = p[k].y - fy[k];
                oa[k] = (unsigned)__builtin_fmaf(fy[k], st.pitch, __builtin_fmaf(fx[k], g.pixB, st.nbase));
            }
            const unsigned pitchi = (unsigned)(st.cpr * 16);
            UH_TR(2); UH_TRV(5, 0); UH_TRV(6, st.nld);
            stage_wait();
            UH_TR(3);
#pragma unroll
            for (int k = 0; k < STEPS; ++k) {
                Pix<C> Ia, Ib, Ic, Id;
                lds_quad<C>(lds, oa[k], oa[k] + pitchi, Ia, Ib, Ic, Id);
                emit(k, 1.0f - ax0[k], ax0[k], 1.0f - ay0[k], ay0[k], Ia, Ib, Ic, Id);
            }
            UH_TR(4);
        } else {                                                        // ---- path B
            UH_TR(2); UH_TRV(5, 1); UH_TRV(6, 0);
#pragma unroll
            for (int k0 = 0; k0 < STEPS; k0 += BT_F) {
                Pix<C> Ia[BT_F], Ib[BT_F], Ic[BT_F], Id[BT_F];                 // 4*BT_F gathers in flight
#pragma unroll
                for (int j = 0; j < BT_F; ++j) {
                    const int k = k0 + j;
                    unsigned oa;
                    if constexpr (SMALL) oa = (unsigned)__builtin_fmaf(fy[k], g.rowB, fx[k] * g.pixB);
                    else oa = (unsigned)fy[k] * (unsigned)g.rowBi + (unsigned)fx[k] * (unsigned)g.pixBi;
                    global_quad<C>(rin, oa, oa + (unsigned)g.rowBi, Ia[j], Ib[j], Ic[j], Id[j]);
                    ax0[k] = p[k].x - fx[k]; ay0[k] = p[k].y - fy[k];
                }
#pragma unroll
                for (int j = 0; j < BT_F; ++j) {
                    const int k = k0 + j;
                    emit(k, 1.0f - ax0[k], ax0[k], 1.0f - ay0[k], ay0[k], Ia[j], Ib[j], Ic[j], Id[j]);
                }
                if (BT_F < STEPS) __builtin_amdgcn_sched_barrier(0);
            }
            UH_TR(3); UH_TR(4);
        }
    } else {                                                            // ---- some tap is clipped
        Stage st;
        st.fits = false;
        Extent ec;
        if constexpr (STAGE_FWD) { ec = wave_extent_clipped_xy<STEPS>(p, g); st = plan_stage<C, LDSW>(ec); }
#ifdef UH_WARP_NO_C1               // developer A/B switch: clipped waves always gather
        st.fits = false;
#endif
        auto corners = [&](int k) {                                     // floor / clip of pixel k (fx / fy are dead here) (:101-109)
            Coord c;
            clip_pair(p[k].x, g.Wm1, c.x0f, c.x1f);
            clip_pair(p[k].y, g.Hm1, c.y0f, c.y1f);
            return c;
        };
        if (STAGE_FWD && st.fits) {                                     // ---- path C1: clipped rectangle through LDS
            stage_dma<C, LDSW>(rin, lds, lane, ec, st, g.rowBi);
            UH_TR(2); UH_TRV(5, 2); UH_TRV(6, st.nld);
            stage_wait();
            UH_TR(3);
#pragma unroll
            for (int k = 0; k < STEPS; ++k) {
                const Coord c = corners(k);
                const TapOff o = staged_offsets(c, st.pitch, g.pixB, st.nbase);
                const Pix<C> Ia = lds_load<C>(lds, o.oa), Ib = lds_load<C>(lds, o.ob);
                const Pix<C> Ic = lds_load<C>(lds, o.oc), Id = lds_load<C>(lds, o.od);
                emit(k, c.x1f - p[k].x, p[k].x - c.x0f, c.y1f - p[k].y, p[k].y - c.y0f, Ia, Ib, Ic, Id);       // (:130-137)
            }
            UH_TR(4);
        } else {                                                        // ---- path C2: clipped gather (far field)
            UH_TR(2); UH_TRV(5, 3); UH_TRV(6, 0);
#pragma unroll
            for (int k0 = 0; k0 < STEPS; k0 += BT_F) {
                Pix<C> Ia[BT_F], Ib[BT_F], Ic[BT_F], Id[BT_F];
                Coord c[BT_F];
#pragma unroll
                for (int j = 0; j < BT_F; ++j) {
                    c[j] = corners(k0 + j);
                    const TapOff o = global_offsets<SMALL>(c[j], g);
                    Ia[j] = buf_load<C>(rin, o.oa, 0); Ib[j] = buf_load<C>(rin, o.ob, 0);
                    Ic[j] = buf_load<C>(rin, o.oc, 0); Id[j] = buf_load<C>(rin, o.od, 0);
                }
#pragma unroll
                for (int j = 0; j < BT_F; ++j) {
                    const int k = k0 + j;
                    emit(k, c[j].x1f - p[k].x, p[k].x - c[j].x0f, c[j].y1f - p[k].y, p[k].y - c[j].y0f, Ia[j], Ib[j], Ic[j], Id[j]);
                }
                if (BT_F < STEPS) __builtin_amdgcn_sched_barrier(0);
            }
            UH_TR(3); UH_TR(4);
        }
    }
#ifdef UH_WARP_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // stores acknowledged
    UH_TR(7); UH_TR_EXIT();
#endif
    if (COND) {
        cnt = wave_sum(cnt);
        if (lane == 0) atomicAdd(condition, cnt);
    }
}

// ------------------------------------------------------------------------------------------------
// Backward w.r.t. theta (and optionally U).  Closed form of TF's autodiff (floor/clip/cast carry no
// gradient; weights use the clipped corners):
//   dx  = sum_c g_c [ ay1 (Ic - Ia) + ay0 (Id - Ib) ]      dy = sum_c g_c [ ax1 (Ib - Ia) + ax0 (Id - Ic) ]
//   (evaluated in the algebraically equal "lerp" form, see accumulate())
//   dxn = dx W/2, dyn = dy H/2, dxs = dxn/t, dys = dyn/t, dt = -(dxn xs + dyn ys)/t^2
//   dTheta = [dxs; dys; dt] (3xN) . grid^T (Nx3)
// The differences (Ic - Ia) ... are exactly 0 where the clip collapsed a corner pair, so far-field
// samples contribute exactly nothing (the TF op order leaves f32 cancellation noise there).
// 1/t is the refined reciprocal the sampling already produced (<= 1 ulp; the gradient is checked
// against the f64 closed form, not bit-for-bit).  Same three paths as the forward; the per-pixel sums are
// the same expressions on every path, so dTheta does not depend on the LDS slice size.
//
// PATCH mode (uh_warp_patch_backward): dOut is not a frame but dPred [B,PP], the gradient of the gray patch gather
// (homography_model.py:263-269).  The frame gradient it stands for is dPred[e]/C on the pixels patch_idx names and 0
// elsewhere, so (a) tiles that miss the patch rectangle are skipped (their partial is exactly the 0 the dense kernel
// would have summed), and (b) a lane takes G = dPred[e]/C for the entry e that SHOULD sit on its pixel if the patch is
// the rectangle anchored at patch_idx[k,0] and whose stored index confirms it.  Entries that are not at their rectangle
// position (arbitrary gathers, duplicates) are counted and handled one by one in the finish kernel, so any index set
// gives the gradient of uh_gray_patch_backward -> uh_warp_backward; on rectangles the sums are bit-identical to it.
// PATCH launches only the block tiles that can touch the rectangle: nrx x nry per image, anchored at the tile holding
// (x0, y0).  The finish kernel treats every other tile's partial as the 0 the dense kernel would have written there
// WITHOUT reading it (same summation order, same bits, no memset of the partial buffer).
struct PatchArgs { const int* idx; int P, PP; int* confirmed; int nrx, nry;
                   int kind; const float* pred; const float* target; const float* stats; const float* dLoss; size_t n; };
__host__ __device__ inline void patch_tile_range(int x0, int y0, int P, int tiles_x, int tiles_y, int th, int& txlo, int& txhi,
                                                 int& tylo, int& tyhi) {
    txlo = x0 / (NWAVE * TW); txhi = min((x0 + P - 1) / (NWAVE * TW), tiles_x - 1);
    tylo = y0 / th;           tyhi = min((y0 + P - 1) / th, tiles_y - 1);
}
// Deterministic finish: dTheta[b][j] = sum over the image's tiles, accumulated in f64, fixed order.
// One wave per image.  The image's partials are `tiles*9` contiguous floats; lane l < 63 walks them
// with stride 63 (= 7*9), so its accumulator index j = l % 9 never changes and every load instruction
// is coalesced and independent of the others; the 7 lanes that share a j then meet in LDS.
// PATCH mode adds the entries the bandwidth kernel could not take (not at their rectangle position): if the image's
// confirmed count falls short of PP, the wave walks the PP entries and samples the stray ones itself (literal path C
// arithmetic, f64 accumulation, fixed order).  For the dataloader's rectangles nothing is added.
// Where the patch gradient comes from: kind < 0 -> dPred [B,PP] as given; kind = UH_LOSS_* (point-wise kinds) -> formed on
// the fly from (pred, target, lc) exactly as uh_patch_loss_backward would have written it (uh_warp_patch_loss_backward).
// ge / C, correctly rounded (the dense chain's uh_gray_patch_backward divides), without the v_div_* sequence: exact
// scalings for C = 1, 2, 4; for C = 3 Markstein's correction step -- y = RN(1/3), q = RN(a y) is within 1 ulp of a/3,
// r = a - 3q is exact in one fma, RN(q + r y) is the correctly rounded quotient.  (Below |a| ~ 1e-30 the residual can leave
// the f32 range and the quotient may be 1 ulp of a DENORMAL off the division's: a gradient of that size adds nothing.)
template <int C>
__device__ __forceinline__ float div_by_channels(float a) {
    if constexpr (C == 3) {
        const float y = 1.0f / 3.0f;
        const float q = a * y;
        const float r = __builtin_fmaf(-3.0f, q, a);
        return __builtin_fmaf(r, y, q);
    } else {
        return a * (1.0f / (float)C);
    }
}
struct LossSrc { int kind; const float* pred; const float* target; LossCoef lc; };
__device__ __forceinline__ float patch_grad(const LossSrc& ls, const float* __restrict__ dPred, size_t e) {
    return ls.kind < 0 ? dPred[e] : loss_grad_point(ls.kind, ls.pred[e], ls.target[e], ls.lc);
}
struct PatchFinish { const float* U; const float* theta; const float* dPred; const int* idx; const int* confirmed;
                     int P, PP, H, W; float sx, sy; int tiles_x; LossSrc ls; };
// `red` = 64 doubles of LDS private to the calling wave.
template <int C, bool PATCH>
__device__ __forceinline__ void finish_image(const float* __restrict__ partial, float* __restrict__ dTheta, int tiles, int b,
                                             const PatchFinish& pf, double* red, int lane) {
    double a = 0.0;
    int txlo = 0, txhi = -1, tylo = 0, tyhi = -1;                        // PATCH: the tiles the bandwidth kernel wrote
    if constexpr (PATCH) {
        const int o = pf.P > 0 ? pf.idx[(size_t)b * pf.PP] : -1;
        if (o >= 0 && o < pf.H * pf.W) {                                  // (an out-of-range anchor: no rectangle, as in the kernel)
            patch_tile_range(o - (o / pf.W) * pf.W, o / pf.W, pf.P, pf.tiles_x, tiles / pf.tiles_x, TH_B, txlo, txhi, tylo, tyhi);
        }
    }
    auto written = [&](int tile) {                                       // every other partial counts as 0 (and is never read)
        if constexpr (!PATCH) return true;
        const int ty = tile / pf.tiles_x, tx = tile - ty * pf.tiles_x;
        return tx >= txlo && tx <= txhi && ty >= tylo && ty <= tyhi;
    };
    const float* __restrict__ rp = partial + (size_t)b * tiles * 9;
    // element i if `ok`, else 0 -- without a branch, so that a batch of them stays in flight together
    auto f32_if = [&](bool ok, int i) { const float t = rp[ok ? i : 0]; return ok ? t : 0.f; };
    if (lane < 63) {
        const int n = tiles * 9;
        // batches of FB loads in flight, added in index order (the sums do not depend on FB)
        constexpr int FB = 8;
        if constexpr (PATCH) {
            // e advances by 63 = 7 tiles: walk (tx, ty) incrementally instead of dividing per element
            int tile = lane / 9, ty = tile / pf.tiles_x, tx = tile - ty * pf.tiles_x;
            for (int e0 = lane; e0 < n; e0 += 63 * FB) {
                float v[FB];
#pragma unroll
                for (int q = 0; q < FB; ++q) {
                    const int e = e0 + 63 * q;
                    v[q] = f32_if(e < n && tx >= txlo && tx <= txhi && ty >= tylo && ty <= tyhi, e);
                    tx += 7;
                    while (tx >= pf.tiles_x) { tx -= pf.tiles_x; ++ty; }
                }
#pragma unroll
                for (int q = 0; q < FB; ++q) a += (double)v[q];
            }
        } else {
            for (int e0 = lane; e0 < n; e0 += 63 * FB) {
                float v[FB];
#pragma unroll
                for (int q = 0; q < FB; ++q) v[q] = f32_if(e0 + 63 * q < n, e0 + 63 * q);
#pragma unroll
                for (int q = 0; q < FB; ++q) a += (double)v[q];
            }
        }
    }
    red[lane] = a;
    double extra = 0.0;                                                  // lane j < 9: stray entries' share of sum j
    if constexpr (PATCH) {
        {
            int cnt = 0;
            const int* __restrict__ rc = pf.confirmed + (size_t)b * tiles;
            for (int e = lane; e < tiles; e += 64) cnt += written(e) ? rc[e] : 0;
            cnt = (int)wave_sum((double)cnt);
            if (cnt != pf.PP) {
                const int H = pf.H, W = pf.W, P = pf.P, PP = pf.PP;
                const int* idx = pf.idx + (size_t)b * PP;
                const int o = idx[0];
                const bool anchored = P > 0 && o >= 0 && o < H * W;
                const int y0 = anchored ? o / W : 0, x0 = anchored ? o - y0 * W : 0;
                const Theta th = load_theta(pf.theta, b);
                const SrcGeom g = make_geom<C>(W, H);
                const float* Ub = pf.U + (size_t)b * H * W * C;
                const float halfW = (float)W * 0.5f, halfH = (float)H * 0.5f;
                double acc[9];
#pragma unroll
                for (int j = 0; j < 9; ++j) acc[j] = 0.0;
                for (int e = lane; e < PP; e += 64) {
                    const int t = idx[e];
                    bool direct = false;
                    if (anchored) {
                        const int v = e / P, u = e - v * P;
                        direct = (x0 + u < W) && (y0 + v < H) && t == (y0 + v) * W + (x0 + u);
                    }
                    if (direct) continue;
                    if (t < 0 || t >= H * W) continue;                  // not a pixel of the frame (tf.gather would raise): adds nothing
                    const int row = t / W, col = t - row * W;
                    const float gx = lin_at(pf.sx, col), gy = lin_at(pf.sy, row);
                    const Coord c = make_coord(th, th.a[0] * gx, th.a[3] * gx, th.a[6] * gx, gy, g);
                    const float gv = patch_grad(pf.ls, pf.dPred, (size_t)b * PP + e) / (float)C;
                    const size_t ia = ((size_t)c.y0f * W + (size_t)c.x0f) * C, ib = ((size_t)c.y1f * W + (size_t)c.x0f) * C;
                    const size_t ic = ((size_t)c.y0f * W + (size_t)c.x1f) * C, id = ((size_t)c.y1f * W + (size_t)c.x1f) * C;
                    const float hx = c.x1f - c.x0f, hy = c.y1f - c.y0f;
                    float s1 = 0.f, sb = 0.f, sc = 0.f;
#pragma unroll
                    for (int ch = 0; ch < C; ++ch) {
                        const float Ia = Ub[ia + ch], Ib = Ub[ib + ch], Ic = Ub[ic + ch], Id = Ub[id + ch];
                        const float ddb = Id - Ib, ddc = Id - Ic, u = (Ic - Ia) - ddb;
                        s1 = fmaf(gv, u, s1); sb = fmaf(gv, ddb, sb); sc = fmaf(gv, ddc, sc);
                    }
                    const float dx = fmaf(c.ay1, s1, hy * sb), dy = fmaf(c.ax1, s1, hx * sc);
                    const float dxs = dx * halfW * c.rt, dys = dy * halfH * c.rt;
                    const float dt = -(dxs * c.xs + dys * c.ys) * c.rt;
                    acc[0] += (double)dxs * gx; acc[1] += (double)dxs * gy; acc[2] += (double)dxs;
                    acc[3] += (double)dys * gx; acc[4] += (double)dys * gy; acc[5] += (double)dys;
                    acc[6] += (double)dt * gx;  acc[7] += (double)dt * gy;  acc[8] += (double)dt;
                }
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    const double tj = wave_sum(acc[j]);
                    if (lane == j) extra = tj;
                }
            }
        }
    }
    wave_lds_sync();
    if (lane < 9) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 7; ++k) t += red[lane + 9 * k];
        dTheta[(size_t)b * 9 + lane] = (float)(t + extra);
    }
}

template <int C, bool PATCH>
__global__ __launch_bounds__(256) void warp_backward_finish_kernel(const float* __restrict__ partial,
                                                                   float* __restrict__ dTheta, int tiles, int B,
                                                                   PatchFinish pf, const float* __restrict__ stats,
                                                                   const float* __restrict__ dLoss, size_t n) {
    __shared__ double red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    if constexpr (PATCH) { if (pf.ls.kind >= 0) pf.ls.lc = loss_coef(pf.ls.kind, stats, dLoss ? dLoss[0] : 1.0f, n); }
    if (b < B) finish_image<C, PATCH>(partial, dTheta, tiles, b, pf, red[wave], lane);
}

template <int C, bool WANT_DU, bool SMALL, bool PATCH = false>
__global__ __launch_bounds__(256, (WANT_DU ? 3 : (C == 4 ? UH_WARP_BWD_MINW - 1 : UH_WARP_BWD_MINW))) void warp_backward_kernel(
        const float* __restrict__ U, const float* __restrict__ theta, const float* __restrict__ dOut,
        float* __restrict__ partial, float* __restrict__ dU,
        int H, int W, int oh, int ow, float sx, float sy, int tiles_x, int tiles, unsigned nblk, PatchArgs pa) {
    constexpr int LDSW = UH_WARP_LDS_PER_WAVE_BWD;
    __shared__ __attribute__((aligned(16))) unsigned char lds_all[STAGE_BWD ? NWAVE * LDSW : 16];
    __shared__ float red[9][NWAVE * 4];
    const int lane = threadIdx.x & 63, wave = wave_id();
    unsigned v = xcd_remap(blockIdx.x, nblk);
    int b, ty, tx;
    int px0 = 0, py0 = 0;                                                // PATCH: top-left pixel of the patch rectangle
    LossSrc ls{-1, nullptr, nullptr, LossCoef{0.f, 0.f, 0.f}};
    if constexpr (PATCH) {
        if (pa.kind >= 0) ls = LossSrc{pa.kind, pa.pred, pa.target, loss_coef(pa.kind, pa.stats, pa.dLoss ? pa.dLoss[0] : 1.0f, pa.n)};   // uniform
        const int per = pa.nrx * pa.nry;
        b = v / per;
        const int r = v - b * per, j = r / pa.nrx, i = r - j * pa.nrx;
        const int o = pa.idx[(size_t)b * pa.PP];                         // uniform -> scalar load
        if (o < 0 || o >= H * W) return;                                 // no valid anchor: every entry is a stray (finish kernel)
        py0 = o / W; px0 = o - py0 * W;
        int txlo, txhi, tylo, tyhi;
        patch_tile_range(px0, py0, pa.P, tiles_x, tiles / tiles_x, TH_B, txlo, txhi, tylo, tyhi);
        tx = txlo + i; ty = tylo + j;
        if (pa.P <= 0 || tx > txhi || ty > tyhi) return;                 // whole block: not a tile of this rectangle
        v = (unsigned)(b * tiles + ty * tiles_x + tx);                   // index of the tile's partial (dense layout)
    } else {
        b = v / tiles;
        const int tile = v - b * tiles;
        ty = tile / tiles_x; tx = tile - ty * tiles_x;
    }
    const int col = (tx * NWAVE + wave) * TW + (lane & (TW - 1));
    const int row0 = ty * TH_B + (lane / TW);                              // this lane's first row
#ifdef UH_WARP_TRACE
    unsigned long long* trp = (g_trace && lane == 0) ? g_trace + ((size_t)v * NWAVE + wave) * UH_TRACE_STRIDE : nullptr;
#endif
    UH_TR_ENTRY(); UH_TR(0);
    float acc[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[j] = 0.f;
    int nconf = 0;                                                       // PATCH: entries this lane confirmed
    bool live = (tx * NWAVE + wave) * TW < ow;                           // wave-uniform; else the wave adds 0
    if constexpr (PATCH) {
        const int wx0 = (tx * NWAVE + wave) * TW, wy0 = ty * TH_B;
        live = live && wx0 < px0 + pa.P && wx0 + TW > px0 && wy0 < py0 + pa.P && wy0 + TH_B > py0;
    }
    if (live) {
        unsigned char* lds = lds_all + (STAGE_BWD ? wave * LDSW : 0);
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
        Proj p[STEPS_B];
        float fx[STEPS_B], fy[STEPS_B];
        Pix<C> G[STEPS_B];
        float gy[STEPS_B];
        // lanes past the right/bottom edge read dOut out of range (rows >= oh: offset >= num_records; columns
        // >= ow: offset 2^31) -> the buffer unit returns 0 -> they add exactly 0 to dTheta
        unsigned voff = col_ok ? (unsigned)row0 * orow + (unsigned)col * (C * 4) : 0x80000000u;
#pragma unroll
        for (int k = 0; k < STEPS_B; ++k) {
            gy[k] = -1.0f + sy * (rowf0 + (float)(k * WY));
            if constexpr (PATCH) {
                const int row = row0 + k * WY, u = col - px0, v = row - py0;
                float gv = 0.f;
                if (u >= 0 && u < pa.P && v >= 0 && v < pa.P && col < ow && row < oh) {
                    const size_t e = (size_t)b * pa.PP + (size_t)(v * pa.P + u);
                    const int ie = pa.idx[e];
                    const float ge = patch_grad(ls, dOut, e);            // (loaded alongside idx[e], not after the compare)
                    if (ie == row * W + col) { gv = div_by_channels<C>(ge); ++nconf; }
                }
#pragma unroll
                for (int ch = 0; ch < C; ++ch) G[k].v[ch] = gv;
            } else {
                G[k] = buf_load<C, UH_WARP_GLOAD_AUX>(rg, voff, 0);
            }
            voff += (unsigned)WY * orow;
        }
        project_all<STEPS_B>(th, A0, A3, A6, gy, g, p);
#pragma unroll
        for (int k = 0; k < STEPS_B; ++k) { fx[k] = floorf(p[k].x); fy[k] = floorf(p[k].y); }
        const Extent e = wave_extent<STEPS_B>(fx, fy, g);
        UH_TR(1);
        // one pixel's contribution to the nine sums
        // hx = x1f - x0f, hy = y1f - y0f (1 for an interior sample, 0 where the clip collapsed the pair).  With
        // ay0 = hy - ay1:  ay1 (Ic-Ia) + ay0 (Id-Ib) = ay1 [(Ic-Ia) - (Id-Ib)] + hy (Id-Ib)  -- the form used here: a
        // sample whose y pair collapsed has Ic-Ia == Id-Ib and hy == 0 and contributes EXACTLY 0 although |ay| reaches
        // 1e7 px there (the literal form leaves eps*|ay|*|dI| of cancellation noise per far-field sample); likewise
        // for x.
        auto accumulate = [&](int k, float ax1, float ay1, float hx, float hy, const Pix<C>& Ia, const Pix<C>& Ib,
                              const Pix<C>& Ic, const Pix<C>& Id) {
            // (Ic-Ia) - (Id-Ib) == (Ib-Ia) - (Id-Ic) =: u (the mixed second difference), so with the channel sums
            // S1 = sum g u, Sb = sum g (Id-Ib), Sc = sum g (Id-Ic):   dx = ay1 S1 + hy Sb,   dy = ax1 S1 + hx Sc
            float s1 = 0.f, sb = 0.f, sc = 0.f;
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                const float ddb = Id.v[ch] - Ib.v[ch], ddc = Id.v[ch] - Ic.v[ch];
                const float u = (Ic.v[ch] - Ia.v[ch]) - ddb;
                s1 = fmaf(G[k].v[ch], u, s1);
                sb = fmaf(G[k].v[ch], ddb, sb);
                sc = fmaf(G[k].v[ch], ddc, sc);
            }
            const float dx = fmaf(ay1, s1, hy * sb), dy = fmaf(ax1, s1, hx * sc);
            const float rt = p[k].rt;
            const float dxs = dx * halfW * rt, dys = dy * halfH * rt;
            const float dt = -(dxs * p[k].xs + dys * p[k].ys) * rt;
            // gx is the same for the lane's four pixels (one column): the gx-weighted sums are formed once, after the loop
            const float gyk = gy[k];
            acc[1] = fmaf(dxs, gyk, acc[1]); acc[2] += dxs;
            acc[4] = fmaf(dys, gyk, acc[4]); acc[5] += dys;
            acc[7] = fmaf(dt,  gyk, acc[7]); acc[8] += dt;
        };
        // optional dU: scatter of the four weighted taps (the only float atomics of the library).  A sample whose x (or y)
        // pair collapsed under the clip sends wa + wc = (ax1 + ax0) ay1 = 0 to ONE pixel: in exact arithmetic it
        // contributes nothing, in f32 it would leave |ax| ~ 1e7 px worth of cancellation noise there -- skipped (`live`).
        auto scatter = [&](int k, const TapOff& o, float ax1, float ax0, float ay1, float ay0, bool live) {
            if (!live) return;
            const float wa = ax1 * ay1, wb = ax1 * ay0, wc = ax0 * ay1, wd = ax0 * ay0;
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {          // G is 0 for masked lanes/rows: adds 0
                atomicAdd(dUb + o.oa / 4 + ch, wa * G[k].v[ch]);
                atomicAdd(dUb + o.ob / 4 + ch, wb * G[k].v[ch]);
                atomicAdd(dUb + o.oc / 4 + ch, wc * G[k].v[ch]);
                atomicAdd(dUb + o.od / 4 + ch, wd * G[k].v[ch]);
            }
        };
        if (e.interior) {                                               // wave-uniform
            Stage st;
            st.fits = false;
            if constexpr (STAGE_BWD) st = plan_stage<C, LDSW>(e);
            float ax0[STEPS_B], ay0[STEPS_B];
            auto og = [&](int k) -> unsigned {                          // global offset of tap (y0, x0): paths B and dU only
                if constexpr (SMALL) return (unsigned)__builtin_fmaf(fy[k], g.rowB, fx[k] * g.pixB);
                else return (unsigned)fy[k] * (unsigned)g.rowBi + (unsigned)fx[k] * (unsigned)g.pixBi;
            };
            if (STAGE_BWD && st.fits) {                                 // ---- path A
                stage_dma<C, LDSW>(rin, lds, lane, e, st, g.rowBi);
                unsigned oa[STEPS_B];
#pragma unroll
                for (int k = 0; k < STEPS_B; ++k) {
                    const float fxk = fx[k], fyk = fy[k];
                    ax0[k] = p[k].x - fxk; ay0[k] = p[k].y - fyk;
                    oa[k] = (unsigned)__builtin_fmaf(fyk, st.pitch, __builtin_fmaf(fxk, g.pixB, st.nbase));
                }
                const unsigned pitchi = (unsigned)(st.cpr * 16);
                UH_TR(2); UH_TRV(5, 0); UH_TRV(6, st.nld);
                stage_wait();
                UH_TR(3);
#pragma unroll
                for (int k = 0; k < STEPS_B; ++k) {
                    Pix<C> Ia, Ib, Ic, Id;
                    lds_quad<C>(lds, oa[k], oa[k] + pitchi, Ia, Ib, Ic, Id);
                    accumulate(k, 1.0f - ax0[k], 1.0f - ay0[k], 1.0f, 1.0f, Ia, Ib, Ic, Id);
                }
                UH_TR(4);
            } else {                                                    // ---- path B
                UH_TR(2); UH_TRV(5, 1); UH_TRV(6, 0);
#pragma unroll
                for (int k0 = 0; k0 < STEPS_B; k0 += BT_B) {
                    Pix<C> Ia[BT_B], Ib[BT_B], Ic[BT_B], Id[BT_B];             // 4*BT_B gathers in flight
#pragma unroll
                    for (int j = 0; j < BT_B; ++j) {
                        const int k = k0 + j;
                        const unsigned o = og(k);
                        global_quad<C>(rin, o, o + (unsigned)g.rowBi, Ia[j], Ib[j], Ic[j], Id[j]);
                        ax0[k] = p[k].x - fx[k]; ay0[k] = p[k].y - fy[k];
                    }
#pragma unroll
                    for (int j = 0; j < BT_B; ++j) {
                        const int k = k0 + j;
                        accumulate(k, 1.0f - ax0[k], 1.0f - ay0[k], 1.0f, 1.0f, Ia[j], Ib[j], Ic[j], Id[j]);
                    }
                    if (BT_B < STEPS_B) __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (WANT_DU) {
#pragma unroll
                for (int k = 0; k < STEPS_B; ++k) {
                    TapOff o;
                    o.oa = og(k); o.ob = o.oa + (unsigned)g.rowBi; o.oc = o.oa + C * 4; o.od = o.ob + C * 4;
                    scatter(k, o, 1.0f - ax0[k], ax0[k], 1.0f - ay0[k], ay0[k], true);
                }
            }
        } else {                                                        // ---- some tap is clipped
            // (a staged clipped path like the forward's C1 cost the backward 27 VGPRs -- 4 instead of 5 waves per SIMD -- and
            // measured no faster than the gather: profiles/r02_microbench_variants.jsonl, `noc1`)
            auto corners = [&](int k) {                                 // floor / clip of pixel k              (:101-109)
                Coord c;
                clip_pair(p[k].x, g.Wm1, c.x0f, c.x1f);
                clip_pair(p[k].y, g.Hm1, c.y0f, c.y1f);
                return c;
            };
            // ---- path C2: clipped gather
            UH_TR(2); UH_TRV(5, 3); UH_TRV(6, 0);
#pragma unroll
            for (int k0 = 0; k0 < STEPS_B; k0 += BT_B) {
                Pix<C> Ia[BT_B], Ib[BT_B], Ic[BT_B], Id[BT_B];             // 4*BT_B gathers in flight
                Coord c[BT_B];
#pragma unroll
                for (int j = 0; j < BT_B; ++j) {
                    c[j] = corners(k0 + j);
                    const TapOff o = global_offsets<SMALL>(c[j], g);
                    Ia[j] = buf_load<C>(rin, o.oa, 0); Ib[j] = buf_load<C>(rin, o.ob, 0);
                    Ic[j] = buf_load<C>(rin, o.oc, 0); Id[j] = buf_load<C>(rin, o.od, 0);
                }
#pragma unroll
                for (int j = 0; j < BT_B; ++j) {
                    const int k = k0 + j;
                    accumulate(k, c[j].x1f - p[k].x, c[j].y1f - p[k].y, c[j].x1f - c[j].x0f, c[j].y1f - c[j].y0f,
                               Ia[j], Ib[j], Ic[j], Id[j]);
                }
                if (BT_B < STEPS_B) __builtin_amdgcn_sched_barrier(0);
            }
            if (WANT_DU) {
#pragma unroll
                for (int k = 0; k < STEPS_B; ++k) {
                    const Coord c = corners(k);
                    scatter(k, global_offsets<SMALL>(c, g), c.x1f - p[k].x, p[k].x - c.x0f, c.y1f - p[k].y, p[k].y - c.y0f,
                            c.x1f != c.x0f && c.y1f != c.y0f);
                }
            }
        }
        acc[0] = gx * acc[2]; acc[3] = gx * acc[5]; acc[6] = gx * acc[8];
#ifdef UH_WARP_TRACE
        if (trp && trp[3] == 0) { UH_TR(3); UH_TR(4); }                   // (the gather paths stamp nothing between issue and here)
#endif
    }
    // block reduction: DPP inside rows of 16 lanes, the 16 row sums of the block meet in LDS (fixed order)
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const float rs = row16_sum(acc[j]);
        if ((lane & 15) == 0) red[j][wave * 4 + (lane >> 4)] = rs;
    }
    __shared__ int redc[NWAVE];
    if constexpr (PATCH) {
        const int wc = (int)wave_sum((float)nconf);                      // <= 256: exact in f32
        if (lane == 0) redc[wave] = wc;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        const float* rr = red[threadIdx.x];
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < NWAVE * 4; ++k) t += rr[k];
        partial[(size_t)v * 9 + threadIdx.x] = t;
    }
    if constexpr (PATCH) {
        if (threadIdx.x == 0) {
            const int nc = (redc[0] + redc[1]) + (redc[2] + redc[3]);
            pa.confirmed[v] = nc;
        }
    }
    UH_TR(7); UH_TR_EXIT();
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
#define UH_BWD(DU, SM) launch_timed(UH_K_WARP_BWD, warp_backward_kernel<C, DU, SM, false>, dim3(nblk), dim3(256), s, U, \
                                   theta, dOut, partial, dU, H, W, oh, ow, lin_step(ow), lin_step(oh), g.tiles_x, g.tiles, nblk, \
                                   PatchArgs{nullptr, 0, 0, nullptr, 0, 0, -1, nullptr, nullptr, nullptr, nullptr, 0})
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
    const TileGeom g = tile_geom(oh, ow, TH_B);
    launch_timed(UH_K_WARP_BWD_FIN, warp_backward_finish_kernel<1, false>, dim3((B + 3) / 4), dim3(256), s,
                 (const float*)partial, dTheta, g.tiles, B, PatchFinish{}, (const float*)nullptr, (const float*)nullptr, (size_t)0);
    return (int)hipGetLastError();
}

// ---- sparse backward: dOut given as the gradient of the gray patch gather ----------------------------------------
extern "C" size_t uh_warp_patch_backward_workspace_bytes(int B, int H, int W, int C) {
    if (check_warp_args(B, H, W, C, H, W)) return 0;
    const TileGeom g = tile_geom(H, W, TH_B);
    return (size_t)B * g.tiles * (9 * sizeof(float) + sizeof(int));
}

// `src` says where the patch gradient comes from (dPred, or a point-wise loss formed on the fly)
struct PatchGradSrc { int kind; const float* dPred; const float* pred; const float* target; const float* stats; const float* dLoss; };
template <int C>
static void launch_patch_bwd(const float* U, const float* theta, const PatchGradSrc& src, const int* idx, float* partial,
                             int* confirmed, float* dTheta, int B, int H, int W, int P, int PP, hipStream_t s) {
    const TileGeom g = tile_geom(H, W, TH_B);
    // block tiles a P x P rectangle can touch: one more than it spans when aligned
    const int nrx = P > 0 ? std::min(g.tiles_x, (P - 1) / (NWAVE * TW) + 2) : 1;
    const int nry = P > 0 ? std::min(g.tiles_y, (P - 1) / TH_B + 2) : 1;
    const unsigned nblk = (unsigned)B * nrx * nry;
    const size_t n = (size_t)B * PP;
    // P == 0 (no rectangle): no block tile is launched per image, the finish kernel does all the work
    const PatchArgs pa{idx, P, PP, confirmed, nrx, nry, src.kind, src.pred, src.target, src.stats, src.dLoss, n};
    if (small_image(H, W, C))
        launch_timed(UH_K_WARP_BWD, warp_backward_kernel<C, false, true, true>, dim3(nblk), dim3(256), s, U, theta, src.dPred,
                     partial, (float*)nullptr, H, W, H, W, lin_step(W), lin_step(H), g.tiles_x, g.tiles, nblk, pa);
    else
        launch_timed(UH_K_WARP_BWD, warp_backward_kernel<C, false, false, true>, dim3(nblk), dim3(256), s, U, theta, src.dPred,
                     partial, (float*)nullptr, H, W, H, W, lin_step(W), lin_step(H), g.tiles_x, g.tiles, nblk, pa);
    const PatchFinish pf{U, theta, src.dPred, idx, confirmed, P, PP, H, W, lin_step(W), lin_step(H), g.tiles_x,
                         LossSrc{src.kind, src.pred, src.target, LossCoef{0.f, 0.f, 0.f}}};
    launch_timed(UH_K_WARP_BWD_FIN, warp_backward_finish_kernel<C, true>, dim3((B + 3) / 4), dim3(256), s,
                 (const float*)partial, dTheta, g.tiles, B, pf, src.stats, src.dLoss, n);
}

static int patch_backward(const float* U, const float* theta, const PatchGradSrc& src, const int* patch_idx, float* dTheta,
                          void* workspace, size_t workspace_bytes, int B, int H, int W, int C, int PP, uh_stream_t stream) {
    if (int e = check_warp_args(B, H, W, C, H, W)) return e;
    if (PP <= 0) return UH_E_SHAPE;
    if ((uint64_t)B * PP >= (1ull << 31)) return UH_E_TOO_LARGE;
    if (!workspace || workspace_bytes < uh_warp_patch_backward_workspace_bytes(B, H, W, C)) return UH_E_WORKSPACE;
    int P = 0;                                   // side of the square patch; 0 = no rectangle (every entry is a stray)
    for (int q = 1; q * q <= PP; ++q) if (q * q == PP) P = q;
    if (P > H || P > W) P = 0;
    const TileGeom g = tile_geom(H, W, TH_B);
    float* partial = (float*)workspace;
    int* confirmed = (int*)(partial + (size_t)B * g.tiles * 9);
    hipStream_t s = (hipStream_t)stream;
    switch (C) {
        case 1: launch_patch_bwd<1>(U, theta, src, patch_idx, partial, confirmed, dTheta, B, H, W, P, PP, s); break;
        case 2: launch_patch_bwd<2>(U, theta, src, patch_idx, partial, confirmed, dTheta, B, H, W, P, PP, s); break;
        case 3: launch_patch_bwd<3>(U, theta, src, patch_idx, partial, confirmed, dTheta, B, H, W, P, PP, s); break;
        default: launch_patch_bwd<4>(U, theta, src, patch_idx, partial, confirmed, dTheta, B, H, W, P, PP, s); break;
    }
    return (int)hipGetLastError();
}

extern "C" int uh_warp_patch_backward(const float* U, const float* theta, const float* dPred, const int* patch_idx,
                                      float* dTheta, void* workspace, size_t workspace_bytes, int B, int H, int W, int C,
                                      int PP, uh_stream_t stream) {
    if (!U || !theta || !dPred || !patch_idx || !dTheta) return UH_E_NULL;
    return patch_backward(U, theta, PatchGradSrc{-1, dPred, nullptr, nullptr, nullptr, nullptr}, patch_idx, dTheta, workspace,
                          workspace_bytes, B, H, W, C, PP, stream);
}

// d loss / d theta straight from (pred_I2, I2_aug): uh_patch_loss_backward -> uh_warp_patch_backward without the dPred
// tensor and its launch.  Point-wise kinds only (the SSIM gradient is a 3x3 stencil: it keeps its own kernel).
extern "C" int uh_warp_patch_loss_backward(int kind, const float* U, const float* theta, const float* pred, const float* target,
                                           const float* stats16, const float* dLoss, const int* patch_idx, float* dTheta,
                                           void* workspace, size_t workspace_bytes, int B, int H, int W, int C, int PP,
                                           uh_stream_t stream) {
    if (!U || !theta || !pred || !target || !stats16 || !patch_idx || !dTheta) return UH_E_NULL;   // dLoss NULL = 1
    if (kind != UH_LOSS_REC && kind != UH_LOSS_L1 && kind != UH_LOSS_L1_SMOOTH && kind != UH_LOSS_NCC) return UH_E_SHAPE;
    return patch_backward(U, theta, PatchGradSrc{kind, nullptr, pred, target, stats16, dLoss}, patch_idx, dTheta, workspace,
                          workspace_bytes, B, H, W, C, PP, stream);
}

#ifdef UH_WARP_TRACE
extern "C" __attribute__((visibility("default"))) int uh_debug_set_trace(void* p) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(uh::g_trace), &p, sizeof(p));
}
#endif

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
