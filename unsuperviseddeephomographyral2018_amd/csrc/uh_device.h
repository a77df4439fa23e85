// uh_device.h -- device-side building blocks shared by every kernel of the hot path (gfx950 only).
//
// Everything that decides WHICH source pixels a sample touches lives here, once, so that forward,
// backward and the fused patch kernel cannot disagree on a floor().  The arithmetic restates
// /root/reference/code/utils/tf_spatial_transformer.py op for op (line numbers in comments) and the
// translation unit is compiled with -ffp-contract=off: one IEEE rounding per written operation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/uh_hotpath.h"

#define UH_WAVE 64

namespace uh {

struct Theta { float a[9]; };          // row-major 3x3, passed in SGPRs

// tf.linspace(-1, 1, n)[i] as TF1 evaluates it: start + step*i, step = (stop-start)/(n-1)   (:162-165)
__host__ __device__ __forceinline__ float lin_step(int n) { return n > 1 ? 2.0f / (float)(n - 1) : 0.0f; }
__device__ __forceinline__ float lin_at(float step, int i) { return -1.0f + step * (float)i; }

// tf.cast(float -> int32) as x86 does it (cvttss2si): NaN / out of range -> INT32_MIN.  `f` is
// already floor()ed.  AMD's v_cvt_i32_f32 saturates instead, which would pick the opposite border.
__device__ __forceinline__ int cast_i32_x86(float f) {
    return (f >= -2147483648.0f && f < 2147483648.0f) ? (int)f : INT32_MIN;
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

struct Sample {
    float xs, ys, t;        // T_g rows after the eps-guard on t            (:213-234)
    float x, y;             // pixel coordinates                              (:97-98)
    int   x0, x1, y0, y1;   // clipped integer corners                        (:101-109)
    float ax1, ax0, ay1, ay0;   // x1f-x, x-x0f, y1f-y, y-y0f  (clipped corners) (:130-137)
};

// Coordinates of output sample (gx, gy) of the normalised grid under theta, source image W x H.
__device__ __forceinline__ Sample make_sample(const Theta& th, float gx, float gy, int W, int H) {
    Sample s;
    // T_g = theta @ [gx, gy, 1]  -- k-sequential, no contraction                              (:213)
    s.xs = (th.a[0] * gx + th.a[1] * gy) + th.a[2];
    s.ys = (th.a[3] * gx + th.a[4] * gy) + th.a[5];
    float t = (th.a[6] * gx + th.a[7] * gy) + th.a[8];
    // t += 1e-6 * (1 - [|t| >= 1e-7])                                                       (:230-234)
    float ge = (fabsf(t) >= 1e-7f) ? 1.0f : 0.0f;
    t = t + 1e-6f * (1.0f - ge);
    s.t = t;
    float xn = s.xs / t;                                                                   // (:239)
    float yn = s.ys / t;                                                                   // (:240)
    s.x = ((xn + 1.0f) * (float)W) / 2.0f;                                                 // (:97)
    s.y = ((yn + 1.0f) * (float)H) / 2.0f;                                                 // (:98)
    int x0 = cast_i32_x86(floorf(s.x));                                                    // (:101)
    int y0 = cast_i32_x86(floorf(s.y));                                                    // (:103)
    // x1 = x0 + 1 cannot overflow: |x0| <= 2^31 - 128 or x0 == INT32_MIN
    s.x0 = clampi(x0, 0, W - 1);  s.x1 = clampi(x0 + 1, 0, W - 1);                         // (:106-107)
    s.y0 = clampi(y0, 0, H - 1);  s.y1 = clampi(y0 + 1, 0, H - 1);                         // (:108-109)
    s.ax1 = (float)s.x1 - s.x;  s.ax0 = s.x - (float)s.x0;                                 // (:130-137)
    s.ay1 = (float)s.y1 - s.y;  s.ay0 = s.y - (float)s.y0;
    return s;
}

// ((wa*Ia + wb*Ib) + wc*Ic) + wd*Id   -- tf.add_n order                                      (:134-138)
__device__ __forceinline__ float blend(const Sample& s, float Ia, float Ib, float Ic, float Id) {
    float wa = s.ax1 * s.ay1, wb = s.ax1 * s.ay0, wc = s.ax0 * s.ay1, wd = s.ax0 * s.ay0;
    return ((wa * Ia + wb * Ib) + wc * Ic) + wd * Id;
}

// ================================================================================================
// Lean sampling path (what the bandwidth kernels use).  Same arithmetic as make_sample()/blend() --
// bit-identical results on every input the reference handles (tests/test_gpu_parity.py checks the two
// against each other and against the f32 oracle) -- but re-expressed for the gfx950 VALU, whose issue
// cost per wave64 instruction was measured with tools/ubench (profiles/r01_gfx950_instruction_costs.txt):
//   2 cycles: v_add/sub/mul/fma_f32, v_add_u32, v_and, v_mov      4 cycles: everything else (cmp, cndmask,
//   cvt, floor, med3, min/max, lshl, mul24/mad24, 64-bit adds, v_div_*)      8 cycles: v_rcp_f32
// The first kernels spent ~500 VALU cycles per 64 output pixels and were VALU-bound, not HBM-bound
// (SQ_ACTIVE_INST_VALU = 70 % of the kernel's duration).  The diet:
//   * ONE reciprocal shared by xs/t and ys/t: the exact Newton/fma sequence LLVM emits for an IEEE f32
//     division (rcp, 2 fma, then per numerator mul + 4 fma), minus v_div_scale/v_div_fixup, which are
//     identities here because the eps-guard bounds |t| >= 1e-7 (no denormal / overflow scaling) --
//     68 -> 32 cycles, same bits;
//   * clip in the float domain (v_med3_f32) with the x86 cvttss2si overflow rule folded into one
//     compare+select, instead of cvt -> int compare -> int clamp -> cvt back;
//   * byte offsets formed in f32 (exact below 2^24) and converted once; 32-bit offsets into a buffer
//     resource (no 64-bit address arithmetic, no sign extension).
// ================================================================================================

typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x3_t __attribute__((ext_vector_type(3)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

template <int C> struct Pix { float v[C]; };

// raw buffer resource over [ptr, ptr + bytes): 32-bit byte offsets, out-of-range reads return 0
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* ptr, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, (int)bytes, 0x00020000);
}

// AUX = cache-policy bits of the buffer instruction (0 = default; 2 = nt, "non-temporal": streamed once)
template <int C, int AUX = 0>
__device__ __forceinline__ Pix<C> buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    Pix<C> p;
    if constexpr (C == 1) {
        p.v[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, AUX));
    } else if constexpr (C == 2) {
        u32x2_t t = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX);
        p.v[0] = __uint_as_float(t.x); p.v[1] = __uint_as_float(t.y);
    } else if constexpr (C == 3) {
        u32x3_t t = __builtin_amdgcn_raw_buffer_load_b96(r, voff, soff, AUX);
        p.v[0] = __uint_as_float(t.x); p.v[1] = __uint_as_float(t.y); p.v[2] = __uint_as_float(t.z);
    } else {
        u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX);
        p.v[0] = __uint_as_float(t.x); p.v[1] = __uint_as_float(t.y); p.v[2] = __uint_as_float(t.z);
        p.v[3] = __uint_as_float(t.w);
    }
    return p;
}

template <int C, int AUX = 0>
__device__ __forceinline__ void buf_store(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, const Pix<C>& p) {
    if constexpr (C == 1) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(p.v[0]), r, voff, soff, AUX);
    } else if constexpr (C == 2) {
        __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{__float_as_uint(p.v[0]), __float_as_uint(p.v[1])}, r, voff, soff, AUX);
    } else if constexpr (C == 3) {
        __builtin_amdgcn_raw_buffer_store_b96(u32x3_t{__float_as_uint(p.v[0]), __float_as_uint(p.v[1]),
                                                      __float_as_uint(p.v[2])}, r, voff, soff, AUX);
    } else {
        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{__float_as_uint(p.v[0]), __float_as_uint(p.v[1]),
                                                       __float_as_uint(p.v[2]), __float_as_uint(p.v[3])}, r, voff, soff, AUX);
    }
}

// Per-image constants of the source frame, prepared once per wave (uniform -> SGPRs).
struct SrcGeom {
    float Wf, Hf;           // (float)W, (float)H                    (:97-98)
    float Wm1, Hm1;         // clip bounds W-1, H-1                  (:106-109)
    float rowB, pixB;       // bytes per source row / per pixel, as f32 (exact: image < 2^24 bytes on this path)
    int   rowBi, pixBi;     // the same as integers (large-image path)
};
template <int C>
__device__ __forceinline__ SrcGeom make_geom(int W, int H) {
    SrcGeom g;
    g.Wf = (float)W; g.Hf = (float)H; g.Wm1 = (float)(W - 1); g.Hm1 = (float)(H - 1);
    g.rowBi = W * C * 4; g.pixBi = C * 4; g.rowB = (float)g.rowBi; g.pixB = (float)g.pixBi;
    return g;
}

// One sample, before addressing: clipped integer corners (as exact f32), 1-D weights, and what the
// backward needs.
struct Coord {
    float x0f, x1f, y0f, y1f;       // clipped corners                                          (:106-109)
    float ax1, ax0, ay1, ay0;       // x1f-x, x-x0f, y1f-y, y-y0f                               (:130-137)
    float xs, ys, t, rt;            // T_g rows after the eps-guard, and the refined reciprocal of t
};
// byte offsets of the four taps: (y0,x0) (y1,x0) (y0,x1) (y1,x1)
struct TapOff { unsigned oa, ob, oc, od; };
struct Tap : TapOff {
    float ax1, ax0, ay1, ay0;
    float hx, hy;                   // x1f - x0f, y1f - y0f: 1, or 0 where the clip collapsed the pair
    float xs, ys, t, rt;
};

// floor + clip of one coordinate, x86 cast semantics included:
//   floor(v) >= 2^31 or NaN -> cvttss2si gives INT32_MIN -> both corners clip to 0
//   floor(v) <  -2^31       -> INT32_MIN as well        -> both corners clip to 0  (med3 does that already)
__device__ __forceinline__ void clip_pair(float v, float hi, float& c0, float& c1) {
    float fl = floorf(v);                                                                   // (:101,103)
    fl = (fl < 2147483648.0f) ? fl : -1.0f;
    c0 = __builtin_amdgcn_fmed3f(fl, 0.0f, hi);                                             // (:106,108)
    c1 = __builtin_amdgcn_fmed3f(fl + 1.0f, 0.0f, hi);                                      // (:107,109)
}

// ---- the sampling law in two halves --------------------------------------------------------------------
// project(): everything up to the pixel coordinates (x, y) -- shared by every path of the bandwidth kernels.
// clip_coord(): floor/clip/weights for samples that may touch the border (what make_coord() does after x, y).
// A wave whose samples are ALL strictly interior (0 <= floor(x), floor(x)+1 <= W-1, same for y: decided once per
// wave from an exact min/max reduction of the floors) skips clip_coord(): there x0f = floor(x), x1f = x0f+1, and
//   ax0 = x - x0f          exact (fractional part of a non-negative float)
//   ax1 = (x0f+1) - x      == 1 - ax0 bit for bit: both round the same real number 1 - frac(x)
// and the four taps sit at oa, oa+pixB, oa+pitch, oa+pitch+pixB: ONE offset instead of four.
struct Proj { float xs, ys, t, rt, x, y; };
// project() in pieces, so that a kernel can test the eps-guard ONCE for all the pixels of a wave instead of paying
// add + compare + select per pixel (project_all): guard_t() = the guard, project_divide() = everything after it, given xs,
// ys and the guarded t.  |t| < 1e-7 is a measure-zero event; the kernels take a wave-uniform slow branch when any lane sees it.
__device__ __forceinline__ float guard_t(float t) {
    return (fabsf(t) >= 1e-7f) ? t : t + 1e-6f;      // == t + 1e-6*(1 - [|t| >= 1e-7])        (:230-234)
}
__device__ __forceinline__ void project_divide(Proj& s, const SrcGeom& g) {
    const float t = s.t;
    const float y0 = __builtin_amdgcn_rcpf(t);
    const float e0 = __builtin_fmaf(-t, y0, 1.0f);
    const float y1 = __builtin_fmaf(e0, y0, y0);
    s.rt = y1;
    float q, r;
    q = s.xs * y1; r = __builtin_fmaf(-t, q, s.xs); q = __builtin_fmaf(r, y1, q);
    r = __builtin_fmaf(-t, q, s.xs); const float xn = __builtin_fmaf(r, y1, q);             // (:239)
    q = s.ys * y1; r = __builtin_fmaf(-t, q, s.ys); q = __builtin_fmaf(r, y1, q);
    r = __builtin_fmaf(-t, q, s.ys); const float yn = __builtin_fmaf(r, y1, q);             // (:240)
    s.x = ((xn + 1.0f) * g.Wf) * 0.5f;               // /2 == *0.5 exactly                      (:97)
    s.y = ((yn + 1.0f) * g.Hf) * 0.5f;                                                      // (:98)
}
// the pixels of one lane: the third row t for all of them, ONE guard test for the wave, then rows 1-2 and the
// divisions pixel by pixel (only the N values of t are live across the test).  gy[k] = -1 + sy*(row0 + k*WY), formed by
// the caller as the reference does.
template <int N>
__device__ __forceinline__ void project_all(const Theta& th, float A0, float A3, float A6, const float (&gy)[N],
                                            const SrcGeom& g, Proj (&p)[N]) {
#pragma unroll
    for (int k = 0; k < N; ++k) p[k].t = (A6 + th.a[7] * gy[k]) + th.a[8];                  // (:213), NOT guarded yet
    float m = fabsf(p[0].t);
#pragma unroll
    for (int k = 1; k < N; ++k) m = fminf(m, fabsf(p[k].t));          // (a NaN t needs no guard: NaN + 1e-6 is NaN)
    if (__builtin_amdgcn_ballot_w64(m < 1e-7f) != 0ull) {              // wave-uniform, practically never taken
#pragma unroll
        for (int k = 0; k < N; ++k) p[k].t = guard_t(p[k].t);
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
        p[k].xs = (A0 + th.a[1] * gy[k]) + th.a[2];                                         // (:213)
        p[k].ys = (A3 + th.a[4] * gy[k]) + th.a[5];
        project_divide(p[k], g);
    }
}
__device__ __forceinline__ Proj project(const Theta& th, float A0, float A3, float A6, float gy, const SrcGeom& g) {
    Proj s;
    s.xs = (A0 + th.a[1] * gy) + th.a[2];                                                   // (:213)
    s.ys = (A3 + th.a[4] * gy) + th.a[5];
    s.t = guard_t((A6 + th.a[7] * gy) + th.a[8]);
    project_divide(s, g);
    return s;
}
__device__ __forceinline__ Coord clip_coord(const Proj& p, const SrcGeom& g) {
    Coord s;
    s.xs = p.xs; s.ys = p.ys; s.t = p.t; s.rt = p.rt;
    clip_pair(p.x, g.Wm1, s.x0f, s.x1f);
    clip_pair(p.y, g.Hm1, s.y0f, s.y1f);
    s.ax1 = s.x1f - p.x;  s.ax0 = p.x - s.x0f;                                              // (:130-137)
    s.ay1 = s.y1f - p.y;  s.ay0 = p.y - s.y0f;
    return s;
}

// A0/A3/A6 = theta[0]*gx, theta[3]*gx, theta[6]*gx (hoisted: gx is fixed per lane); gy varies per row.
__device__ __forceinline__ Coord make_coord(const Theta& th, float A0, float A3, float A6, float gy, const SrcGeom& g) {
    return clip_coord(project(th, A0, A3, A6, gy, g), g);
}

// tap offsets inside the image in global memory
template <bool SMALL>
__device__ __forceinline__ TapOff global_offsets(const Coord& c, const SrcGeom& g) {
    TapOff o;
    if constexpr (SMALL) {      // every product below is an exact small integer in f32
        const float xa = c.x0f * g.pixB, xc = c.x1f * g.pixB;
        o.oa = (unsigned)__builtin_fmaf(c.y0f, g.rowB, xa);
        o.ob = (unsigned)__builtin_fmaf(c.y1f, g.rowB, xa);
        o.oc = (unsigned)__builtin_fmaf(c.y0f, g.rowB, xc);
        o.od = (unsigned)__builtin_fmaf(c.y1f, g.rowB, xc);
    } else {
        const unsigned xa = (unsigned)c.x0f * (unsigned)g.pixBi, xc = (unsigned)c.x1f * (unsigned)g.pixBi;
        const unsigned ra = (unsigned)c.y0f * (unsigned)g.rowBi, rb = (unsigned)c.y1f * (unsigned)g.rowBi;
        o.oa = ra + xa; o.ob = rb + xa; o.oc = ra + xc; o.od = rb + xc;
    }
    return o;
}

// tap offsets inside a staged copy of the source rectangle [bx0..] x [by0..] with row pitch `pitch` bytes;
// nbase = -(by0*pitch + bx0*pixB).  All quantities are exact small integers in f32 (region <= 64 KB).
__device__ __forceinline__ TapOff staged_offsets(const Coord& c, float pitch, float pixB, float nbase) {
    TapOff o;
    const float xa = __builtin_fmaf(c.x0f, pixB, nbase), xc = __builtin_fmaf(c.x1f, pixB, nbase);
    o.oa = (unsigned)__builtin_fmaf(c.y0f, pitch, xa);
    o.ob = (unsigned)__builtin_fmaf(c.y1f, pitch, xa);
    o.oc = (unsigned)__builtin_fmaf(c.y0f, pitch, xc);
    o.od = (unsigned)__builtin_fmaf(c.y1f, pitch, xc);
    return o;
}

template <int C, bool SMALL>
__device__ __forceinline__ Tap make_tap(const Theta& th, float A0, float A3, float A6, float gy, const SrcGeom& g) {
    const Coord c = make_coord(th, A0, A3, A6, gy, g);
    Tap s;
    static_cast<TapOff&>(s) = global_offsets<SMALL>(c, g);
    s.ax1 = c.ax1; s.ax0 = c.ax0; s.ay1 = c.ay1; s.ay0 = c.ay0;
    s.hx = c.x1f - c.x0f; s.hy = c.y1f - c.y0f;
    s.xs = c.xs; s.ys = c.ys; s.t = c.t; s.rt = c.rt;
    return s;
}

// one pixel out of LDS at a 4-byte-aligned byte offset.  12-byte pixels are read as ds_read2_b32 + ds_read_b32:
// on gfx950 a b64/b96/b128 LDS read that is not naturally aligned is serialised lane by lane (64 cycles, measured).
template <int C>
__device__ __forceinline__ Pix<C> lds_load(const unsigned char* lds, unsigned off) {
    Pix<C> p;
    const float* q = reinterpret_cast<const float*>(lds + off);
#pragma unroll
    for (int c = 0; c < C; ++c) p.v[c] = q[c];
    return p;
}

// wave-wide min / max of NON-NEGATIVE integers (DPP butterflies, each step one fused v_min/max_i32_dpp:
// lanes whose row is masked or whose source is invalid take the identity `OLD`), result wave-uniform.
template <int CTRL, int ROW_MASK, int OLD>
__device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(OLD, v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ int wave_min_nonneg(int v) {
    constexpr int ID = 0x7fffffff;
    v = min(v, dpp_i32<0xB1, 0xf, ID>(v)); v = min(v, dpp_i32<0x4E, 0xf, ID>(v));
    v = min(v, dpp_i32<0x141, 0xf, ID>(v)); v = min(v, dpp_i32<0x140, 0xf, ID>(v));
    v = min(v, dpp_i32<0x142, 0xa, ID>(v)); v = min(v, dpp_i32<0x143, 0xc, ID>(v));
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_max_nonneg(int v) {
    v = max(v, dpp_i32<0xB1, 0xf, 0>(v)); v = max(v, dpp_i32<0x4E, 0xf, 0>(v));
    v = max(v, dpp_i32<0x141, 0xf, 0>(v)); v = max(v, dpp_i32<0x140, 0xf, 0>(v));
    v = max(v, dpp_i32<0x142, 0xa, 0>(v)); v = max(v, dpp_i32<0x143, 0xc, 0>(v));
    return __builtin_amdgcn_readlane(v, 63);
}

// ((wa*Ia + wb*Ib) + wc*Ic) + wd*Id   -- tf.add_n order                                      (:134-138)
__device__ __forceinline__ float blend4(float wa, float wb, float wc, float wd, float Ia, float Ib, float Ic, float Id) {
    return ((wa * Ia + wb * Ib) + wc * Ic) + wd * Id;
}

// ---- XCD-aware block remap -------------------------------------------------------------------
// The dispatcher places block b on XCD b % 8.  Give every XCD one CONTIGUOUS range of virtual block
// ids so that the tiles of one image (which share source rows) meet in one 4 MiB L2.  Bijective for
// any grid size (cdna_hip_programming.md section 5, "XCD swizzle must be bijective").  Speed only.
// (Round 5 measured the price of this choice -- the XCDs finish a batch-64 launch 5 - 7 us apart because whole images differ in
// cost -- and the alternative, chunks of 1 / 3 / 15 tile rows dealt round-robin to the XCDs: +0.5 ... +2.7 us warm (lost L2
// sharing), no gain cold beyond the session noise: profiles/r05_cold_forward_xcd_chunk_variants.jsonl.  Contiguous stays.)
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
#ifdef UH_NO_XCD_REMAP            // developer A/B switch only (tools/variants.sh)
    return bid;
#endif
    unsigned q = nblk >> 3, r = nblk & 7u, xcd = bid & 7u, slot = bid >> 3;
    unsigned base = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + slot;
}

// ---- wave / block reductions -------------------------------------------------------------------
// f32: DPP butterfly inside each row of 16 lanes, then row_bcast15 / row_bcast31 (gfx9 DPP controls) --
// six v_add_f32_dpp, no LDS traffic (a __shfl_down tree is six ds_bpermute_b32 round trips).  The total is
// returned wave-uniform (v_readlane of lane 63).  Fixed order -> bit-reproducible.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f32<0xB1, 0xf>(v);      // quad_perm:[1,0,3,2]
    v += dpp_f32<0x4E, 0xf>(v);      // quad_perm:[2,3,0,1]
    v += dpp_f32<0x141, 0xf>(v);     // row_half_mirror
    v += dpp_f32<0x140, 0xf>(v);     // row_mirror      -> every lane holds its row's sum
    v += dpp_f32<0x142, 0xa>(v);     // row_bcast:15    -> rows 1,3 += rows 0,2
    v += dpp_f32<0x143, 0xc>(v);     // row_bcast:31    -> rows 2,3 += row 1
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// Sum inside each row of 16 lanes only (4 fused v_add_f32_dpp); every lane of a row ends up with its
// row's sum.  Block reductions finish the 4 rows x NWAVE waves through LDS.
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f32<0xB1, 0xf>(v);
    v += dpp_f32<0x4E, 0xf>(v);
    v += dpp_f32<0x141, 0xf>(v);
    v += dpp_f32<0x140, 0xf>(v);
    return v;
}
// f64 (tiny finishing kernels only): shuffle tree, total broadcast from lane 0
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, UH_WAVE);
    return __shfl(v, 0, UH_WAVE);
}

// wave index of this thread as a scalar (threadIdx.x >> 6 is wave-uniform but the compiler cannot know)
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// LDS hand-over between the lanes of ONE wave (its ds ops execute in order; this only stops the compiler reordering)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- point-wise photometric loss gradients (shared by uh_patch_loss_backward and uh_warp_patch_loss_backward) -------------
// dPred_p = dLoss * f(x_p, y_p; global sums), x = pred_I2, y = I2_aug, N = B*P*P, stats = the 16 floats of
// uh_patch_losses_forward:
//   rec    : d / (N rec)                      rec = sqrt(mean d^2)                              (homography_model.py:303)
//   l1     : sign(d) / N                                                                        (:328)
//   smooth : (|d| < 1 ? d : sign(d)) / N                                                        (:136-139)
//   ncc    : -(1/ncc) [ (y/|y| - x/|x|)/|x| - x (<x,y>/|y| - |x|)/|x|^3 ]                        (:161-166)
// A zero norm (rec == 0, ncc == 0, |x| == 0) is 0/0 in the reference's autodiff; 0 is written here.
// rec and ncc divide by GLOBAL sums over the whole shard: one pair whose pred is NaN (degenerate predicted corners) makes
// those sums NaN and with them every pair's dTheta -- so under UH_DLT_ZERO_NONFINITE_GRAD the DLT backward zeroes ALL B
// rows for that step (the step is a no-op apart from Adam's moment decay), where l1 / l1_smooth / ssim lose only the bad
// pair.  The reference would have propagated NaN into every variable in all five cases.  The trainer's log line counts the
// zeroed pairs (uh_dlt_zeroed_pairs), so such a step shows up as B pairs at once.
struct LossCoef { float c0, c1, c2; };
__device__ __forceinline__ LossCoef loss_coef(int kind, const float* __restrict__ stats, float g, size_t n) {
    LossCoef c{0.f, 0.f, 0.f};
    const float inv_n = 1.0f / (float)n;
    if (kind == UH_LOSS_REC) {
        const float rec = stats[0];
        c.c0 = rec > 0.f ? g * inv_n / rec : 0.f;
    } else if (kind == UH_LOSS_NCC) {
        const float lx = sqrtf(stats[9]), ly = sqrtf(stats[10]), sxy = stats[11], ncc = stats[4];
        if (lx > 0.f && ly > 0.f && ncc > 0.f) {
            const float k = -g / ncc;
            c.c0 = k / (lx * ly);                                   // * y
            c.c1 = -k / (lx * lx) - k * (sxy / ly - lx) / (lx * lx * lx);   // * x
        }
    } else {
        c.c2 = g * inv_n;
    }
    return c;
}
__device__ __forceinline__ float loss_grad_point(int kind, float x, float y, const LossCoef& c) {
    const float d = x - y;
    if (kind == UH_LOSS_REC) return c.c0 * d;
    if (kind == UH_LOSS_NCC) return c.c0 * y + c.c1 * x;
    if (kind == UH_LOSS_L1) return d > 0.f ? c.c2 : (d < 0.f ? -c.c2 : 0.f);
    return fabsf(d) < 1.0f ? c.c2 * d : (d > 0.f ? c.c2 : -c.c2);
}

}  // namespace uh
