// uh_device.h -- device-side building blocks shared by every kernel of the hot path (gfx950 only).
//
// Everything that decides WHICH source pixels a sample touches lives here, once, so that forward,
// backward and the fused patch kernel cannot disagree on a floor().  The arithmetic restates
// /root/reference/code/utils/tf_spatial_transformer.py op for op (line numbers in comments) and the
// translation unit is compiled with -ffp-contract=off: one IEEE rounding per written operation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define UH_WAVE 64

namespace uh {

struct Theta { float a[9]; };          // row-major 3x3, passed in SGPRs

// tf.linspace(-1, 1, n)[i] as TF1 evaluates it: start + step*i, step = (stop-start)/(n-1)   (:162-165)
__device__ __forceinline__ float lin_step(int n) { return n > 1 ? 2.0f / (float)(n - 1) : 0.0f; }
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

// ---- XCD-aware block remap -------------------------------------------------------------------
// The dispatcher places block b on XCD b % 8.  Give every XCD one CONTIGUOUS range of virtual block
// ids so that the tiles of one image (which share source rows) meet in one 4 MiB L2.  Bijective for
// any grid size (cdna_hip_programming.md section 5, "XCD swizzle must be bijective").  Speed only.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
#ifdef UH_NO_XCD_REMAP            // developer A/B switch only (tools/variants.sh)
    return bid;
#endif
    unsigned q = nblk >> 3, r = nblk & 7u, xcd = bid & 7u, slot = bid >> 3;
    unsigned base = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + slot;
}

// ---- wave / block reductions -------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, UH_WAVE);
    return v;    // valid in lane 0
}

}  // namespace uh
