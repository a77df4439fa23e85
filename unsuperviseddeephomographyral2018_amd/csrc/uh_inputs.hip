// uh_inputs.hip -- GPU producer of the dataloader's OUTPUT CONTRACT.
//
// The reference builds every training sample on the host inside TF queue runners
// (/root/reference/code/dataloader.py:160-227): cast the decoded uint8 frames to f32, photometric augmentation of
// the pair (gamma, brightness, per-channel colour, clip to [0,255] -- :323-375), (x - mean)/std (:172-177, :317-318),
// channel mean -> gray, gather of the P x P patch of I and I' by flat indices (:203-227).  Here one kernel does all of
// that for a whole batch of decoded uint8 frames already in HBM and writes exactly the tensors HomographyModel takes:
//   I_aug, I_prime_aug [B,H,W,3] f32;  I1, I2, I1_aug, I2_aug [B,P,P] f32;  patch_indices [B,P*P] i32.
// HBM-bound: reads 2*3 bytes, writes 2*12 bytes per pixel (+ the patches): 2.3 MB per 240x320 pair.
#include "uh_device.h"
#include "uh_host.h"

namespace uh {

struct Norm3 { float mean[3], inv_std_unused[3], std[3]; };
struct AugP { float gamma, bright, col[3]; };

// one channel value: augmentation in the reference's op order, then standardisation.  `vg` = v ** gamma (:357) comes from
// the block's table: a decoded pixel is one of 256 values and gamma is fixed per image, so the block evaluates powf()
// 2 x 256 times instead of 24 times per thread (the kernel was powf-bound: 123 us against 42 us without augmentation).
__device__ __forceinline__ float augment(float vg, float bright, float col) {
    float v = vg * bright;                // * random_brightness                        (:362)
    v = v * col;                          // * color_image                              (:369)
    return fminf(fmaxf(v, 0.0f), 255.0f); // tf.clip_by_value(., 0, 255)               (:373)
}

// PIX pixels per thread (4 when H*W % 4 == 0: 12 input bytes = 3 dwords, 48 output bytes = 3 x float4 per image)
template <int PIX, bool AUG>
__global__ __launch_bounds__(256) void prepare_inputs_kernel(
        const unsigned char* __restrict__ I8, const unsigned char* __restrict__ Ip8, const float* __restrict__ aug,
        const float* __restrict__ pts1, Norm3 nm, float* __restrict__ Ia, float* __restrict__ Ipa,
        float* __restrict__ I1, float* __restrict__ I2, float* __restrict__ I1a, float* __restrict__ I2a,
        int* __restrict__ pidx, int H, int W, int P, int groups_per_image) {
    const int b = blockIdx.y;
    const int grp = blockIdx.x * 256 + threadIdx.x;
    __shared__ float gamma_lut[AUG ? 2 : 1][AUG ? 256 : 1];     // v ** gamma for v = 0..255, image I and image I'
    if constexpr (AUG) {
        const float* q = aug + (size_t)b * 10;
        gamma_lut[0][threadIdx.x] = powf((float)threadIdx.x, q[0]);               // img ** random_gamma       (:357)
        gamma_lut[1][threadIdx.x] = powf((float)threadIdx.x, q[5]);
        __syncthreads();
    }
    if (grp >= groups_per_image) return;
    const int N = H * W;
    const int p0 = grp * PIX;                                   // first pixel of this thread
    const size_t ib = (size_t)b * N;
    AugP a0{1.f, 1.f, {1.f, 1.f, 1.f}}, a1 = a0;
    if (AUG) {
        const float* q = aug + (size_t)b * 10;                  // uniform -> scalar loads
        a0 = AugP{q[0], q[1], {q[2], q[3], q[4]}};
        a1 = AugP{q[5], q[6], {q[7], q[8], q[9]}};
    }
    const int x0 = (int)pts1[(size_t)b * 8], y0 = (int)pts1[(size_t)b * 8 + 1];   // top-left corner of the patch (:197-199)
    unsigned char raw[2][PIX * 3];
    if constexpr (PIX == 4) {
        const uint32_t* s0 = reinterpret_cast<const uint32_t*>(I8 + (ib + p0) * 3);
        const uint32_t* s1 = reinterpret_cast<const uint32_t*>(Ip8 + (ib + p0) * 3);
        uint32_t w0[3] = {s0[0], s0[1], s0[2]}, w1[3] = {s1[0], s1[1], s1[2]};
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            raw[0][k] = (unsigned char)(w0[k >> 2] >> (8 * (k & 3)));
            raw[1][k] = (unsigned char)(w1[k >> 2] >> (8 * (k & 3)));
        }
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) { raw[0][k] = I8[(ib + p0) * 3 + k]; raw[1][k] = Ip8[(ib + p0) * 3 + k]; }
    }
    float outa[2][PIX * 3];
#pragma unroll
    for (int px = 0; px < PIX; ++px) {
        float gray[2] = {0.f, 0.f}, graya[2] = {0.f, 0.f};
#pragma unroll
        for (int im = 0; im < 2; ++im) {
            const AugP& ap = im == 0 ? a0 : a1;
            float g = 0.f, ga = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = (float)raw[im][px * 3 + c];                         // tf.cast(image, tf.float32) (:244)
                const float n = (v - nm.mean[c]) / nm.std[c];                       // norm_img (:317-318)
                const float na = AUG ? (augment(gamma_lut[AUG ? im : 0][AUG ? raw[im][px * 3 + c] : 0], ap.bright, ap.col[c]) - nm.mean[c]) / nm.std[c] : n;
                outa[im][px * 3 + c] = na;
                g = c == 0 ? n : g + n;  ga = c == 0 ? na : ga + na;
            }
            gray[im] = g / 3.0f; graya[im] = ga / 3.0f;                             // reduce_mean(I, 2) (:210-213)
        }
        const int p = p0 + px;
        const int yy = p / W, xx = p - yy * W;
        const int u = xx - x0, v = yy - y0;
        if (u >= 0 && u < P && v >= 0 && v < P) {                                   // gather by patch_indices (:203-227)
            const size_t e = (size_t)b * P * P + (size_t)v * P + u;
            I1[e] = gray[0]; I2[e] = gray[1]; I1a[e] = graya[0]; I2a[e] = graya[1];
            pidx[e] = p;                                                            // (v + y0)*W + (u + x0)
        }
    }
    if constexpr (PIX == 4) {
        float4* d0 = reinterpret_cast<float4*>(Ia + (ib + p0) * 3);
        float4* d1 = reinterpret_cast<float4*>(Ipa + (ib + p0) * 3);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            d0[k] = make_float4(outa[0][4 * k], outa[0][4 * k + 1], outa[0][4 * k + 2], outa[0][4 * k + 3]);
            d1[k] = make_float4(outa[1][4 * k], outa[1][4 * k + 1], outa[1][4 * k + 2], outa[1][4 * k + 3]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) { Ia[(ib + p0) * 3 + k] = outa[0][k]; Ipa[(ib + p0) * 3 + k] = outa[1][k]; }
    }
}

}  // namespace uh

using namespace uh;

extern "C" int uh_prepare_inputs(const unsigned char* I_u8, const unsigned char* Iprime_u8, const float* aug,
                                 const float* pts1, const float* mean3_host, const float* std3_host,
                                 float* I_aug, float* Iprime_aug, float* I1, float* I2, float* I1_aug,
                                 float* I2_aug, int* patch_idx, int B, int H, int W, int P, uh_stream_t stream) {
    if (!I_u8 || !Iprime_u8 || !pts1 || !mean3_host || !std3_host || !I_aug || !Iprime_aug || !I1 || !I2 ||
        !I1_aug || !I2_aug || !patch_idx)
        return UH_E_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || P <= 0 || P > H || P > W || B > 65535) return UH_E_SHAPE;
    if ((uint64_t)H * W * 12 >= (1ull << 31)) return UH_E_TOO_LARGE;
    Norm3 nm;
    for (int c = 0; c < 3; ++c) { nm.mean[c] = mean3_host[c]; nm.std[c] = std3_host[c]; nm.inv_std_unused[c] = 0.f; }
    hipStream_t s = (hipStream_t)stream;
    const int N = H * W;
    const bool vec = (N % 4) == 0;
    const int groups = vec ? N / 4 : N;
    dim3 grid((groups + 255) / 256, B), block(256);
#define UH_PREP(PIX, AUG) launch_timed(UH_K_PREPARE, prepare_inputs_kernel<PIX, AUG>, grid, block, s, I_u8, Iprime_u8, aug, \
                                       pts1, nm, I_aug, Iprime_aug, I1, I2, I1_aug, I2_aug, patch_idx, H, W, P, groups)
    if (vec) { if (aug) UH_PREP(4, true); else UH_PREP(4, false); }
    else     { if (aug) UH_PREP(1, true); else UH_PREP(1, false); }
#undef UH_PREP
    return (int)hipGetLastError();
}
