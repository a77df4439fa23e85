/*
 * uh_hotpath.h -- C ABI of the MI355X (gfx950) hot path of "Unsupervised Deep Homography":
 * Tensor-DLT + projective Spatial-Transformer bilinear warp + photometric L1, forward and backward.
 *
 * The reference (tynguyen/unsupervisedDeepHomographyRAL2018) is pure Python/TF1: it has no FFI.  The
 * seam these entry points replace is the Python-level operator seam (paths relative to
 * /root/reference/code/):
 *
 *   uh_dlt_forward / uh_dlt_backward
 *        <- HomographyModel.solve_DLT()              homography_model.py:169-250  (+ tf autodiff of it:
 *           MatrixSolveGrad) and the theta = M^-1 H M fold of transform()   homography_model.py:254
 *   uh_warp_forward / uh_warp_backward
 *        <- transformer(U, theta, out_size)          utils/tf_spatial_transformer.py:18-251
 *           (_meshgrid :141, _transform :182, _interpolate :76) and tf autodiff of it w.r.t. theta
 *   uh_gray_patch_forward / uh_gray_patch_backward
 *        <- reduce_mean(axis=3) + flat gather by patch_indices + batch_indices
 *                                                     homography_model.py:74-76,263-269
 *   uh_l1_loss_forward / uh_l1_loss_backward
 *        <- build_losses() l1 branch                  homography_model.py:328
 *   uh_prepare_inputs
 *        <- the per-sample graph of Dataloader.__init__ (cast, augment, standardise, gray patches, patch indices)
 *                                                     dataloader.py:160-227,317-375
 *   uh_patch_losses_forward
 *        <- build_losses(): rec / ssim / l1 / l1_smooth / ncc / h monitors   homography_model.py:136-166,286-352
 *   uh_tail_create / uh_tail_run
 *        <- solve_DLT + transform + l1 loss and their backward as one hipGraph
 *   uh_warp_patch_l1_fwdbwd
 *        <- the composition of the four above restricted to the P x P loss patch
 *
 * Conventions
 *   - every `const float*` / `float*` / `const int*` below is a DEVICE pointer owned by the caller,
 *     contiguous, row-major, 4-byte aligned, unless the comment says HOST.  Images are NHWC.  The conv
 *     epilogues and uh_prepare_inputs move 16 bytes per lane: give them 16-byte aligned tensors (any
 *     hipMalloc / torch allocation is 256-byte aligned).
 *   - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream) and returns; nothing allocates or frees device memory.  The calls that SYNCHRONISE are
 *     exactly two, both host-side read-backs: uh_dlt_zeroed_pairs (waits for `stream`) and
 *     uh_profile_read (waits for the profiler's events).  uh_tail_run stream-captures on the first
 *     sight of an argument set (host work, no device wait).
 *   - state the library keeps between calls -- all of it:
 *       (1) the optional launch profiler (uh_profile_*): a process-wide on/off flag, a mask, an event
 *           pool and per-kernel totals behind one mutex; off by default;
 *       (2) one device-side counter per device, the pairs zeroed by UH_DLT_ZERO_NONFINITE_GRAD
 *           (written only when that flag is passed; read and cleared by uh_dlt_zeroed_pairs);
 *       (3) per uh_tail_plan (owned by the caller, not global): its captured hipGraphs (LRU of 8) and
 *           launch counters, behind the plan's mutex.
 *     No library-owned streams, no events outside the profiler, no host or device allocations.
 *   - entry points are thread-safe (uh_tail_run is serialised per plan; see its comment).
 *   - return value: 0 = ok; >0 = a hipError_t raised by the launch; <0 = argument error (UH_E_*).
 *   - arithmetic: IEEE f32, FP contraction OFF on the forward paths so that results are op-for-op
 *     those of the un-fused TF-CPU graph (see DESIGN.md "Numerics").
 */
#ifndef UH_HOTPATH_H
#define UH_HOTPATH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UH_ABI_VERSION 8

#if defined(__GNUC__)
#define UH_API __attribute__((visibility("default")))
#else
#define UH_API
#endif

/* argument errors */
#define UH_E_NULL        (-1)   /* a required pointer is NULL                     */
#define UH_E_SHAPE       (-2)   /* non-positive or inconsistent dimension          */
#define UH_E_CHANNELS    (-3)   /* C not in {1,2,3,4}                             */
#define UH_E_WORKSPACE   (-4)   /* workspace missing or smaller than *_workspace_bytes() */
#define UH_E_TOO_LARGE   (-5)   /* one image exceeds 2^31 bytes, or patch index range */
#define UH_E_CAPTURING   (-6)   /* a synchronising call (uh_dlt_zeroed_pairs) on a stream that is being captured */

/* flags for uh_dlt_* */
#define UH_DLT_SOLVE_F32   0u   /* default: f32 partial-pivot LU, the tf.matrix_solve semantics  */
#define UH_DLT_SOLVE_F64   1u   /* same algorithm carried in f64, result rounded to f32           */
#define UH_DLT_ZERO_NONFINITE_GRAD 8u   /* uh_dlt_backward: a system whose dh4p has a NaN / Inf entry gets dh4p = 0 for all 8.
                                         * Not reference behaviour (tf.matrix_solve raises on a singular system and lets NaN
                                         * through otherwise): the trainer sets it so that ONE pair whose predicted corners are
                                         * degenerate (collinear p2, theta = NaN) contributes no gradient instead of turning every
                                         * variable into NaN -- observed once in 5.4 M pairs (profiles/r02_train_long_*).        */

typedef void* uh_stream_t;     /* hipStream_t */

UH_API int         uh_abi_version(void);
UH_API const char* uh_error_string(int code);

/* ---- Tensor-DLT ------------------------------------------------------------------------------
 * pts1 [B,8] (x,y of 4 corners), h4p [B,8] (predicted corner deltas)  ->  H [B,9] row-major, h33=1.
 * If theta != NULL also writes theta = Minv * H * M  [B,9]; M and Minv are HOST pointers to 9 floats
 * (the constants of homography_model.py:63-72; may be NULL when theta is NULL).                  */
UH_API int uh_dlt_forward(const float* pts1, const float* h4p, float* H, float* theta,
                   const float* M_host, const float* Minv_host,
                   int B, unsigned flags, uh_stream_t stream);

/* Gradient w.r.t. h4p.  Exactly one of dH / dtheta must be non-NULL: with dtheta the kernel first
 * folds dH = Minv^T dtheta M^T.  H is the forward output.                                        */
UH_API int uh_dlt_backward(const float* pts1, const float* h4p, const float* H,
                    const float* dH, const float* dtheta,
                    const float* M_host, const float* Minv_host,
                    float* dh4p, int B, unsigned flags, uh_stream_t stream);

/* How many pairs UH_DLT_ZERO_NONFINITE_GRAD has zeroed on the current device since the last reset (the one deliberate
 * departure from the reference, which lets tf.matrix_solve's NaN through, homography_model.py:242: the trainer prints
 * this so that the guard is never silent).  SYNCHRONOUS on `stream` (only): a one-thread kernel ordered after the work
 * already enqueued there takes the count -- and with reset != 0 clears it -- in one atomic exchange, so no increment is
 * lost between the read and the clear; launches still pending on other streams are counted by the next call.
 * UH_E_CAPTURING when `stream` is being captured, or is the NULL stream while a blocking capture is open (call it at log
 * time, outside any capture).  Concurrent callers are serialised inside (the take passes through one device word).       */
UH_API int uh_dlt_zeroed_pairs(unsigned long long* count, int reset, uh_stream_t stream);

/* ---- Spatial transformer ----------------------------------------------------------------------
 * U [B,H,W,C], theta [B,9] -> out [B,oh,ow,C].  `condition` (device float[1], may be NULL) receives
 * sum over all samples of [|t| > 1e-7]  (tf_spatial_transformer.py:235).                          */
UH_API int uh_warp_forward(const float* U, const float* theta, float* out, float* condition,
                    int B, int H, int W, int C, int oh, int ow, uh_stream_t stream);

/* Validation twin of uh_warp_forward: the literal op-for-op transcription (compiler IEEE division, integer cast and
 * clamps, one pixel per thread, no staging).  Slow; exists so that tests can check the optimised kernel bit for bit on
 * the GPU at full sizes.  Same arguments minus `condition`.                                                         */
UH_API int uh_warp_forward_literal(const float* U, const float* theta, float* out,
                            int B, int H, int W, int C, int oh, int ow, uh_stream_t stream);

UH_API size_t uh_warp_backward_workspace_bytes(int B, int H, int W, int C, int oh, int ow);

/* dTheta [B,9] = d loss/d theta given dOut [B,oh,ow,C].  dU (may be NULL) receives d loss/d U
 * [B,H,W,C] (scatter-add; the reference never requests it).  Deterministic two-stage reduction for
 * dTheta through `workspace` (>= uh_warp_backward_workspace_bytes bytes, device, 8-byte aligned). */
UH_API int uh_warp_backward(const float* U, const float* theta, const float* dOut,
                     float* dTheta, float* dU, void* workspace, size_t workspace_bytes,
                     int B, int H, int W, int C, int oh, int ow, uh_stream_t stream);

/* The same gradient when dOut is the gradient of the gray patch gather (homography_model.py:263-269): dPred [B,PP] and
 * patch_idx [B,PP] stand for the frame gradient dPred[e]/C on the pixels patch_idx names, 0 elsewhere (out_size = H x W).
 * Equals uh_gray_patch_backward -> uh_warp_backward for ANY index set, without materialising that frame: tiles that
 * miss the patch rectangle are skipped, entries that are not at their rectangle position are added one by one.  On the
 * dataloader's rectangles dTheta is bit-identical to the dense chain.                                               */
UH_API size_t uh_warp_patch_backward_workspace_bytes(int B, int H, int W, int C);
UH_API int uh_warp_patch_backward(const float* U, const float* theta, const float* dPred, const int* patch_idx,
                           float* dTheta, void* workspace, size_t workspace_bytes,
                           int B, int H, int W, int C, int PP, uh_stream_t stream);

/* ---- gray + patch gather ------------------------------------------------------------------------
 * warped [B,H,W,C], patch_idx [B,PP] int32 (flat y*W+x within one image) -> pred [B,PP]
 * pred[k,i] = mean_c warped[k, patch_idx[k,i], c]                                                  */
UH_API int uh_gray_patch_forward(const float* warped, const int* patch_idx, float* pred,
                          int B, int H, int W, int C, int PP, uh_stream_t stream);
/* dWarped [B,H,W,C] is fully overwritten: scatter-add of dPred/C into a zero frame (duplicates sum; any index set). */
UH_API int uh_gray_patch_backward(const float* dPred, const int* patch_idx, float* dWarped,
                           int B, int H, int W, int C, int PP, uh_stream_t stream);

/* ---- photometric L1 -------------------------------------------------------------------------------
 * loss[0] = mean |pred - target| over n elements.  workspace: uh_l1_loss_workspace_bytes(n).        */
UH_API size_t uh_l1_loss_workspace_bytes(size_t n);
UH_API int uh_l1_loss_forward(const float* pred, const float* target, float* loss, void* workspace,
                       size_t workspace_bytes, size_t n, uh_stream_t stream);
/* dPred = dLoss[0] * sign(pred - target) / n   (dLoss is a device scalar)                           */
UH_API int uh_l1_loss_backward(const float* pred, const float* target, const float* dLoss, float* dPred,
                        size_t n, uh_stream_t stream);

/* ---- producer of the dataloader's output contract ------------------------------------
 * Decoded uint8 frames I, I' [B,H,W,3] -> the tensors HomographyModel takes (dataloader.py:160-227, 317-375):
 *   aug [B,2,5] (device, may be NULL = no augmentation): per image gamma, brightness, colour r,g,b; value =
 *       clip(v^gamma * brightness * colour_c, 0, 255)                                        dataloader.py:353-375
 *   I_aug, Iprime_aug [B,H,W,3] = (augmented - mean_c)/std_c                                  :172-177
 *   I1, I2 [B,P,P] = channel mean of the NON-augmented standardised frames at the patch; I1_aug, I2_aug the same of
 *       the augmented ones                                                                    :210-227
 *   patch_idx [B,P*P] = (v + y0)*W + (u + x0), (x0, y0) = pts1[b, 0:2]                        :197-207
 * mean3_host / std3_host: HOST pointers to 3 floats (dataloader.py:99-100).                                       */
UH_API int uh_prepare_inputs(const unsigned char* I_u8, const unsigned char* Iprime_u8, const float* aug,
                      const float* pts1, const float* mean3_host, const float* std3_host,
                      float* I_aug, float* Iprime_aug, float* I1, float* I2, float* I1_aug, float* I2_aug,
                      int* patch_idx, int B, int H, int W, int P, uh_stream_t stream);

/* ---- all photometric losses in one pass, and the gradient of the trained one ----------------
 * pred, target [B,P,P] (one channel)  ->  out16[16] (device):
 *   [0] rec_loss  [1] ssim_loss  [2] l1_loss  [3] l1_smooth_loss  [4] ncc_loss        homography_model.py:136-166,286-352
 *   [5] h_loss = sqrt(mean((h4p - gt)^2)) over [B,8] when h4p/gt are given (both or neither), else 0      :288
 *   [6..12] the raw sums behind them: |d|, d^2, smooth-l1, x^2, y^2, x*y, ssim term (x = pred, y = target, d = x - y);
 *   [13..15] reserved (0).                                                                                          */
UH_API size_t uh_patch_losses_workspace_bytes(int B, int P);
UH_API int uh_patch_losses_forward(const float* pred, const float* target, const float* h4p, const float* gt,
                            float* out16, void* workspace, size_t workspace_bytes, int B, int P, uh_stream_t stream);
/* dPred [B,P,P] = dLoss[0] * d loss_kind / d pred  (TF autodiff of the loss expressions above; dLoss is a device scalar;
 * stats16 is the out16 of uh_patch_losses_forward on the same pred / target: it carries the global norms rec and ncc
 * divide by).  kind = index of the loss in out16.                                                                   */
#define UH_LOSS_REC        0
#define UH_LOSS_SSIM       1
#define UH_LOSS_L1         2
#define UH_LOSS_L1_SMOOTH  3
#define UH_LOSS_NCC        4
UH_API int uh_patch_loss_backward(int kind, const float* pred, const float* target, const float* stats16,
                           const float* dLoss, float* dPred, int B, int P, uh_stream_t stream);
/* The two launches either side of the loss, folded into their neighbours (same results, bit for bit, as the calls they
 * replace; they exist because each of the replaced kernels is 5-7 us of launch latency for < 1 us of work):
 *   uh_gather_patch_losses_forward = uh_gray_patch_forward + uh_patch_losses_forward: pred [B,P,P] (written) is the gray
 *     gather of warped [B,H,W,C] at patch_idx [B,P*P], out16 as above.  Workspace: uh_patch_losses_workspace_bytes(B, P).
 *   uh_warp_patch_loss_backward   = uh_patch_loss_backward + uh_warp_patch_backward for the point-wise kinds (REC, L1,
 *     L1_SMOOTH, NCC; UH_E_SHAPE for SSIM, whose gradient is a stencil): dTheta [B,9] straight from (pred, target,
 *     stats16, dLoss), no dPred tensor; dLoss == NULL means 1.  Workspace: uh_warp_patch_backward_workspace_bytes.  */
UH_API int uh_gather_patch_losses_forward(const float* warped, const int* patch_idx, const float* target, const float* h4p,
                                   const float* gt, float* pred, float* out16, void* workspace, size_t workspace_bytes,
                                   int B, int H, int W, int C, int P, uh_stream_t stream);
UH_API int uh_warp_patch_loss_backward(int kind, const float* U, const float* theta, const float* pred, const float* target,
                                const float* stats16, const float* dLoss, const int* patch_idx, float* dTheta,
                                void* workspace, size_t workspace_bytes, int B, int H, int W, int C, int PP,
                                uh_stream_t stream);

/* ---- fused patch path ---------------------------------------------------------
 * For the P x P loss patch only: sample -> gray -> |pred - I2| -> loss, and d loss/d theta for
 * dLoss = 1, in ONE pass that never materialises the warped frame.
 *   U [B,H,W,C], theta [B,9], I2 [B,PP], patch_idx [B,PP]  ->  pred [B,PP], loss[1], dTheta [B,9]
 * dTheta may be NULL (forward only).                                                                */
UH_API size_t uh_warp_patch_l1_workspace_bytes(int B, int PP);
UH_API int uh_warp_patch_l1_fwdbwd(const float* U, const float* theta, const float* I2,
                            const int* patch_idx, float* pred, float* loss, float* dTheta,
                            void* workspace, size_t workspace_bytes,
                            int B, int H, int W, int C, int PP, uh_stream_t stream);

/* ---- the whole l1_loss tail as one call / one hipGraph launch --------------------------
 * h4p -> DLT -> theta -> warp -> gray patch -> L1, and d L1/d h4p for dLoss = 1 (the caller scales it): solve_DLT +
 * transform + the l1 branch of build_losses and their backward             homography_model.py:169-269,321-330
 * A plan fixes the shapes and owns the graph cache (one captured hipGraph per distinct set of pointers + stream,
 * LRU of 8; the key also holds the CONTENTS of M / Minv, and after 8 consecutive misses the chain is enqueued eagerly instead of
 * re-captured -- until one of the last 64 eagerly-run argument sets comes back, which is then captured on that second sighting);
 * intermediates (theta, dtheta, warped, reduction partials) live in the caller's workspace.
 * uh_tail_run is serialised per plan; do not enqueue other work on `stream` from another thread during the call
 * (the first call with a new argument set stream-captures).  With UH_TAIL_GRAPH unset, or while the launch profiler
 * is on, or on the NULL stream, the kernels are enqueued directly.
 *   pts1, h4p [B,8]; U [B,H,W,C]; I2 [B,P*P]; patch_idx [B,P*P]  ->  H [B,9], pred [B,P*P], loss[1], dh4p [B,8]
 *   dh4p may be NULL (forward only).  M_host / Minv_host: HOST 3x3 constants as in uh_dlt_forward.                 */
#define UH_TAIL_FUSED_PATCH  2u   /* use uh_warp_patch_l1_fwdbwd instead of the full-frame warp (no `warped`)      */
#define UH_TAIL_GRAPH        4u   /* replay a captured hipGraph                                                     */
typedef struct uh_tail_plan uh_tail_plan;
UH_API int    uh_tail_create(uh_tail_plan** plan, int B, int H, int W, int C, int P, unsigned flags /* | UH_DLT_SOLVE_F64 | UH_DLT_ZERO_NONFINITE_GRAD */);
UH_API size_t uh_tail_workspace_bytes(const uh_tail_plan* plan);
UH_API size_t uh_tail_warped_offset(const uh_tail_plan* plan);   /* byte offset of `warped` [B,H,W,C] in the workspace; (size_t)-1 when fused */
UH_API int    uh_tail_run(uh_tail_plan* plan, const float* pts1, const float* h4p, const float* U, const float* I2,
                   const int* patch_idx, const float* M_host, const float* Minv_host, float* H, float* pred,
                   float* loss, float* dh4p, void* workspace, size_t workspace_bytes, uh_stream_t stream);
UH_API int    uh_tail_stats(const uh_tail_plan* plan, long long* launches, long long* captures);
UH_API void   uh_tail_destroy(uh_tail_plan* plan);

/* ---- bias + ReLU epilogue of the regressor's conv layers (outside the reference's hot path; conv GEMMs stay MIOpen) --
 * y [npix, C] NHWC activation (npix = N*H*W), C % 4 == 0 and 1024 % C == 0.   homography_model.py:88-95 (_conv2d)
 *   forward : y <- max(y + bias[c], 0) in place; mask <- one bit per element (y > 0) for the backward
 *   backward: g = bit ? gy : 0 ; dbias[c] = sum g          (deterministic two-stage reduction through `workspace`)
 * The backward reads gy + BITS, not the activation (ABI 8; ABI <= 7 re-read y).  `mask` is opaque: uh_relu_mask_bytes /
 * uh_pool_mask_bytes bytes, written by the forward and consumed by the backward of the SAME shape.  Both forwards accept
 * mask == NULL (forward only: no backward will follow).                                                              */
UH_API size_t uh_relu_mask_bytes(size_t npix, int C);
UH_API int    uh_bias_relu_forward(float* y, const float* bias, void* mask, size_t npix, int C, uh_stream_t stream);
UH_API size_t uh_bias_relu_backward_workspace_bytes(size_t npix, int C);
UH_API int    uh_bias_relu_backward(const void* mask, const float* gy, float* g, float* dbias, void* workspace,
                             size_t workspace_bytes, size_t npix, int C, uh_stream_t stream);

/* the same fused with the 2x2/2 max-pool that follows (homography_model.py:102-105): y [N,H,W,C] (H, W even) is only READ
 * (relu(y + b) is never written back at full resolution: the next conv consumes `pooled`); pooled / gpooled [N,H/2,W/2,C];
 * mask <- per pooled element, which of its four window elements receives the gradient (the first maximum, as max_pool2d;
 * none when the maximum is not positive).                                                                             */
UH_API size_t uh_pool_mask_bytes(int N, int H, int W, int C);
UH_API int    uh_bias_relu_pool_forward(const float* y, const float* bias, float* pooled, void* mask, int N, int H, int W,
                                 int C, uh_stream_t stream);
UH_API size_t uh_bias_relu_pool_backward_workspace_bytes(int N, int H, int W, int C);
UH_API int    uh_bias_relu_pool_backward(const void* mask, const float* gpooled, float* g, float* dbias, void* workspace,
                                  size_t workspace_bytes, int N, int H, int W, int C, uh_stream_t stream);

/* ---- in-library kernel timing (used by bench.py for the roofline figure) -----------------------------
 * When enabled, every launch above is bracketed by hipEventRecord on ITS stream; uh_profile_read()
 * synchronises those events and accumulates per-kernel totals.  Not thread-safe while enabled.      */
#define UH_K_DLT_FWD        0
#define UH_K_DLT_BWD        1
#define UH_K_WARP_FWD       2
#define UH_K_WARP_BWD       3   /* per-tile partial kernel (the bandwidth kernel)                    */
#define UH_K_WARP_BWD_FIN   4   /* tiny finishing reduction                                          */
#define UH_K_GRAY_FWD       5
#define UH_K_GRAY_BWD       6
#define UH_K_L1_FWD         7
#define UH_K_L1_BWD         8
#define UH_K_PATCH_FUSED    9
#define UH_K_PATCH_FIN     10
#define UH_K_LOSSES        11
#define UH_K_LOSSES_FIN    12
#define UH_K_PREPARE       13
#define UH_K_EPI_FWD       14
#define UH_K_EPI_BWD       15
#define UH_K_LOSS_BWD      16
#define UH_K_COUNT         17
/* on = 0: off; 1: time every kernel; otherwise a mask: bit (k + 1) set = time kernel UH_K_k only (timing a dispatch
 * costs a few us of pipeline bubble, so a throughput run times just the kernels it reports).  Returns the previous
 * on/off state; resets the counters.                                                                              */
UH_API int uh_profile_enable(int on);
UH_API int uh_profile_read(double* total_ms /*[UH_K_COUNT]*/, long long* launches /*[UH_K_COUNT]*/);
UH_API const char* uh_kernel_name(int k);

#ifdef __cplusplus
}
#endif
#endif /* UH_HOTPATH_H */
