"""Host-side operators over the C ABI (include/uh_hotpath.h).

torch is used for device memory, streams and autograd bookkeeping only: every number is produced by
the HIP kernels in csrc/.  All tensors must live on a HIP device ("cuda" in torch-ROCm); passing CPU
tensors raises -- there is no CPU fallback.

Mirrors, by name and argument meaning:
  transformer(U, theta, out_size)       /root/reference/code/utils/tf_spatial_transformer.py:18
  solve_dlt(pts1, h4p)                  /root/reference/code/homography_model.py:169-250
  warp_gather(U, theta, idx, P)         transform(), homography_model.py:257-269 (full-frame warp + gray gather, sparse backward)
  warp_patch / warp_patch_l1            the same restricted to the loss patch (no warped frame)
  patch_losses(pred, I2, h4p, gt, train)  build_losses(), homography_model.py:286-352 (six values, HIP gradient of the trained one)
  photometric_tail(...)                 solve_DLT + transform + l1 loss and their backward as one library call / hipGraph
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.UHError('%s must be a tensor on the HIP device (got %s); the hot path has no CPU fallback'
                           % (name, getattr(t, 'device', type(t))))
    if t.dtype != torch.float32:
        raise _lib.UHError('%s must be float32 (got %s)' % (name, t.dtype))
    return t.contiguous()


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _host9(a):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(9))
    return a, a.ctypes.data_as(C.c_void_p)


def m_and_minv(img_w, img_h):
    """The reference's constants (homography_model.py:63-69): M in f32, M^-1 by np.linalg.inv."""
    M = np.array([[img_w / 2.0, 0., img_w / 2.0],
                  [0., img_h / 2.0, img_h / 2.0],
                  [0., 0., 1.]]).astype(np.float32)
    return M, np.linalg.inv(M)


# ------------------------------------------------------------------------------------------------
class _DLTSolve(torch.autograd.Function):
    """pts1, h4p -> (H [B,3,3], theta [B,3,3] or None).  Gradient flows to h4p only (pts1 is data)."""

    @staticmethod
    def forward(ctx, pts1, h4p, M, Minv, flags):
        lib = _lib.load()
        pts1 = _f32(pts1, 'pts1').reshape(-1, 8)
        h4p = _f32(h4p, 'h4p').reshape(-1, 8)
        B = pts1.shape[0]
        if h4p.shape[0] != B:
            raise _lib.UHError('pts1 and h4p batch mismatch')
        H = torch.empty((B, 3, 3), dtype=torch.float32, device=pts1.device)
        theta = torch.empty_like(H) if M is not None else None
        Mh = Mih = None
        if M is not None:
            ctx.M_keep, Mh = _host9(M)
            ctx.Mi_keep, Mih = _host9(Minv)
        _lib.check(lib.uh_dlt_forward(_ptr(pts1), _ptr(h4p), _ptr(H), _ptr(theta), Mh, Mih, B, flags, _stream()),
                   'uh_dlt_forward')
        ctx.save_for_backward(pts1, h4p, H)
        ctx.set_materialize_grads(False)       # an unused output arrives as None, not as zeros
        ctx.flags = flags
        ctx.has_theta = M is not None
        if theta is None:
            return H
        return H, theta

    @staticmethod
    def backward(ctx, dH, dtheta=None):
        lib = _lib.load()
        pts1, h4p, H = ctx.saved_tensors
        B = pts1.shape[0]
        dh4p = torch.empty((B, 8), dtype=torch.float32, device=pts1.device)

        def run(gH, gT):
            Mh = Mih = None
            if gT is not None:      # ctx.*_keep own the host buffers for the duration of the call
                Mh = ctx.M_keep.ctypes.data_as(C.c_void_p)
                Mih = ctx.Mi_keep.ctypes.data_as(C.c_void_p)
            _lib.check(lib.uh_dlt_backward(_ptr(pts1), _ptr(h4p), _ptr(H), _ptr(gH), _ptr(gT), Mh, Mih,
                                           _ptr(dh4p), B, ctx.flags, _stream()), 'uh_dlt_backward')

        if dH is None and dtheta is None:
            return None, torch.zeros_like(dh4p), None, None, None
        if dH is None:
            run(None, _f32(dtheta, 'dtheta'))
        elif dtheta is not None:
            # both outputs used downstream: dH_total = dH + Minv^T dtheta M^T  (rare; monitoring code)
            Mt = torch.from_numpy(ctx.M_keep.reshape(3, 3)).to(dH.device)
            Mit = torch.from_numpy(ctx.Mi_keep.reshape(3, 3)).to(dH.device)
            run((dH + Mit.t() @ dtheta @ Mt.t()).contiguous(), None)
        else:
            run(_f32(dH, 'dH'), None)
        return None, dh4p, None, None, None


def solve_dlt(pts1, h4p, img_w=None, img_h=None, solve_f64=False, zero_nonfinite_grad=False):
    """Tensor-DLT.  Returns H [B,3,3]; with img_w/img_h also theta = M^-1 H M (transform() :254)."""
    flags = _lib.UH_DLT_SOLVE_F64 if solve_f64 else _lib.UH_DLT_SOLVE_F32
    if zero_nonfinite_grad:          # a pair with degenerate predicted corners contributes no gradient (see the header)
        flags |= _lib.UH_DLT_ZERO_NONFINITE_GRAD
    if img_w is None:
        return _DLTSolve.apply(pts1, h4p, None, None, flags)
    M, Minv = m_and_minv(img_w, img_h)
    return _DLTSolve.apply(pts1, h4p, M, Minv, flags)


# ------------------------------------------------------------------------------------------------
class _ProjectiveWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, U, theta, out_h, out_w, want_condition):
        lib = _lib.load()
        U = _f32(U, 'U')
        if U.dim() != 4:
            raise _lib.UHError('U must be [B,H,W,C]')
        B, H, W, Cc = U.shape
        theta = _f32(theta, 'theta').reshape(-1, 9)
        if theta.shape[0] != B:
            raise _lib.UHError('theta must be [B,3,3] / [B,9]')
        out = torch.empty((B, out_h, out_w, Cc), dtype=torch.float32, device=U.device)
        cond = torch.empty((1,), dtype=torch.float32, device=U.device) if want_condition else None
        _lib.check(lib.uh_warp_forward(_ptr(U), _ptr(theta), _ptr(out), _ptr(cond), B, H, W, Cc, out_h, out_w,
                                       _stream()), 'uh_warp_forward')
        ctx.save_for_backward(U, theta)
        ctx.dims = (B, H, W, Cc, out_h, out_w)
        if want_condition:
            ctx.mark_non_differentiable(cond)
            return out, cond
        return out

    @staticmethod
    def backward(ctx, dOut, dcond=None):
        lib = _lib.load()
        U, theta = ctx.saved_tensors
        B, H, W, Cc, oh, ow = ctx.dims
        dOut = _f32(dOut, 'dOut')
        dTheta = torch.empty((B, 9), dtype=torch.float32, device=U.device)
        dU = torch.empty_like(U) if ctx.needs_input_grad[0] else None
        nbytes = lib.uh_warp_backward_workspace_bytes(B, H, W, Cc, oh, ow)
        ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=U.device)
        _lib.check(lib.uh_warp_backward(_ptr(U), _ptr(theta), _ptr(dOut), _ptr(dTheta), _ptr(dU), _ptr(ws), nbytes,
                                        B, H, W, Cc, oh, ow, _stream()), 'uh_warp_backward')
        return dU, dTheta, None, None, None


def transformer(U, theta, out_size, name='SpatialTransformer', with_condition=True, **kwargs):
    """Spatial Transformer Layer -- same signature and return as the reference's
    transformer(U, theta, out_size) -> (output, condition)  (tf_spatial_transformer.py:18,249-251).
    theta may be [B,3,3], [B,9] ... anything reshapable to (-1,3,3) (:190)."""
    res = _ProjectiveWarp.apply(U, theta.reshape(-1, 9), int(out_size[0]), int(out_size[1]), with_condition)
    if with_condition:
        return res[0], res[1][0]
    return res, None


def transformer_literal(U, theta, out_size):
    """Validation twin of transformer(): literal op order, compiler division (uh_warp_forward_literal).  No gradient."""
    lib = _lib.load()
    U = _f32(U.detach(), 'U')
    B, H, W, Cc = U.shape
    theta = _f32(theta.detach(), 'theta').reshape(-1, 9)
    out = torch.empty((B, int(out_size[0]), int(out_size[1]), Cc), dtype=torch.float32, device=U.device)
    _lib.check(lib.uh_warp_forward_literal(_ptr(U), _ptr(theta), _ptr(out), B, H, W, Cc, int(out_size[0]),
                                           int(out_size[1]), _stream()), 'uh_warp_forward_literal')
    return out


# ------------------------------------------------------------------------------------------------
class _GrayPatch(torch.autograd.Function):
    @staticmethod
    def forward(ctx, warped, patch_idx, patch_size):
        lib = _lib.load()
        warped = _f32(warped, 'warped')
        B, H, W, Cc = warped.shape
        if patch_idx.dtype != torch.int32 or not patch_idx.is_cuda:
            raise _lib.UHError('patch_indices must be an int32 tensor on the HIP device')
        idx = patch_idx.contiguous().reshape(B, -1)
        PP = idx.shape[1]
        pred = torch.empty((B, PP), dtype=torch.float32, device=warped.device)
        _lib.check(lib.uh_gray_patch_forward(_ptr(warped), _ptr(idx), _ptr(pred), B, H, W, Cc, PP, _stream()),
                   'uh_gray_patch_forward')
        ctx.save_for_backward(idx)
        ctx.dims = (B, H, W, Cc, PP)
        return pred.reshape(B, patch_size, patch_size, 1)

    @staticmethod
    def backward(ctx, dPred):
        lib = _lib.load()
        (idx,) = ctx.saved_tensors
        B, H, W, Cc, PP = ctx.dims
        dPred = _f32(dPred, 'dPred')
        dW = torch.empty((B, H, W, Cc), dtype=torch.float32, device=dPred.device)
        _lib.check(lib.uh_gray_patch_backward(_ptr(dPred), _ptr(idx), _ptr(dW), B, H, W, Cc, PP, _stream()),
                   'uh_gray_patch_backward')
        return dW, None, None


class _WarpGather(torch.autograd.Function):
    """transform() of the reference as ONE autograd node: warped = transformer(U, theta, (H, W)); pred_I2 = gather(gray(
    warped), patch_indices)  (homography_model.py:257-269).  The forward materialises `warped` exactly like the reference
    (uh_warp_forward + uh_gray_patch_forward).  The backward does NOT build the 79 %-zero [B,H,W,C] gradient frame of
    tf.gather's scatter and stream it through the dense warp backward: uh_warp_patch_backward takes dPred [B,PP] and
    the indices, skips the tiles outside the patch rectangle and is bit-identical to the dense chain on rectangles."""

    @staticmethod
    def forward(ctx, U, theta, patch_idx, patch_size):
        lib = _lib.load()
        U = _f32(U, 'U'); theta9 = _f32(theta, 'theta').reshape(-1, 9)
        B, H, W, Cc = U.shape
        if patch_idx.dtype != torch.int32 or not patch_idx.is_cuda:
            raise _lib.UHError('patch_indices must be an int32 tensor on the HIP device')
        idx = patch_idx.contiguous().reshape(B, -1)
        PP = idx.shape[1]
        warped = torch.empty((B, H, W, Cc), dtype=torch.float32, device=U.device)
        _lib.check(lib.uh_warp_forward(_ptr(U), _ptr(theta9), _ptr(warped), None, B, H, W, Cc, H, W, _stream()),
                   'uh_warp_forward')
        pred = torch.empty((B, PP), dtype=torch.float32, device=U.device)
        _lib.check(lib.uh_gray_patch_forward(_ptr(warped), _ptr(idx), _ptr(pred), B, H, W, Cc, PP, _stream()),
                   'uh_gray_patch_forward')
        ctx.save_for_backward(U, theta9, idx)
        ctx.dims = (B, H, W, Cc, PP)
        ctx.theta_shape = theta.shape
        ctx.mark_non_differentiable(warped)
        return warped, pred.reshape(B, patch_size, patch_size, 1)

    @staticmethod
    def backward(ctx, dWarped, dPred):
        lib = _lib.load()
        U, theta9, idx = ctx.saved_tensors
        B, H, W, Cc, PP = ctx.dims
        if not ctx.needs_input_grad[1]:
            return None, None, None, None
        dPred = _f32(dPred, 'dPred')
        dT = torch.empty((B, 9), dtype=torch.float32, device=U.device)
        nbytes = lib.uh_warp_patch_backward_workspace_bytes(B, H, W, Cc)
        ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=U.device)
        _lib.check(lib.uh_warp_patch_backward(_ptr(U), _ptr(theta9), _ptr(dPred), _ptr(idx), _ptr(dT), _ptr(ws), nbytes,
                                              B, H, W, Cc, PP, _stream()), 'uh_warp_patch_backward')
        return None, dT.reshape(ctx.theta_shape), None, None


def warp_gather(U, theta, patch_indices, patch_size):
    """-> (warped [B,H,W,C] (no gradient), pred_I2 [B,P,P,1]); d pred_I2 / d theta by the sparse warp backward."""
    return _WarpGather.apply(U, theta, patch_indices, patch_size)


class _WarpGatherLosses(torch.autograd.Function):
    """transform() + build_losses() of the reference as ONE autograd node with the fewest launches the un-fused path
    allows (homography_model.py:257-269, 286-352): forward = uh_warp_forward (the full `warped` frame, as the reference
    materialises it) + uh_gather_patch_losses_forward (gray + gather + all six loss values); backward =
    uh_warp_patch_loss_backward (the loss gradient is formed inside the sparse warp backward: no dPred tensor).  Same
    values, bit for bit, as warp_gather -> patch_losses (tests/test_gpu_parity) in 3 + 2 kernels instead of 4 + 3.  SSIM's gradient is a stencil: it keeps
    uh_patch_loss_backward."""

    @staticmethod
    def forward(ctx, U, theta, patch_idx, patch_size, target, h4p, gt, kind):
        lib = _lib.load()
        U = _f32(U, 'U'); theta9 = _f32(theta, 'theta').reshape(-1, 9)
        target = _f32(target, 'target')
        B, H, W, Cc = U.shape
        P = int(patch_size)
        if patch_idx.dtype != torch.int32 or not patch_idx.is_cuda:
            raise _lib.UHError('patch_indices must be an int32 tensor on the HIP device')
        idx = patch_idx.contiguous().reshape(B, -1)
        if idx.shape[1] != P * P or target.numel() != B * P * P:
            raise _lib.UHError('patch_indices / target must hold B x P*P entries')
        warped = torch.empty((B, H, W, Cc), dtype=torch.float32, device=U.device)
        _lib.check(lib.uh_warp_forward(_ptr(U), _ptr(theta9), _ptr(warped), None, B, H, W, Cc, H, W, _stream()),
                   'uh_warp_forward')
        pred = torch.empty((B, P, P, 1), dtype=torch.float32, device=U.device)
        out = torch.empty((16,), dtype=torch.float32, device=U.device)
        nbytes = lib.uh_patch_losses_workspace_bytes(B, P)
        ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=U.device)
        _lib.check(lib.uh_gather_patch_losses_forward(_ptr(warped), _ptr(idx), _ptr(target), _ptr(h4p), _ptr(gt), _ptr(pred),
                                                      _ptr(out), _ptr(ws), nbytes, B, H, W, Cc, P, _stream()),
                   'uh_gather_patch_losses_forward')
        ctx.kind = kind
        ctx.dims = (B, H, W, Cc, P)
        ctx.theta_shape = theta.shape
        if kind >= 0:
            ctx.save_for_backward(U, theta9, idx, pred, target, out)
        ctx.mark_non_differentiable(warped, pred)
        return warped, pred, out[:6]

    @staticmethod
    def backward(ctx, dWarped, dPred, dOut):
        none = (None,) * 8
        if ctx.kind < 0 or not ctx.needs_input_grad[1]:
            return none
        lib = _lib.load()
        U, theta9, idx, pred, target, stats = ctx.saved_tensors
        B, H, W, Cc, P = ctx.dims
        g = _f32(dOut[ctx.kind:ctx.kind + 1], 'dLoss')
        dT = torch.empty((B, 9), dtype=torch.float32, device=U.device)
        nbytes = lib.uh_warp_patch_backward_workspace_bytes(B, H, W, Cc)
        ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=U.device)
        if ctx.kind == _lib.LOSS_KINDS['ssim_loss']:
            dP = torch.empty_like(pred)
            _lib.check(lib.uh_patch_loss_backward(ctx.kind, _ptr(pred), _ptr(target), _ptr(stats), _ptr(g), _ptr(dP), B, P,
                                                  _stream()), 'uh_patch_loss_backward')
            _lib.check(lib.uh_warp_patch_backward(_ptr(U), _ptr(theta9), _ptr(dP), _ptr(idx), _ptr(dT), _ptr(ws), nbytes,
                                                  B, H, W, Cc, P * P, _stream()), 'uh_warp_patch_backward')
        else:
            _lib.check(lib.uh_warp_patch_loss_backward(ctx.kind, _ptr(U), _ptr(theta9), _ptr(pred), _ptr(target), _ptr(stats),
                                                       _ptr(g), _ptr(idx), _ptr(dT), _ptr(ws), nbytes, B, H, W, Cc, P * P,
                                                       _stream()), 'uh_warp_patch_loss_backward')
        return (None, dT.reshape(ctx.theta_shape)) + (None,) * 6


def warp_gather_losses(U, theta, patch_indices, patch_size, target, h4p=None, gt=None, train=None):
    """-> (warped [B,H,W,C], pred_I2 [B,P,P,1], losses [6] = rec, ssim, l1, l1_smooth, ncc, h_loss).  `train` names the
    loss being trained on: that element carries d/d theta; warped, pred_I2 and the other five are stop_gradient values
    (use warp_gather + patch_losses when a gradient through pred_I2 itself is needed)."""
    B = U.shape[0]
    if h4p is not None:
        h4p = _f32(h4p.detach(), 'h4p').reshape(-1); gt = _f32(gt.detach(), 'gt').reshape(-1)
        if h4p.numel() != B * 8 or gt.numel() != B * 8:
            raise _lib.UHError('h4p and gt must be [B,8]')
    kind = -1
    if train is not None:
        if train not in _lib.LOSS_KINDS:
            raise _lib.UHError('warp_gather_losses: no gradient kernel for %r' % (train,))
        kind = _lib.LOSS_KINDS[train]
    if kind < 0:
        theta = theta.detach()
    return _WarpGatherLosses.apply(U, theta, patch_indices, patch_size, target.detach(), h4p, gt, kind)


def gray_patch_gather(warped, patch_indices, patch_size):
    """reduce_mean(axis=3) + flat gather (homography_model.py:263-269) -> [B,P,P,1]."""
    return _GrayPatch.apply(warped, patch_indices, patch_size)


class _L1Loss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        lib = _lib.load()
        pred = _f32(pred, 'pred'); target = _f32(target, 'target')
        n = pred.numel()
        if target.numel() != n:
            raise _lib.UHError('pred/target size mismatch')
        loss = torch.empty((1,), dtype=torch.float32, device=pred.device)
        nbytes = lib.uh_l1_loss_workspace_bytes(n)
        ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=pred.device)
        _lib.check(lib.uh_l1_loss_forward(_ptr(pred), _ptr(target), _ptr(loss), _ptr(ws), nbytes, n, _stream()),
                   'uh_l1_loss_forward')
        ctx.save_for_backward(pred, target)
        return loss[0]

    @staticmethod
    def backward(ctx, dLoss):
        lib = _lib.load()
        pred, target = ctx.saved_tensors
        g = _f32(dLoss.reshape(1), 'dLoss')
        dPred = torch.empty_like(pred)
        _lib.check(lib.uh_l1_loss_backward(_ptr(pred), _ptr(target), _ptr(g), _ptr(dPred), pred.numel(), _stream()),
                   'uh_l1_loss_backward')
        return dPred, None


def l1_loss(pred, target):
    """reduce_mean(abs(pred - target)) (homography_model.py:328)."""
    return _L1Loss.apply(pred, target)


class _PatchLosses(torch.autograd.Function):
    """(pred, target[, h4p, gt]) -> [6] (rec, ssim, l1, l1_smooth, ncc, h_loss) in one launch; element `kind` (the loss
    being trained on) carries d/d pred through uh_patch_loss_backward, the others are stop_gradient monitors exactly as
    in build_losses() (homography_model.py:286-352)."""

    @staticmethod
    def forward(ctx, pred, target, h4p, gt, kind):
        lib = _lib.load()
        pred = _f32(pred, 'pred'); target = _f32(target, 'target')
        B, P = pred.shape[0], pred.shape[1]
        out = torch.empty((16,), dtype=torch.float32, device=pred.device)
        nbytes = lib.uh_patch_losses_workspace_bytes(B, P)
        ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=pred.device)
        _lib.check(lib.uh_patch_losses_forward(_ptr(pred), _ptr(target), _ptr(h4p), _ptr(gt), _ptr(out), _ptr(ws), nbytes,
                                               B, P, _stream()), 'uh_patch_losses_forward')
        ctx.kind = kind
        ctx.dims = (B, P)
        if kind >= 0:
            ctx.save_for_backward(pred, target, out)
        return out[:6]

    @staticmethod
    def backward(ctx, dOut):
        if ctx.kind < 0 or not ctx.needs_input_grad[0]:
            return None, None, None, None, None
        lib = _lib.load()
        pred, target, stats = ctx.saved_tensors
        B, P = ctx.dims
        g = _f32(dOut[ctx.kind:ctx.kind + 1], 'dLoss')
        dPred = torch.empty_like(pred)
        _lib.check(lib.uh_patch_loss_backward(ctx.kind, _ptr(pred), _ptr(target), _ptr(stats), _ptr(g), _ptr(dPred), B, P,
                                              _stream()), 'uh_patch_loss_backward')
        return dPred, None, None, None, None


def patch_losses(pred, target, h4p=None, gt=None, train=None):
    """All photometric losses of build_losses() in one launch (homography_model.py:136-166, 286-352): returns a [6]
    tensor (rec, ssim, l1, l1_smooth, ncc, h_loss); h_loss = 0 unless h4p and gt are given.  `train` names the loss
    that is being trained on ('rec_loss', 'ssim_loss', 'l1_loss', 'l1_smooth_loss', 'ncc_loss'): that element is
    differentiable w.r.t. pred (HIP gradient kernel); every other element is a stop_gradient monitor."""
    if pred.dim() != 4 or pred.shape[3] != 1 or pred.shape[1] != pred.shape[2] or target.shape != pred.shape:
        raise _lib.UHError('patch_losses expects pred and target of shape [B,P,P,1]')
    B = pred.shape[0]
    if h4p is not None:
        h4p = _f32(h4p.detach(), 'h4p').reshape(-1); gt = _f32(gt.detach(), 'gt').reshape(-1)
        if h4p.numel() != B * 8 or gt.numel() != B * 8:
            raise _lib.UHError('h4p and gt must be [B,8]')
    kind = -1
    if train is not None:
        if train not in _lib.LOSS_KINDS:
            raise _lib.UHError('patch_losses: no gradient kernel for %r' % (train,))
        kind = _lib.LOSS_KINDS[train]
    if kind < 0:
        pred = pred.detach()
    return _PatchLosses.apply(pred, target.detach(), h4p, gt, kind)


# ------------------------------------------------------------------------------------------------
class _WarpPatchL1(torch.autograd.Function):
    """Fused patch path: (U, theta, I2, patch_idx) -> (loss, pred_I2).  The kernel produces dTheta for
    dLoss = 1 in the same pass; backward just scales it."""

    @staticmethod
    def forward(ctx, U, theta, I2, patch_idx, patch_size):
        lib = _lib.load()
        U = _f32(U, 'U'); I2 = _f32(I2, 'I2')
        B, H, W, Cc = U.shape
        theta = _f32(theta, 'theta').reshape(-1, 9)
        if patch_idx.dtype != torch.int32 or not patch_idx.is_cuda:
            raise _lib.UHError('patch_indices must be an int32 tensor on the HIP device')
        idx = patch_idx.contiguous().reshape(B, -1)
        PP = idx.shape[1]
        pred = torch.empty((B, PP), dtype=torch.float32, device=U.device)
        loss = torch.empty((1,), dtype=torch.float32, device=U.device)
        need_grad = ctx.needs_input_grad[1]
        dTheta = torch.empty((B, 9), dtype=torch.float32, device=U.device) if need_grad else None
        nbytes = lib.uh_warp_patch_l1_workspace_bytes(B, PP)
        ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=U.device)
        _lib.check(lib.uh_warp_patch_l1_fwdbwd(_ptr(U), _ptr(theta), _ptr(I2), _ptr(idx), _ptr(pred), _ptr(loss),
                                               _ptr(dTheta), _ptr(ws), nbytes, B, H, W, Cc, PP, _stream()),
                   'uh_warp_patch_l1_fwdbwd')
        ctx.dTheta = dTheta
        pred = pred.reshape(B, patch_size, patch_size, 1)
        ctx.mark_non_differentiable(pred)
        return loss[0], pred

    @staticmethod
    def backward(ctx, dLoss, dPred=None):
        return None, (ctx.dTheta * dLoss).reshape(-1, 9), None, None, None


class _WarpPatch(torch.autograd.Function):
    """Fused patch path for the losses that need a global norm before their gradient exists (rec, ssim, smooth-l1, ncc):
    pred_I2 = gather(gray(warp(U, theta)), patch_indices) sampled on the patch only (the warped frame never exists);
    d pred_I2 / d theta by uh_warp_patch_backward once the loss kernel has produced dPred."""

    @staticmethod
    def forward(ctx, U, theta, I2, patch_idx, patch_size):
        lib = _lib.load()
        U = _f32(U, 'U'); I2 = _f32(I2, 'I2')
        B, H, W, Cc = U.shape
        theta9 = _f32(theta, 'theta').reshape(-1, 9)
        if patch_idx.dtype != torch.int32 or not patch_idx.is_cuda:
            raise _lib.UHError('patch_indices must be an int32 tensor on the HIP device')
        idx = patch_idx.contiguous().reshape(B, -1)
        PP = idx.shape[1]
        pred = torch.empty((B, PP), dtype=torch.float32, device=U.device)
        loss = torch.empty((1,), dtype=torch.float32, device=U.device)
        nbytes = lib.uh_warp_patch_l1_workspace_bytes(B, PP)
        ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=U.device)
        _lib.check(lib.uh_warp_patch_l1_fwdbwd(_ptr(U), _ptr(theta9), _ptr(I2), _ptr(idx), _ptr(pred), _ptr(loss), None,
                                               _ptr(ws), nbytes, B, H, W, Cc, PP, _stream()), 'uh_warp_patch_l1_fwdbwd')
        ctx.save_for_backward(U, theta9, idx)
        ctx.dims = (B, H, W, Cc, PP)
        ctx.theta_shape = theta.shape
        return pred.reshape(B, patch_size, patch_size, 1)

    @staticmethod
    def backward(ctx, dPred):
        lib = _lib.load()
        U, theta9, idx = ctx.saved_tensors
        B, H, W, Cc, PP = ctx.dims
        if not ctx.needs_input_grad[1]:
            return None, None, None, None, None
        dPred = _f32(dPred, 'dPred')
        dT = torch.empty((B, 9), dtype=torch.float32, device=U.device)
        nbytes = lib.uh_warp_patch_backward_workspace_bytes(B, H, W, Cc)
        ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=U.device)
        _lib.check(lib.uh_warp_patch_backward(_ptr(U), _ptr(theta9), _ptr(dPred), _ptr(idx), _ptr(dT), _ptr(ws), nbytes,
                                              B, H, W, Cc, PP, _stream()), 'uh_warp_patch_backward')
        return None, dT.reshape(ctx.theta_shape), None, None, None


def warp_patch(U, theta, I2, patch_indices, patch_size):
    """-> pred_I2 [B,P,P,1] sampled on the patch only; differentiable w.r.t. theta (sparse warp backward)."""
    return _WarpPatch.apply(U, theta, I2, patch_indices, patch_size)


def warp_patch_l1(U, theta, I2, patch_indices, patch_size):
    """-> (l1_loss scalar, pred_I2 [B,P,P,1]); pred_I2 carries no gradient on this path."""
    return _WarpPatchL1.apply(U, theta.reshape(-1, 9), I2, patch_indices, patch_size)


# ------------------------------------------------------------------------------------------------
class TailPlan(object):
    """uh_tail_plan + its persistent workspace for one shape.  Keeping the workspace (and
    therefore theta / warped / dPred ...) at fixed addresses is what lets the captured hipGraph be replayed."""
    _cache = {}

    def __init__(self, B, H, W, Cc, P, device, fused_patch=False, graph=True, solve_f64=False, zero_nonfinite_grad=False):
        lib = _lib.load()
        self.dims = (B, H, W, Cc, P)
        self.fused = bool(fused_patch)
        flags = (_lib.UH_TAIL_FUSED_PATCH if fused_patch else 0) | (_lib.UH_TAIL_GRAPH if graph else 0) \
            | (_lib.UH_DLT_SOLVE_F64 if solve_f64 else 0) | (_lib.UH_DLT_ZERO_NONFINITE_GRAD if zero_nonfinite_grad else 0)
        h = C.c_void_p()
        _lib.check(lib.uh_tail_create(C.byref(h), B, H, W, Cc, P, flags), 'uh_tail_create')
        self.handle = h
        self.nbytes = lib.uh_tail_workspace_bytes(h)
        self.ws = torch.empty((self.nbytes,), dtype=torch.uint8, device=device)
        self.M, self.Minv = m_and_minv(W, H)
        self._Mh = np.ascontiguousarray(self.M.reshape(9), np.float32)
        self._Mih = np.ascontiguousarray(self.Minv.reshape(9).astype(np.float32))
        # plan-owned I/O buffers: fixed addresses = one captured graph serves every step (the results are valid until
        # the next run of this plan)
        f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=device)
        self.h4p_in, self.H_out, self.pred_out = f(B, 8), f(B, 3, 3), f(B, P, P, 1)
        self.loss_out, self.dh4p_out = f(1), f(B, 8)
        off = lib.uh_tail_warped_offset(h)
        self.warped = None
        if not self.fused:
            n = B * H * W * Cc * 4
            self.warped = self.ws[off:off + n].view(torch.float32).view(B, H, W, Cc)

    def stats(self):
        a = C.c_longlong(); b = C.c_longlong()
        _lib.check(_lib.load().uh_tail_stats(self.handle, C.byref(a), C.byref(b)), 'uh_tail_stats')
        return {'launches': a.value, 'captures': b.value}

    def __del__(self):
        try:
            if self.handle:
                _lib.load().uh_tail_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    @classmethod
    def get(cls, B, H, W, Cc, P, device, fused_patch=False, graph=True, solve_f64=False, zero_nonfinite_grad=False):
        key = (B, H, W, Cc, P, str(device), bool(fused_patch), bool(graph), bool(solve_f64), bool(zero_nonfinite_grad))
        if key not in cls._cache:
            cls._cache[key] = cls(B, H, W, Cc, P, device, fused_patch, graph, solve_f64, zero_nonfinite_grad)
        return cls._cache[key]


class _PhotometricTail(torch.autograd.Function):
    """(pts1, h4p, U, I2, patch_idx) -> (l1_loss, pred_I2, H) in ONE library call; d loss/d h4p for dLoss = 1 is
    produced by the same call and scaled in backward."""

    @staticmethod
    def forward(ctx, pts1, h4p, U, I2, patch_idx, plan):
        lib = _lib.load()
        B, H, W, Cc, P = plan.dims
        pts1 = _f32(pts1, 'pts1').reshape(B, 8); h4p = _f32(h4p, 'h4p').reshape(B, 8)
        U = _f32(U, 'U'); I2 = _f32(I2, 'I2')
        if tuple(U.shape) != (B, H, W, Cc) or I2.numel() != B * P * P:
            raise _lib.UHError('photometric_tail: tensors do not match the plan %s' % (plan.dims,))
        if patch_idx.dtype != torch.int32 or not patch_idx.is_cuda:
            raise _lib.UHError('patch_indices must be an int32 tensor on the HIP device')
        idx = patch_idx.contiguous()
        plan.h4p_in.copy_(h4p)                   # the regressor's output lands at a fixed address
        h4p = plan.h4p_in
        Hm, pred, loss = plan.H_out, plan.pred_out, plan.loss_out
        need = ctx.needs_input_grad[1]
        dh4p = plan.dh4p_out if need else None
        _lib.check(lib.uh_tail_run(plan.handle, _ptr(pts1), _ptr(h4p), _ptr(U), _ptr(I2), _ptr(idx),
                                   plan._Mh.ctypes.data_as(C.c_void_p), plan._Mih.ctypes.data_as(C.c_void_p), _ptr(Hm),
                                   _ptr(pred), _ptr(loss), _ptr(dh4p), C.c_void_p(plan.ws.data_ptr()), plan.nbytes,
                                   _stream()), 'uh_tail_run')
        # The plan's buffers are overwritten by the NEXT run of the same plan (TailPlan.get caches per shape, so a second
        # tower or a gradient-accumulation micro-step shares it).  What autograd and the caller keep must not alias
        # them: hand out copies (dh4p, loss, H are tiny; pred is B*P*P floats).  plan.warped stays plan-owned and is
        # documented as valid until the next run.
        ctx.dh4p = dh4p.clone() if dh4p is not None else None
        pred = pred.clone(); Hm = Hm.clone(); loss = loss.clone()
        ctx.mark_non_differentiable(pred, Hm)
        return loss[0], pred, Hm

    @staticmethod
    def backward(ctx, dLoss, dPred=None, dH=None):
        return None, (ctx.dh4p * dLoss if ctx.dh4p is not None else None), None, None, None, None


def photometric_tail(pts1, h4p, U, I2, patch_indices, patch_size, fused_patch=False, graph=True, solve_f64=False,
                     zero_nonfinite_grad=False):
    """-> (l1_loss, pred_I2 [B,P,P,1], H_mat [B,3,3], plan).  plan.warped is the warped frame (full-frame mode)."""
    B, H, W, Cc = U.shape
    plan = TailPlan.get(B, H, W, Cc, int(patch_size), U.device, fused_patch, graph, solve_f64, zero_nonfinite_grad)
    loss, pred, Hm = _PhotometricTail.apply(pts1, h4p, U, I2, patch_indices, plan)
    return loss, pred, Hm, plan


# ------------------------------------------------------------------------------------------------
class _ConvBiasReLU(torch.autograd.Function):
    """relu(conv2d(x, w) + b) with the conv on stock MIOpen and the bias+ReLU epilogue (and its backward, including
    the bias gradient) as one HIP pass each (csrc/uh_epilogue.hip).  Activations are channels_last (NHWC storage).  The
    backward's ReLU mask is kept as one bit per element: it reads gy + bits, not gy + y (12 -> 8 B/element).
    (PyTorch-ROCm's fused aten::miopen_convolution_relu was measured as a replacement for conv + this pass in round 5:
    miopenStatusUnknownError on channels_last inputs at batch 64, and 0.8 - 2.3 x the time of this route on NCHW copies:
    profiles/r05_conv_relu_probe.jsonl)"""

    @staticmethod
    def forward(ctx, x, weight, bias, padding):
        lib = _lib.load()
        y = torch.nn.functional.conv2d(x, weight, None, 1, padding)
        if not y.is_contiguous(memory_format=torch.channels_last):
            y = y.contiguous(memory_format=torch.channels_last)
        N, Cc, Hh, Ww = y.shape
        need_bw = any(ctx.needs_input_grad[:3])                # no_grad / frozen layer: no mask is written (mask == NULL)
        mask = torch.empty((lib.uh_relu_mask_bytes(N * Hh * Ww, Cc),), dtype=torch.uint8, device=y.device) if need_bw else None
        _lib.check(lib.uh_bias_relu_forward(_ptr(y), _ptr(_f32(bias, 'bias')), _ptr(mask), N * Hh * Ww, Cc, _stream()),
                   'uh_bias_relu_forward')
        if need_bw:
            ctx.save_for_backward(x, weight, mask)
        ctx.padding, ctx.shape = padding, (N, Cc, Hh, Ww)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, weight, mask = ctx.saved_tensors
        N, Cc, Hh, Ww = ctx.shape
        gy = gy.contiguous(memory_format=torch.channels_last)
        g = torch.empty_like(gy, memory_format=torch.channels_last)
        db = torch.empty((Cc,), dtype=torch.float32, device=gy.device)
        npix = N * Hh * Ww
        nbytes = lib.uh_bias_relu_backward_workspace_bytes(npix, Cc)
        ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=gy.device)
        _lib.check(lib.uh_bias_relu_backward(_ptr(mask), _ptr(gy), _ptr(g), _ptr(db), _ptr(ws), nbytes, npix, Cc, _stream()),
                   'uh_bias_relu_backward')
        p = ctx.padding
        dx, dw, _ = torch.ops.aten.convolution_backward(
            g, x, weight, None, [1, 1], [p, p], [1, 1], False, [0, 0], 1,
            [ctx.needs_input_grad[0], ctx.needs_input_grad[1], False])
        return dx, dw, (db if ctx.needs_input_grad[2] else None), None


def conv_bias_relu(x, weight, bias, padding=1):
    """relu(conv2d(x, weight, bias, stride 1, padding)) -- homography_model.py:88-95 (_conv2d without batch norm)."""
    return _ConvBiasReLU.apply(x, weight, bias, padding)


class _ConvBiasReLUPool(torch.autograd.Function):
    """max_pool2d(relu(conv2d(x, w) + b), 2, 2): the epilogue produces the pooled map directly, and its backward routes the
    pooled gradient straight to the conv output (no full-resolution gradient of the pool in between).  The routing is kept as
    4 bits per pooled element: relu(conv + b) is never written back at full resolution (the conv output is released right
    after the forward) and the backward reads gpooled + bits (9 -> 5 B per conv-output element each way)."""

    @staticmethod
    def forward(ctx, x, weight, bias, padding):
        lib = _lib.load()
        y = torch.nn.functional.conv2d(x, weight, None, 1, padding)
        if not y.is_contiguous(memory_format=torch.channels_last):
            y = y.contiguous(memory_format=torch.channels_last)
        N, Cc, Hh, Ww = y.shape
        pooled = torch.empty((N, Cc, Hh // 2, Ww // 2), dtype=torch.float32, device=y.device,
                             memory_format=torch.channels_last)
        need_bw = any(ctx.needs_input_grad[:3])
        mask = torch.empty((lib.uh_pool_mask_bytes(N, Hh, Ww, Cc),), dtype=torch.uint8, device=y.device) if need_bw else None
        _lib.check(lib.uh_bias_relu_pool_forward(_ptr(y), _ptr(_f32(bias, 'bias')), _ptr(pooled), _ptr(mask), N, Hh, Ww, Cc,
                                                 _stream()), 'uh_bias_relu_pool_forward')
        if need_bw:
            ctx.save_for_backward(x, weight, mask)
        ctx.padding, ctx.shape = padding, (N, Cc, Hh, Ww)
        return pooled

    @staticmethod
    def backward(ctx, gp):
        lib = _lib.load()
        x, weight, mask = ctx.saved_tensors
        N, Cc, Hh, Ww = ctx.shape
        gp = gp.contiguous(memory_format=torch.channels_last)
        g = torch.empty((N, Cc, Hh, Ww), dtype=torch.float32, device=gp.device, memory_format=torch.channels_last)
        db = torch.empty((Cc,), dtype=torch.float32, device=gp.device)
        nbytes = lib.uh_bias_relu_pool_backward_workspace_bytes(N, Hh, Ww, Cc)
        ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=gp.device)
        _lib.check(lib.uh_bias_relu_pool_backward(_ptr(mask), _ptr(gp), _ptr(g), _ptr(db), _ptr(ws), nbytes, N, Hh, Ww, Cc,
                                                  _stream()), 'uh_bias_relu_pool_backward')
        p = ctx.padding
        dx, dw, _ = torch.ops.aten.convolution_backward(
            g, x, weight, None, [1, 1], [p, p], [1, 1], False, [0, 0], 1,
            [ctx.needs_input_grad[0], ctx.needs_input_grad[1], False])
        return dx, dw, (db if ctx.needs_input_grad[2] else None), None


def conv_bias_relu_pool(x, weight, bias, padding=1):
    """max_pool2d(relu(conv2d(x, weight, bias)), 2, 2) -- _conv2d + _maxpool2d (homography_model.py:88-105); H, W even."""
    return _ConvBiasReLUPool.apply(x, weight, bias, padding)
