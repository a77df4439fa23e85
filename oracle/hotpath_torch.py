"""CPU ORACLE (test infrastructure, NOT product code) -- torch-CPU restatement of the reference's TF
graph for the hot path, OP FOR OP: every intermediate the TF graph materialises is materialised here
(tiled constants, [B,3,N] grids, four row-gathers, add_n ...), and the backward is torch autograd, the
counterpart of TF autodiff.  Two uses only:

  * bench.py's `cpu_baseline` leg: "reference-equivalent op graph on torch-CPU" timed on the GPU
    box's host cores (TensorFlow 1.x itself cannot be installed here);
  * tests: an autograd cross-check of the closed-form gradients in oracle/hotpath_numpy.py.

file:line below are relative to /root/reference/code/.  "parity unpinned" for tf.matrix_solve /
tf.linspace rounding: see oracle/hotpath_numpy.py header.
"""
import numpy as np
import torch

# ---- Aux selector matrices (utils/utils.py:11-122), rebuilt from their definition ------------------
def _aux():
    f = torch.float32
    z = torch.zeros(8, 8, dtype=f)
    M1 = z.clone(); M2 = z.clone(); M4 = z.clone(); M5 = z.clone()
    M71 = z.clone(); M72 = z.clone(); M8 = z.clone(); Mb = z.clone()
    M3 = torch.zeros(8, 1, dtype=f); M6 = torch.zeros(8, 1, dtype=f)
    for i in range(4):
        e, o = 2 * i, 2 * i + 1
        M1[o, e] = 1; M2[o, o] = 1; M3[o, 0] = 1          # odd rows: x, y, 1
        M4[e, e] = -1; M5[e, o] = -1; M6[e, 0] = -1       # even rows: -x, -y, -1
        M71[e, o] = 1; M71[o, e] = 1                      # even rows pick y', odd rows pick x'
        M72[e, e] = 1; M72[o, e] = -1                     # x, -x
        M8[e, o] = 1; M8[o, o] = -1                       # y, -y
        Mb[e, o] = -1; Mb[o, e] = 1                       # -y', x'
    return dict(M1=M1, M2=M2, M3=M3, M4=M4, M5=M5, M6=M6, M71=M71, M72=M72, M8=M8, Mb=Mb)


AUX = _aux()


def solve_DLT(pts_1, pred_h4p):
    """homography_model.py:169-250."""
    B = pts_1.shape[0]
    pts_1_tile = pts_1.unsqueeze(2)
    pred_pts_2_tile = pred_h4p.unsqueeze(2) + pts_1_tile
    T = {k: v.to(pts_1.dtype).unsqueeze(0).repeat(B, 1, 1) for k, v in AUX.items()}     # tf.tile
    A1 = T['M1'] @ pts_1_tile
    A2 = T['M2'] @ pts_1_tile
    A3 = T['M3']
    A4 = T['M4'] @ pts_1_tile
    A5 = T['M5'] @ pts_1_tile
    A6 = T['M6']
    A7 = (T['M71'] @ pred_pts_2_tile) * (T['M72'] @ pts_1_tile)
    A8 = (T['M71'] @ pred_pts_2_tile) * (T['M8'] @ pts_1_tile)
    A_mat = torch.stack([a.reshape(-1, 8) for a in (A1, A2, A3, A4, A5, A6, A7, A8)], dim=1).transpose(1, 2)
    b_mat = T['Mb'] @ pred_pts_2_tile
    H_8el = torch.linalg.solve(A_mat, b_mat)                      # tf.matrix_solve (LU, partial pivoting)
    H_9el = torch.cat([H_8el, torch.ones(B, 1, 1, dtype=pts_1.dtype)], 1)
    return H_9el.reshape(-1, 3, 3)


def _linspace_tf(start, stop, num, dtype):
    step = torch.tensor((stop - start) / (num - 1), dtype=dtype)
    return torch.tensor(start, dtype=dtype) + step * torch.arange(num, dtype=dtype)


def transformer(U, theta, out_size):
    """utils/tf_spatial_transformer.py:18-251 (_meshgrid :141, _transform :182, _interpolate :76)."""
    dt = U.dtype
    B, H, W, C = U.shape
    oh, ow = out_size
    # _meshgrid
    x_t = torch.ones(oh, 1, dtype=dt) @ _linspace_tf(-1.0, 1.0, ow, dt).unsqueeze(0)
    y_t = _linspace_tf(-1.0, 1.0, oh, dt).unsqueeze(1) @ torch.ones(1, ow, dtype=dt)
    grid = torch.cat([x_t.reshape(1, -1), y_t.reshape(1, -1), torch.ones(1, oh * ow, dtype=dt)], 0)
    grid = grid.reshape(-1).repeat(B).reshape(B, 3, -1)                         # tf.tile
    T_g = theta.reshape(-1, 3, 3) @ grid
    x_s, y_s, t_s = T_g[:, 0:1, :], T_g[:, 1:2, :], T_g[:, 2:3, :]
    t_s_flat = t_s.reshape(-1)
    small = 1e-7
    smallers = 1e-6 * (1.0 - (t_s_flat.abs() >= small).to(dt))
    t_s_flat = t_s_flat + smallers
    condition = (t_s_flat.abs() > small).to(dt).sum()
    x = x_s.reshape(-1) / t_s_flat
    y = y_s.reshape(-1) / t_s_flat
    # _interpolate
    x = (x + 1.0) * float(W) / 2.0
    y = (y + 1.0) * float(H) / 2.0
    big = 2147483648.0
    def cast_i32(v):                                   # x86 cvttss2si: out of range / NaN -> INT_MIN
        f = torch.floor(v.detach())
        ok = (f >= -big) & (f < big)
        return torch.where(ok, f, torch.full_like(f, -big)).to(torch.int64)
    x0 = cast_i32(x); x1 = x0 + 1
    y0 = cast_i32(y); y1 = y0 + 1
    x0 = x0.clamp(0, W - 1); x1 = x1.clamp(0, W - 1)
    y0 = y0.clamp(0, H - 1); y1 = y1.clamp(0, H - 1)
    base = (torch.arange(B) * (W * H)).unsqueeze(1).repeat(1, oh * ow).reshape(-1)      # _repeat
    base_y0 = base + y0 * W
    base_y1 = base + y1 * W
    idx_a, idx_b, idx_c, idx_d = base_y0 + x0, base_y1 + x0, base_y0 + x1, base_y1 + x1
    im_flat = U.reshape(-1, C)
    Ia = im_flat.index_select(0, idx_a); Ib = im_flat.index_select(0, idx_b)
    Ic = im_flat.index_select(0, idx_c); Id = im_flat.index_select(0, idx_d)
    x0_f, x1_f, y0_f, y1_f = x0.to(dt), x1.to(dt), y0.to(dt), y1.to(dt)
    wa = ((x1_f - x) * (y1_f - y)).unsqueeze(1)
    wb = ((x1_f - x) * (y - y0_f)).unsqueeze(1)
    wc = ((x - x0_f) * (y1_f - y)).unsqueeze(1)
    wd = ((x - x0_f) * (y - y0_f)).unsqueeze(1)
    output = wa * Ia + wb * Ib + wc * Ic + wd * Id                                        # tf.add_n
    return output.reshape(B, oh, ow, C), condition


def transform(I, H_mat, patch_indices, img_w, img_h, patch_size):
    """homography_model.py:252-269."""
    B = I.shape[0]
    M = np.array([[img_w / 2.0, 0., img_w / 2.0], [0., img_h / 2.0, img_h / 2.0], [0., 0., 1.]]).astype(np.float32)
    M_t = torch.from_numpy(M).to(I.dtype).unsqueeze(0).repeat(B, 1, 1)
    M_inv_t = torch.from_numpy(np.linalg.inv(M)).to(I.dtype).unsqueeze(0).repeat(B, 1, 1)
    theta = (M_inv_t @ H_mat) @ M_t
    warped, _ = transformer(I, theta, (img_h, img_w))
    gray = warped.mean(3).reshape(-1)
    batch_indices = (torch.arange(B) * (img_w * img_h)).unsqueeze(1).repeat(1, patch_size * patch_size).reshape(-1)
    pixel_indices = patch_indices.reshape(-1).to(torch.int64) + batch_indices
    pred_I2 = gray.index_select(0, pixel_indices).reshape(B, patch_size, patch_size, 1)
    return pred_I2, warped, theta


def photometric_l1(I, I2, pts1, pred_h4p, patch_indices, img_w, img_h, patch_size):
    """solve_DLT -> transform -> l1 branch of build_losses (:328)."""
    H_mat = solve_DLT(pts1, pred_h4p)
    pred_I2, warped, theta = transform(I, H_mat, patch_indices, img_w, img_h, patch_size)
    return (pred_I2 - I2).abs().mean(), pred_I2, warped, theta, H_mat


# ---- the photometric losses, op for op (homography_model.py:136-166, 298-352) ----------------------------------
def _L1_smooth_loss(x, y):
    abs_diff = torch.abs(x - y)
    return torch.mean(torch.where(abs_diff < 1, 0.5 * abs_diff * abs_diff, abs_diff - 0.5))


def _SSIM_loss(x, y, size=3):
    C1 = 0.01 ** 2
    C2 = 0.03 ** 2
    x = x.permute(0, 3, 1, 2); y = y.permute(0, 3, 1, 2)                       # slim.avg_pool2d is NHWC; torch is NCHW
    pool = lambda v: torch.nn.functional.avg_pool2d(v, size, 1)                # 'VALID'
    mu_x = pool(x); mu_y = pool(y)
    sigma_x = pool(x ** 2) - mu_x ** 2
    sigma_y = pool(y ** 2) - mu_y ** 2
    sigma_xy = pool(x * y) - mu_x * mu_y
    SSIM_n = (2 * mu_x * mu_y + C1) * (2 * sigma_xy + C2)
    SSIM_d = (mu_x ** 2 + mu_y ** 2 + C1) * (sigma_x + sigma_y + C2)
    SSIM = SSIM_n / SSIM_d
    return torch.clamp((1 - SSIM) / 2, 0, 1)


def _NCC_loss(x, y):
    len_x = torch.sqrt(torch.sum(x * x))
    len_y = torch.sqrt(torch.sum(y * y))
    return torch.sqrt(torch.sum((x / len_x - y / len_y) ** 2))


def patch_loss(loss_type, pred_I2, I2):
    """The loss tensor build_losses() trains on for `loss_type` (everything but h_loss)."""
    if loss_type == 'rec_loss':
        return torch.sqrt(torch.mean((pred_I2 - I2) ** 2))                     # :303
    if loss_type == 'ssim_loss':
        return torch.mean(_SSIM_loss(pred_I2, I2))                             # :314
    if loss_type == 'l1_loss':
        return torch.mean(torch.abs(pred_I2 - I2))                             # :328
    if loss_type == 'l1_smooth_loss':
        return _L1_smooth_loss(pred_I2, I2)                                    # :339
    if loss_type == 'ncc_loss':
        return _NCC_loss(I2, pred_I2)                                          # :350
    raise ValueError(loss_type)


def photometric_loss(loss_type, I, I2, pts1, pred_h4p, patch_indices, img_w, img_h, patch_size):
    """solve_DLT -> transform -> the `loss_type` branch of build_losses."""
    H_mat = solve_DLT(pts1, pred_h4p)
    pred_I2, warped, theta = transform(I, H_mat, patch_indices, img_w, img_h, patch_size)
    return patch_loss(loss_type, pred_I2, I2), pred_I2


def train_step_cpu(net, opt, batch, img_w, img_h, patch_size, loss_type='l1_loss', h4p_offset=None):
    """One full reference-equivalent train step on CPU tensors (VGG from the product package is plain
    torch and runs on CPU; the hot path is the op graph above).  h4p_offset: bench.py's hook -- per-pair corner
    offsets added to the regressor's output, the same tensor the GPU leg adds (TrainStep.h4p_offset)."""
    opt.zero_grad(set_to_none=True)
    pred_h4p = net(torch.cat([batch['I1_aug'], batch['I2_aug']], 3))
    if h4p_offset is not None:
        pred_h4p = pred_h4p + h4p_offset
    if loss_type == 'h_loss':
        loss = torch.sqrt(torch.mean((pred_h4p - batch['gt']) ** 2))
        with torch.no_grad():
            photometric_l1(batch['I_aug'], batch['I2_aug'], batch['pts1'], pred_h4p, batch['patch_indices'],
                           img_w, img_h, patch_size)
    else:
        loss = photometric_l1(batch['I_aug'], batch['I2_aug'], batch['pts1'], pred_h4p, batch['patch_indices'],
                              img_w, img_h, patch_size)[0]
    loss.backward()
    opt.step()
    return loss
