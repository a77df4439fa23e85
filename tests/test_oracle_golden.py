"""CPU: the oracle against the reference-derived golden vectors (tests/golden/make_golden.py) and
against its own frozen outputs.  This is what "pins" the oracle."""
import numpy as np
import pytest

from oracle import hotpath_numpy as O


def test_warp_matches_reference_numpy_transformer(golden):
    """ref_numpy_transformer.npz = outputs of the reference's utils/numpy_spatial_transformer.py
    (_meshgrid + _interpolate, f64, gray).  Includes the reference's own self-test H (:157)."""
    g = golden('ref_numpy_transformer.npz')
    img = g['img']
    assert np.array_equal(O.meshgrid(48, 64, np.float64), g['grid'])
    for i in range(g['thetas'].shape[0]):
        out, _ = O.transformer(img[None, :, :, None], g['thetas'][i][None], (48, 64), np.float64)
        np.testing.assert_allclose(out[0, :, :, 0], g['outs'][i], rtol=0, atol=1e-10)
        out32, _ = O.transformer(img[None, :, :, None].astype(np.float32),
                                 g['thetas'][i][None].astype(np.float32), (48, 64), np.float32)
        # f32 faithful restatement: in-frame pixels agree with the reference to f32 rounding
        # (image range 0..255 -> 1e-4 relative to 255 ~ 3e-2 abs is far looser than observed)
        err = np.abs(out32[0, :, :, 0] - g['outs'][i])
        assert np.percentile(err, 99) < 5e-3, np.percentile(err, 99)


def test_rgb_warp_matches_reference_numpy_transformer_per_channel(golden):
    """ref_numpy_transformer_rgb.npz (round 5) = the reference's _meshgrid + _interpolate called once per channel of an RGB image
    (its own 3-channel branch raises under numpy >= 2; bilinear sampling is per channel, so this IS its C = 3 semantics): the
    oracle's C = 3 path -- the shape the product runs -- agrees with reference code, not only with the restatement."""
    g = golden('ref_numpy_transformer_rgb.npz')
    img = g['img']
    assert img.shape == (40, 56, 3)
    for i in range(g['thetas'].shape[0]):
        out, _ = O.transformer(img[None], g['thetas'][i][None], (40, 56), np.float64)
        np.testing.assert_allclose(out[0], g['outs'][i], rtol=0, atol=1e-10)
        out32, _ = O.transformer(img[None].astype(np.float32), g['thetas'][i][None].astype(np.float32), (40, 56), np.float32)
        err = np.abs(out32[0] - g['outs'][i])
        assert np.percentile(err, 99) < 5e-3, (i, np.percentile(err, 99))
    assert (g['outs'][2] == 0).mean() > 0.05            # the strong homography does leave the frame (out-of-range taps give 0 there)


def test_dlt_system_matches_reference_aux_matrices(golden):
    """ref_dlt_system.npz = A, b built by the reference's formula with its own Aux_M* constants."""
    d = golden('ref_dlt_system.npz')
    A, b = O.dlt_system(d['pts1'], d['h4p'], np.float32)
    assert np.array_equal(A, d['A'])
    assert np.array_equal(b, d['b'])


def test_dlt_reprojection_and_f32_vs_f64(golden):
    d = golden('ref_dlt_system.npz')
    H64 = O.solve_dlt_lapack64(d['pts1'], d['h4p'])
    p = d['pts1'].reshape(-1, 4, 2).astype(np.float64)
    q = p + (d['h4p'].reshape(-1, 4, 2) + d['pts1'].reshape(-1, 4, 2)).astype(np.float32) - d['pts1'].reshape(-1, 4, 2)
    ph = np.concatenate([p, np.ones(p.shape[:2] + (1,))], -1)
    r = np.einsum('bij,bkj->bki', H64, ph)
    r = r[..., :2] / r[..., 2:]
    p2 = (d['h4p'] + d['pts1']).astype(np.float32).reshape(-1, 4, 2)
    assert np.abs(r - p2).max() < 1e-8
    H32 = O.solve_dlt(d['pts1'], d['h4p'], np.float32)
    r32 = np.einsum('bij,bkj->bki', H32.astype(np.float64), ph)
    r32 = r32[..., :2] / r32[..., 2:]
    assert np.abs(r32 - p2).max() < 5e-2          # f32 LU of a cond~1e6 system: reprojection in px
    del q


def test_chain_frozen(golden):
    """The oracle reproduces its own committed outputs bit for bit (guards against silent edits)."""
    g = golden('chain_small.npz')
    f32 = O.photometric_chain(g['I'], g['I2'], g['pts1'], g['pred_h4p'], g['patch_indices'], 32, np.float32)
    assert np.array_equal(f32['H'], g['H32'])
    assert np.array_equal(f32['theta'], g['theta32'])
    assert np.array_equal(f32['warped'], g['warped32'])
    assert np.float32(f32['l1_loss']) == g['loss32']
    bw = O.photometric_chain_backward(g['I'], g['I2'], g['pts1'], g['pred_h4p'], g['patch_indices'], 32)
    np.testing.assert_allclose(bw['dh4p'], g['dh4p64'], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(bw['dtheta'], g['dtheta64'], rtol=1e-12, atol=1e-15)


def test_gt_homography_reproduces_I2(golden):
    """I' was generated as I(H_gt p): warping I with the GT deltas must give I2 on the patch."""
    d = O.synthetic_batch(3, 3, H=60, W=80, P=32, rho=8)
    f = O.photometric_chain(d['I'], d['I2'], d['pts1'], d['gt'], d['patch_indices'], 32, np.float64)
    assert f['l1_loss'] < 1e-6


def test_closed_form_gradient_vs_finite_difference():
    d = O.synthetic_batch(11, 2, H=40, W=56, P=16, rho=6)
    bw = O.photometric_chain_backward(d['I'], d['I2'], d['pts1'], d['pred_h4p'], d['patch_indices'], 16)

    def loss(h):
        A, b = O.dlt_system(d['pts1'].astype(np.float64), h, np.float64)
        hh = np.linalg.solve(A, b[..., None])[..., 0]
        Hm = np.ones((2, 9)); Hm[:, :8] = hh
        th = O.theta_from_H(Hm.reshape(-1, 3, 3), 56, 40, np.float64)
        w, _ = O.transformer(d['I'], th, (40, 56), np.float64)
        return O.l1_loss(O.gray_patch_gather(w, d['patch_indices'], 16, np.float64), d['I2'], np.float64)

    eps = 1e-5
    h0 = d['pred_h4p'].astype(np.float64)
    for k in range(2):
        for j in range(8):
            hp = h0.copy(); hm = h0.copy()
            hp[k, j] += eps; hm[k, j] -= eps
            fd = (loss(hp) - loss(hm)) / (2 * eps)
            assert abs(fd - bw['dh4p'][k, j]) < 1e-7 + 1e-5 * abs(fd)


def test_transformer_backward_with_dU_vs_torch_autograd():
    """Closed-form d/dtheta and d/dU against torch-CPU autograd over an f64 re-expression."""
    torch = pytest.importorskip('torch')
    rs = np.random.RandomState(5)
    B, H, W, C = 2, 12, 16, 3
    U = rs.randn(B, H, W, C)
    theta = np.tile(np.eye(3), (B, 1, 1)) + 0.05 * rs.randn(B, 3, 3)
    g = rs.randn(B, H, W, C)
    dth, dU = O.transformer_backward(U, theta, g, (H, W), np.float64, want_dU=True)

    Ut = torch.tensor(U, requires_grad=True); tt = torch.tensor(theta, requires_grad=True)
    grid = torch.tensor(O.meshgrid(H, W, np.float64))
    T = tt @ grid
    xn = T[:, 0] / T[:, 2]; yn = T[:, 1] / T[:, 2]
    x = (xn + 1) * W / 2; y = (yn + 1) * H / 2
    x0 = torch.floor(x).long(); y0 = torch.floor(y).long()
    x1 = (x0 + 1).clamp(0, W - 1); y1 = (y0 + 1).clamp(0, H - 1)
    x0 = x0.clamp(0, W - 1); y0 = y0.clamp(0, H - 1)
    bi = torch.arange(B)[:, None]
    Ia = Ut[bi, y0, x0]; Ib = Ut[bi, y1, x0]; Ic = Ut[bi, y0, x1]; Id = Ut[bi, y1, x1]
    wa = ((x1 - x) * (y1 - y))[..., None]; wb = ((x1 - x) * (y - y0))[..., None]
    wc = ((x - x0) * (y1 - y))[..., None]; wd = ((x - x0) * (y - y0))[..., None]
    out = (wa * Ia + wb * Ib + wc * Ic + wd * Id).reshape(B, H, W, C)
    (out * torch.tensor(g)).sum().backward()
    np.testing.assert_allclose(dth, tt.grad.numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(dU, Ut.grad.numpy(), rtol=1e-9, atol=1e-12)


def test_int32_overflow_semantics():
    """Far-field samples beyond int32 collapse to corner (0,0) twice (x86 cvttss2si -> INT_MIN), so
    the blend cancels; NaN-free, finite output."""
    U = np.random.RandomState(0).randn(1, 8, 8, 3).astype(np.float32)
    theta = np.array([[[1, 0, 0], [0, 1, 0], [1.0, 0, 1.0 - 2e-7]]], np.float32)   # t ~ 0 on the left edge
    out, cond = O.transformer(U, theta, (8, 8), np.float32)
    assert np.isfinite(out).all()
    x = O._float_to_int32_x86(np.array([3e9, -3e9, np.nan, 5.0], np.float32))
    assert list(x) == [-2147483648, -2147483648, -2147483648, 5]


def test_patch_losses_oracle_vs_torch_ops():
    """The NumPy restatement of the monitor losses (homography_model.py:136-166) against an independent evaluation
    with torch ops (avg_pool2d == slim.avg_pool2d VALID, stride 1)."""
    import torch
    import torch.nn.functional as F
    rs = np.random.RandomState(11)
    x = rs.randn(3, 24, 24, 1); y = x + 0.7 * rs.randn(3, 24, 24, 1)
    ref = O.patch_losses(x, y, rs.randn(3, 8), rs.randn(3, 8))
    xt = torch.from_numpy(x).permute(0, 3, 1, 2); yt = torch.from_numpy(y).permute(0, 3, 1, 2)
    pool = lambda v: F.avg_pool2d(v, 3, 1)
    mx, my = pool(xt), pool(yt)
    sx = pool(xt ** 2) - mx ** 2; sy = pool(yt ** 2) - my ** 2; sxy = pool(xt * yt) - mx * my
    ssim = ((2 * mx * my + 1e-4) * (2 * sxy + 9e-4)) / ((mx ** 2 + my ** 2 + 1e-4) * (sx + sy + 9e-4))
    assert abs(float(torch.clamp((1 - ssim) / 2, 0, 1).mean()) - ref['ssim_loss']) < 1e-12
    ad = (xt - yt).abs()
    assert abs(float(torch.where(ad < 1, 0.5 * ad * ad, ad - 0.5).mean()) - ref['l1_smooth_loss']) < 1e-12
    assert abs(float(torch.sqrt(((xt - yt) ** 2).mean())) - ref['rec_loss']) < 1e-12
    lx = torch.sqrt((xt ** 2).sum()); ly = torch.sqrt((yt ** 2).sum())
    assert abs(float(torch.sqrt(((yt / ly - xt / lx) ** 2).sum())) - ref['ncc_loss']) < 1e-12


def test_division_by_three_without_a_divider_is_correctly_rounded():
    """csrc/uh_warp.hip div_by_channels<3>: the sparse backward forms dPred/3 as q' = fma(fma(-3, q, a), y, q) with
    y = RN(1/3), q = RN(a y) (Markstein's correction step) instead of the v_div_* sequence, and must equal the dense
    chain's true division bit for bit.  Checked here for EVERY f32 mantissa (the sequence is invariant under scaling by
    powers of two away from under/overflow), both signs, with the fused multiply-adds evaluated exactly in f64."""
    m = np.arange(1 << 23, dtype=np.uint32) | np.uint32(0x3f800000)
    for sign in (np.uint32(0), np.uint32(0x80000000)):
        a = (m | sign).view(np.float32)
        y = np.float32(1.0) / np.float32(3.0)
        q = (a * y).astype(np.float32)
        r = (a.astype(np.float64) - 3.0 * q.astype(np.float64)).astype(np.float32)          # fma(-3, q, a)
        qq = (q.astype(np.float64) + r.astype(np.float64) * np.float64(y)).astype(np.float32)   # fma(r, y, q)
        assert np.array_equal(qq, (a / np.float32(3.0)).astype(np.float32))
