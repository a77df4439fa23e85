"""GPU parity, randomised: seeded sweeps over shapes / channel counts / output sizes / homographies that the hand-picked cases of
tests/test_gpu_parity.py do not enumerate -- 1-pixel frames and outputs, sizes straddling the 64 x 16 tile and the 256-pixel
wave, flips (negative scale), sign-changing t, samples far outside the frame.  Same bars as the fixed cases: forward
bit-equal to the f32 oracle (`assert_same_bits`), gradients against the f64 closed form, sparse == dense bit for bit, the
DLT within 4 ulp of the NumPy f32 LU in Eigen's operation order."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import hotpath_numpy as O          # noqa: E402  (checker only)
from test_gpu_parity import T, assert_same_bits, relerr  # noqa: E402


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('no HIP device')
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def ops(uh_lib_path):
    from unsuperviseddeephomographyral2018_amd import ops as _ops
    return _ops


def _random_theta(rs, B, kind):
    eye = np.tile(np.eye(3), (B, 1, 1))
    if kind == 'mild':
        th = eye + 0.08 * rs.randn(B, 3, 3)
    elif kind == 'strong':                                   # t changes sign inside the frame for many draws
        th = eye + 0.5 * rs.randn(B, 3, 3)
    elif kind == 'flip':                                     # negative scales / 90-degree-ish rotations
        th = eye * rs.choice([-1.0, 1.0], size=(B, 3, 1)) + 0.2 * rs.randn(B, 3, 3)
        th[:, 2, 2] = 1.0
    elif kind == 'far':                                      # large translations: most samples outside the frame
        th = eye + 0.05 * rs.randn(B, 3, 3)
        th[:, :2, 2] += rs.uniform(-6, 6, size=(B, 2))
    else:                                                    # 'tiny_t': the eps-guard and overflow territory
        th = eye + 0.3 * rs.randn(B, 3, 3)
        th[:, 2] *= rs.choice([1e-8, 1e-4, 1.0], size=(B, 1))
    return th.astype(np.float32)


SIZES = [1, 2, 3, 5, 15, 16, 17, 31, 33, 47, 63, 64, 65, 79, 97]


@pytest.mark.parametrize('seed', range(12))
def test_fuzz_warp_forward_is_bit_equal_to_the_oracle(ops, dev, seed):
    rs = np.random.RandomState(1000 + seed)
    for case in range(10):
        B = int(rs.randint(1, 5)); C = int(rs.randint(1, 5))
        H, W = int(rs.choice(SIZES)), int(rs.choice(SIZES))
        oh, ow = (H, W) if rs.rand() < 0.5 else (int(rs.choice(SIZES)), int(rs.choice(SIZES)))
        kind = ['mild', 'strong', 'flip', 'far', 'tiny_t'][(seed + case) % 5]
        U = rs.randn(B, H, W, C).astype(np.float32)
        theta = _random_theta(rs, B, kind)
        out, cond = ops.transformer(T(U, dev), T(theta, dev), (oh, ow))
        ref, c = O.transformer(U, theta, (oh, ow), np.float32)
        assert_same_bits(out.cpu().numpy(), ref, 'seed %d case %d: B%d %dx%dx%d -> %dx%d %s' % (seed, case, B, H, W, C, oh, ow, kind))
        assert float(cond) == float(c), (seed, case, kind)
        # the literal validation kernel agrees as well (it shares no fast path with the lean one)
        lit = ops.transformer_literal(T(U, dev), T(theta, dev), (oh, ow))
        assert_same_bits(lit.cpu().numpy(), ref, 'literal, seed %d case %d' % (seed, case))


@pytest.mark.parametrize('seed', range(6))
def test_fuzz_warp_backward_dtheta_and_dU(ops, dev, seed):
    """dtheta (and dU for every second case) against the f64 closed form evaluated at the f32 sample positions, on mild
    homographies (the violent laws have their own running-error-bounded tests); out_size differs from the frame half the time."""
    rs = np.random.RandomState(2000 + seed)
    for case in range(5):
        B = int(rs.randint(1, 4)); C = int(rs.randint(1, 5))
        H, W = int(rs.choice(SIZES[3:])), int(rs.choice(SIZES[3:]))
        oh, ow = (H, W) if rs.rand() < 0.5 else (int(rs.choice(SIZES[3:])), int(rs.choice(SIZES[3:])))
        lo = rs.rand(B, H // 4 + 2, W // 4 + 2, C).astype(np.float32)          # smooth-ish image: bilinear upsample of noise
        t = torch.from_numpy(lo).permute(0, 3, 1, 2)
        U = torch.nn.functional.interpolate(t, size=(H, W), mode='bilinear', align_corners=True).permute(0, 2, 3, 1).contiguous().numpy()
        theta = _random_theta(rs, B, 'mild')
        dOut = rs.randn(B, oh, ow, C).astype(np.float32)
        want_dU = case % 2 == 1
        Ut = T(U, dev).requires_grad_(want_dU); th = T(theta, dev).requires_grad_(True)
        out, _ = ops.transformer(Ut, th, (oh, ow), with_condition=False)
        out.backward(T(dOut, dev))
        ref = O.transformer_backward(U, theta, dOut, (oh, ow), np.float64, want_dU=want_dU, coord_dtype=np.float32)
        dth_ref = ref[0] if isinstance(ref, tuple) else ref
        what = 'seed %d case %d: B%d %dx%dx%d -> %dx%d' % (seed, case, B, H, W, C, oh, ow)
        assert relerr(th.grad.cpu().numpy().reshape(B, 9), np.asarray(dth_ref).reshape(B, 9)) <= 3e-4, what
        if want_dU:
            dU_ref = ref[1]
            assert np.abs(Ut.grad.cpu().numpy() - dU_ref).max() <= 1e-4 * max(np.abs(dU_ref).max(), 1e-30), what


@pytest.mark.parametrize('seed', range(4))
def test_fuzz_sparse_backward_equals_dense_chain(ops, dev, seed):
    """uh_warp_patch_backward on random patch rectangles (any size, anywhere inside the frame, C in 1..4) == the dense
    uh_gray_patch_backward -> uh_warp_backward chain, bit for bit."""
    rs = np.random.RandomState(3000 + seed)
    for case in range(5):
        B = int(rs.randint(1, 5)); C = int(rs.randint(1, 5))
        H, W = int(rs.randint(20, 90)), int(rs.randint(20, 120))
        P = int(rs.randint(3, min(H, W) - 1))
        U = rs.randn(B, H, W, C).astype(np.float32)
        theta = _random_theta(rs, B, ['mild', 'strong', 'far'][case % 3])
        x0 = rs.randint(0, W - P + 1, B); y0 = rs.randint(0, H - P + 1, B)
        u = np.arange(P)
        idx = ((u[None, :, None] + y0[:, None, None]) * W + (u[None, None, :] + x0[:, None, None])).reshape(B, P * P).astype(np.int32)
        dPred = rs.randn(B, P, P, 1).astype(np.float32)
        Ut, it = T(U, dev), T(idx, dev)
        th1 = T(theta, dev).requires_grad_(True)
        _, pred = ops.warp_gather(Ut, th1, it, P)
        pred.backward(T(dPred, dev))
        th2 = T(theta, dev).requires_grad_(True)
        warped, _ = ops.transformer(Ut, th2, (H, W), with_condition=False)
        ops.gray_patch_gather(warped, it, P).backward(T(dPred, dev))
        a, b = th1.grad, th2.grad
        assert (torch.isnan(a) == torch.isnan(b)).all() and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b)), \
            'seed %d case %d' % (seed, case)


def test_fuzz_dlt_forward_and_backward(ops, dev):
    """2 048 random quadrilaterals (patch corners anywhere in a 240 x 320 frame, deltas up to +-60 px with sub-pixel noise):
    f32 solve within 4 ulp of the NumPy f32 LU in Eigen's operation order, f64 solve == LAPACK to 2e-7, backward against the
    f64 closed form with the per-system conditioning bound of DESIGN section 4."""
    rs = np.random.RandomState(4000)
    B = 2048
    x0 = rs.randint(0, 190, B); y0 = rs.randint(0, 110, B); P = rs.randint(24, 129, B)
    pts1 = np.stack([x0, y0, x0 + P, y0, x0 + P, y0 + P, x0, y0 + P], 1).astype(np.float32)
    h4p = (rs.uniform(-1, 1, (B, 8)) * np.minimum(60, 0.45 * P)[:, None] + rs.randn(B, 8)).astype(np.float32)
    H = ops.solve_dlt(T(pts1, dev), T(h4p, dev)).cpu().numpy()
    H32 = O.solve_dlt(pts1, h4p, np.float32)
    np.testing.assert_allclose(H, H32, rtol=5e-7, atol=0)
    H64 = O.solve_dlt_lapack64(pts1, h4p)
    Hd = ops.solve_dlt(T(pts1, dev), T(h4p, dev), solve_f64=True).cpu().numpy()
    np.testing.assert_allclose(Hd, H64, rtol=2e-7, atol=1e-12)
    dH = rs.randn(B, 3, 3).astype(np.float32); dH[:, 2, 2] = 0
    # this law is wider than the dataloader's (small patches with deltas up to 0.45 P: near-collinear predicted corners): cond(A) in
    # pixel units runs from 1e5 to > 1e9.  The documented per-system bound (2e-3 in f32, 1e-5 with the f64 solve)
    # is asserted where cond(A) <= 1e7 -- the range SURVEY section 7 gives for the data law -- and scaled by cond / 1e7 beyond it.
    A64, _ = O.dlt_system(pts1.astype(np.float64), h4p.astype(np.float64), np.float64)
    cond = np.linalg.cond(A64)
    assert (cond <= 1e7).mean() > 0.8
    for f64, bound in ((False, 2e-3), (True, 1e-5)):
        hp = T(h4p, dev).requires_grad_(True)
        Hm = ops.solve_dlt(T(pts1, dev), hp, solve_f64=f64)
        Hm.backward(T(dH, dev))
        got = hp.grad.cpu().numpy()
        assert np.isfinite(got).all()
        ref = O.solve_dlt_backward(pts1, h4p, H64, dH.astype(np.float64), np.float64)
        err = np.abs(got - ref).max(axis=1) / np.maximum(np.abs(ref).max(axis=1), 1e-30)
        lim = bound * np.maximum(1.0, cond / 1e7)
        worst = int(np.argmax(err / lim))
        assert (err <= lim).all(), (f64, float(err[worst]), float(cond[worst]), float(np.median(err)))


@pytest.mark.parametrize('seed', range(4))
def test_fuzz_prepare_inputs(dev, uh_lib_path, seed):
    """uh_prepare_inputs on random frame sizes (H*W % 4 == 0 -> the 4-pixels-per-lane path, otherwise the scalar one), patch sizes
    and positions incl. patches touching the frame border, with joint / disjoint / identity augmentation rows mixed in one batch."""
    from unsuperviseddeephomographyral2018_amd import dataloader as dl
    rs = np.random.RandomState(5000 + seed)
    for case in range(4):
        B = int(rs.randint(1, 6)); H, W = int(rs.randint(8, 70)), int(rs.randint(8, 90))
        P = int(rs.randint(2, min(H, W) + 1))
        I = rs.randint(0, 256, (B, H, W, 3)).astype(np.uint8); Ip = rs.randint(0, 256, (B, H, W, 3)).astype(np.uint8)
        x0 = rs.randint(0, W - P + 1, B); y0 = rs.randint(0, H - P + 1, B)
        pts1 = np.stack([x0, y0, x0 + P, y0, x0 + P, y0 + P, x0, y0 + P], 1).astype(np.float32)
        aug = None
        if case % 2 == 1:
            aug = dl.sample_augmentation(B, 'train' if case == 1 else 'test', 0.7, torch.Generator().manual_seed(seed * 10 + case))
        got = dl.prepare_inputs(torch.from_numpy(I).to(dev), torch.from_numpy(Ip).to(dev), torch.from_numpy(pts1), P, aug)
        ref = O.prepare_inputs(I, Ip, pts1, P, None if aug is None else aug.numpy())
        tol = 1e-6 if aug is None else 2e-4                    # powf: a few ulp of values up to 510, after standardisation
        for k in ('I_aug', 'I_prime_aug', 'I1', 'I2', 'I1_aug', 'I2_aug'):
            assert np.abs(got[k].cpu().numpy() - ref[k]).max() <= tol, (seed, case, k, B, H, W, P)
        assert np.array_equal(got['patch_indices'].cpu().numpy(), ref['patch_indices']), (seed, case)


@pytest.mark.parametrize('seed', range(3))
def test_fuzz_patch_losses_and_fused_patch_path(ops, dev, seed):
    """Six loss values for random B and P (3 .. 70: not multiples of the 16 x 16 SSIM tile or the 1 024-pixel loss chunk) against
    the f64 restatement; and the fused patch kernel against the un-fused chain on random rectangles: pred_I2 bit-equal, the l1
    value and d theta within the documented bounds."""
    rs = np.random.RandomState(6000 + seed)
    for case in range(4):
        B = int(rs.randint(1, 7)); P = int(rs.randint(3, 71))
        x = (rs.randn(B, P, P, 1) * 1.5).astype(np.float32)
        y = (x + rs.randn(B, P, P, 1) * rs.choice([0.05, 0.8, 2.0], size=(B, 1, 1, 1))).astype(np.float32)
        h4p = rs.randn(B, 8).astype(np.float32) * 20; gt = rs.randint(-45, 46, (B, 8)).astype(np.float32)
        got = ops.patch_losses(T(x, dev), T(y, dev), T(h4p, dev), T(gt, dev)).cpu().numpy()
        ref = O.patch_losses(x, y, h4p, gt)
        for i, k in enumerate(('rec_loss', 'ssim_loss', 'l1_loss', 'l1_smooth_loss', 'ncc_loss', 'h_loss')):
            assert abs(got[i] - ref[k]) <= 1e-5 * max(1.0, abs(ref[k])), (seed, case, B, P, k, got[i], ref[k])
    for case in range(4):
        B = int(rs.randint(1, 5)); C = int(rs.randint(1, 5))
        H, W = int(rs.randint(24, 100)), int(rs.randint(24, 130))
        P = int(rs.randint(4, min(H, W) - 1))
        lo = rs.rand(B, H // 4 + 2, W // 4 + 2, C).astype(np.float32)
        U = torch.nn.functional.interpolate(torch.from_numpy(lo).permute(0, 3, 1, 2), size=(H, W), mode='bilinear',
                                            align_corners=True).permute(0, 2, 3, 1).contiguous().numpy()
        theta = _random_theta(rs, B, 'mild')
        x0 = rs.randint(0, W - P + 1, B); y0 = rs.randint(0, H - P + 1, B)
        u = np.arange(P)
        idx = ((u[None, :, None] + y0[:, None, None]) * W + (u[None, None, :] + x0[:, None, None])).reshape(B, P * P).astype(np.int32)
        I2 = rs.rand(B, P, P, 1).astype(np.float32)
        Ut, it, I2t = T(U, dev), T(idx, dev), T(I2, dev)
        th1 = T(theta, dev).requires_grad_(True)
        loss1, pred1 = ops.warp_patch_l1(Ut, th1, I2t, it, P)
        loss1.backward()
        th2 = T(theta, dev).requires_grad_(True)
        _, pred2 = ops.warp_gather(Ut, th2, it, P)
        loss2 = ops.patch_losses(pred2, I2t, train='l1_loss')[2]
        loss2.backward()
        what = 'seed %d case %d: B%d %dx%dx%d P%d' % (seed, case, B, H, W, C, P)
        assert torch.equal(pred1, pred2), what
        assert abs(float(loss1.detach()) - float(loss2.detach())) <= 1e-6 * max(1.0, abs(float(loss2.detach()))), what
        assert relerr(th1.grad.cpu().numpy(), th2.grad.cpu().numpy()) <= 2e-4, what
