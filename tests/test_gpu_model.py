"""GPU: the HomographyModel call surface, every loss type, the train step, test-mode statistics, and an
end-to-end gradient comparison against the reference-equivalent op graph on torch-CPU."""
import copy

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import hotpath_torch as OT      # noqa: E402  (checker only)


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('no HIP device')
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def pkg(uh_lib_path):
    from unsuperviseddeephomographyral2018_amd import homography_model as hm, synthetic, homography_CNN_synthetic as drv
    return hm, synthetic, drv


def params(hm, mode, B, P, W, H, loss_type):
    return hm.homography_model_params(mode=mode, batch_size=B, patch_size=P, img_w=W, img_h=H, loss_type=loss_type,
                                      use_batch_norm=False, augment_list=['normalize'],
                                      leftright_consistent_weight=0)


B, H, W, P, RHO = 4, 120, 160, 64, 20


@pytest.mark.parametrize('loss_type', ['h_loss', 'rec_loss', 'ssim_loss', 'l1_loss', 'l1_smooth_loss', 'ncc_loss'])
def test_call_surface_and_gradients(pkg, dev, loss_type):
    hm, synthetic, _ = pkg
    batch = synthetic.make_batch(B, H, W, P, RHO, seed=1, device=dev)
    torch.manual_seed(0)
    m = hm.HomographyModel(params(hm, 'train', B, P, W, H, loss_type), *synthetic.model_args(batch), solve_f64=True)
    for attr in ('pred_h4p', 'H_mat', 'pred_I2', 'h_loss', 'rec_loss', 'ssim_loss', 'l1_loss', 'l1_smooth_loss',
                 'ncc_loss', 'I', 'I_prime', 'I1', 'I2', 'I1_aug', 'I2_aug', 'pts_1', 'gt', 'model_input'):
        assert hasattr(m, attr), attr
    assert m.pred_h4p.shape == (B, 8) and m.H_mat.shape == (B, 3, 3) and m.pred_I2.shape == (B, P, P, 1)
    for n in ('h_loss', 'rec_loss', 'ssim_loss', 'l1_loss', 'l1_smooth_loss', 'ncc_loss'):
        v = getattr(m, n)
        assert torch.isfinite(v).all()
        # others are stop_gradient monitors: they hang off the same one-launch loss node, but no gradient reaches the
        # model through them (checked on the values below and in test_patch_loss_backward_vs_torch_autograd)
        if n == loss_type:
            assert v.requires_grad
    net = hm.get_variables()
    net.zero_grad(set_to_none=True)
    for n in ('h_loss', 'rec_loss', 'ssim_loss', 'l1_loss', 'l1_smooth_loss', 'ncc_loss'):
        if n != loss_type and getattr(m, n).requires_grad:              # a monitor: differentiating it yields exactly 0
            (gz,) = torch.autograd.grad(getattr(m, n), m.pred_h4p, retain_graph=True, allow_unused=True)
            assert gz is None or float(gz.abs().max()) == 0.0, n
    (g_h4p,) = torch.autograd.grad(m.loss, m.pred_h4p, retain_graph=True)
    m.loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
    assert float(net.fc2.weight.grad.abs().sum()) > 0
    # gradient VALUES: d loss / d pred_h4p through DLT -> warp -> gather -> loss (all HIP) against f64 autograd of the
    # reference-equivalent op graph on CPU (oracle/hotpath_torch.py) at the same pred_h4p.  cond(A) of the DLT system is
    # ~1e7, so an f32 solve moves theta by ~1e-3 relative; the tolerance is relative to the gradient's largest entry.
    h = m.pred_h4p.detach().cpu().double().requires_grad_(True)
    cpu = {k: v.detach().cpu() for k, v in batch.items()}
    if loss_type == 'h_loss':
        ref_loss = torch.sqrt(torch.mean((h - cpu['gt'].double()) ** 2))
    else:
        ref_loss, _ = OT.photometric_loss(loss_type, cpu['I_aug'].double(), cpu['I2_aug'].double(), cpu['pts1'].double(), h,
                                          cpu['patch_indices'], W, H, P)
    ref_loss.backward()
    assert abs(float(m.loss) - float(ref_loss)) <= 1e-4 * max(1.0, abs(float(ref_loss))), (float(m.loss), float(ref_loss))
    err = (g_h4p.cpu().double() - h.grad).abs().max() / h.grad.abs().max()
    assert float(err) < 5e-3, (loss_type, float(err))
    # ... and the fused patch path (no warped frame) gives the same loss and gradient
    torch.manual_seed(0)
    mf = hm.HomographyModel(params(hm, 'train', B, P, W, H, loss_type), *synthetic.model_args(batch),
                            reuse_variables=True, fused_patch=True, solve_f64=True)
    if loss_type != 'h_loss':
        hf = mf.pred_h4p.detach().cpu().double().requires_grad_(True)
        lf, _ = OT.photometric_loss(loss_type, cpu['I_aug'].double(), cpu['I2_aug'].double(), cpu['pts1'].double(), hf,
                                    cpu['patch_indices'], W, H, P)
        lf.backward()
        (gf,) = torch.autograd.grad(mf.loss, mf.pred_h4p)
        assert abs(float(mf.loss) - float(lf)) <= 1e-4 * max(1.0, abs(float(lf)))
        assert float((gf.cpu().double() - hf.grad).abs().max() / hf.grad.abs().max()) < 5e-3, loss_type
    # second tower shares the variables
    m2 = hm.HomographyModel(params(hm, 'train', B, P, W, H, loss_type), *synthetic.model_args(batch),
                            reuse_variables=True, model_index=1)
    assert m2.net is net


def test_monitor_values_match_numpy(pkg, dev):
    hm, synthetic, _ = pkg
    batch = synthetic.make_batch(B, H, W, P, RHO, seed=2, device=dev)
    m = hm.HomographyModel(params(hm, 'test', B, P, W, H, 'l1_loss'), *synthetic.model_args(batch))
    pred = m.pred_I2.detach().cpu().numpy().astype(np.float64); I2 = batch['I2_aug'].cpu().numpy().astype(np.float64)
    ph = m.pred_h4p.detach().cpu().numpy().astype(np.float64); gt = batch['gt'].cpu().numpy().astype(np.float64)
    assert abs(float(m.l1_loss) - np.abs(pred - I2).mean()) < 1e-6
    assert abs(float(m.rec_loss) - np.sqrt(((pred - I2) ** 2).mean())) < 1e-5
    assert abs(float(m.h_loss) - np.sqrt(((ph - gt) ** 2).mean())) < 1e-4
    ad = np.abs(pred - I2)
    assert abs(float(m.l1_smooth_loss) - np.where(ad < 1, 0.5 * ad ** 2, ad - 0.5).mean()) < 1e-6
    bh = np.sqrt(((ph - gt) ** 2).mean(1)); hid = np.sqrt((gt ** 2).mean(1)); fail = bh >= hid
    assert float(m.num_fail) == fail.sum()
    assert abs(float(m.bounded_h_loss) - np.where(fail, hid, bh).mean()) < 1e-4


def test_fused_patch_equals_full_frame_path(pkg, dev):
    hm, synthetic, _ = pkg
    batch = synthetic.make_batch(B, H, W, P, RHO, seed=3, device=dev)
    torch.manual_seed(0)
    net = hm.VGGRegressor(P).to(dev).to(memory_format=torch.channels_last).eval()
    grads = []
    for fused in (False, True):
        net.zero_grad(set_to_none=True)
        m = hm.HomographyModel(params(hm, 'test', B, P, W, H, 'l1_loss'), *synthetic.model_args(batch), net=net,
                               fused_patch=fused)
        m.l1_loss.backward()
        grads.append((float(m.l1_loss.detach()), m.pred_I2.detach().clone(), net.fc2.weight.grad.clone()))
    assert abs(grads[0][0] - grads[1][0]) < 1e-7
    assert torch.equal(grads[0][1], grads[1][1])
    rel = float((grads[0][2] - grads[1][2]).abs().max() / grads[0][2].abs().max())
    assert rel < 1e-4, rel


def test_end_to_end_gradient_vs_cpu_op_graph(pkg, dev):
    """Same weights, same inputs, dropout off: d l1_loss / d variables from the HIP path vs the TF-graph-shaped
    torch-CPU restatement with autograd.  Tolerance 2e-3 relative to each tensor's max (f32, different
    LU/summation orders, ~1e6-conditioned DLT)."""
    hm, synthetic, _ = pkg
    batch = synthetic.make_batch(B, H, W, P, RHO, seed=4, device=dev)
    torch.manual_seed(1)
    net = hm.VGGRegressor(P).eval()
    net_gpu = copy.deepcopy(net).to(dev).to(memory_format=torch.channels_last).eval()
    # This test pins the HOT PATH's gradient chain, so the regressor runs on the stock torch conv+bias+ReLU+pool ops
    # (bitwise reproducible forward).  With the fused epilogue MIOpen is called without a bias and may pick a split-K
    # (atomic) forward solver: activations then vary by ~1e-7 from run to run, which on these smooth inputs flips a few
    # near-tied max-pool winners and moves the early layers' gradients by ~1e-3 -- legitimate f32 behaviour, but not what
    # a 2e-3 comparison against a CPU graph can pin.  The epilogue has its own tests (op level and whole regressor).
    net_gpu.fused_epilogue = False
    m = hm.HomographyModel(params(hm, 'test', B, P, W, H, 'l1_loss'), *synthetic.model_args(batch), net=net_gpu)
    m.l1_loss.backward()
    cb = {k: v.cpu() for k, v in batch.items()}
    pred = net(torch.cat([cb['I1_aug'], cb['I2_aug']], 3))
    loss, pred_I2, warped, theta, Hm = OT.photometric_l1(cb['I_aug'], cb['I2_aug'], cb['pts1'], pred,
                                                         cb['patch_indices'], W, H, P)
    loss.backward()
    assert abs(float(loss) - float(m.l1_loss.detach())) < 1e-4                      # north_star: L1 within 1e-4
    assert float((pred_I2 - m.pred_I2.detach().cpu()).abs().max()) < 2e-3
    for (n, pc), pg in zip(net.named_parameters(), net_gpu.parameters()):
        rel = float((pc.grad - pg.grad.cpu()).abs().max() / pc.grad.abs().max().clamp_min(1e-30))
        # Layers after the last max-pool (convs.6, convs.7, fc1, fc2) see the hot path's gradient through smooth ops only:
        # tight.  Below a max-pool the comparison is at the mercy of near-tied pooling winners: a 1e-7 difference between the
        # CPU and MIOpen conv outputs (MIOpen may pick split-K atomic solvers depending on what ran earlier in the process)
        # flips a few winners on these smooth inputs and moves those gradients by ~1e-3..1e-2.
        tight = n.split('.')[0] in ('fc1', 'fc2') or n.startswith(('convs.6', 'convs.7'))
        assert rel < (2e-3 if tight else 5e-2), (n, rel)


def test_train_step_overfits_fixed_batch(pkg, dev):
    hm, synthetic, drv = pkg
    args = drv.build_parser().parse_args(['--batch_size', str(B), '--img_h', str(H), '--img_w', str(W),
                                          '--patch_size', str(P), '--rho', str(RHO), '--lr', '1e-4'])
    torch.manual_seed(0)
    step = drv.TrainStep(args, dev, world=1)
    batch = synthetic.make_batch(B, H, W, P, RHO, seed=5, device=dev)
    losses = [float(step(batch).l1_loss.detach()) for _ in range(60)]
    assert np.isfinite(losses).all()
    assert np.mean(losses[-10:]) < 0.9 * np.mean(losses[:10]), (losses[:3], losses[-3:])
    assert step.global_step == 60
    # the Adam update is TF1's: the fused optimizer was handed the per-step eps of the 60th update (tests/test_optimizer.py)
    assert step.adam_t == 60 and step.opt.param_groups[0]['eps'] == drv.tf_adam_eps(60)
    assert all(float(st['step']) == 60.0 for st in step.opt.state.values())
    sd = step.state_dict()
    step2 = drv.TrainStep(args, dev, world=1)
    step2.load_state_dict(sd)
    assert step2.global_step == 60 and step2.adam_t == 60
    del sd['adam_t']                                            # a checkpoint of rounds 1-5: the counter comes from the optimizer state
    step3 = drv.TrainStep(args, dev, world=1)
    step3.load_state_dict(sd)
    assert step3.adam_t == 60


def test_supervised_mode_and_test_statistics(pkg, dev):
    hm, synthetic, drv = pkg
    args = drv.build_parser().parse_args(['--batch_size', '8', '--img_h', str(H), '--img_w', str(W),
                                          '--patch_size', str(P), '--rho', str(RHO), '--loss_type', 'h_loss',
                                          '--num_test_data', '16'])
    torch.manual_seed(0)
    step = drv.TrainStep(args, dev, world=1)
    batch = synthetic.make_batch(8, H, W, P, RHO, seed=6, device=dev)
    first = float(step(batch).h_loss.detach())
    for _ in range(80):
        m = step(batch)
    assert float(m.h_loss.detach()) < first
    res = drv.TestHomography(args, step_fn=step).run()
    assert res['num_pairs'] == 3 * 2 * 8 and 0 <= res['fail_percent'] <= 100 and res['mean_corner_error'] > 0
    # --save_visual (the reference's test-mode default, :539-552): one correspondence image per test step into --results_dir
    import os
    import tempfile
    from PIL import Image
    with tempfile.TemporaryDirectory() as td:
        args.results_dir = os.path.join(td, 'report')
        res2 = drv.TestHomography(args, step_fn=step).run(save_visual=True)
        files = sorted(os.listdir(args.results_dir))
        assert len(files) == 3 * 2 and all(f.endswith('.jpg') and '_h_loss_loss_' in f for f in files), files
        assert files[0].startswith('0_') and any(f.startswith('8_') for f in files)        # named by step * batch_size
        assert Image.open(os.path.join(args.results_dir, files[0])).size == (2 * W, H)
        assert abs(res2['mean_corner_error'] - res['mean_corner_error']) <= 1e-4 * res['mean_corner_error']      # same statistics (MIOpen's solvers are not bit-repeatable)


@pytest.mark.parametrize('fused', [False, True])
def test_graph_tail_equals_unfused_chain(pkg, dev, fused):
    """--graph_tail: DLT -> warp -> gather -> L1 and their backward as ONE library call replaying a captured hipGraph
    (csrc/uh_tail.hip).  Same kernels, same order => identical loss / pred_I2 / H and identical parameter gradients; on
    a non-default stream the chain is captured once and replayed afterwards."""
    hm, synthetic, _ = pkg
    from unsuperviseddeephomographyral2018_amd import ops
    batch = synthetic.make_batch(B, H, W, P, RHO, seed=11, device=dev)
    ops.TailPlan._cache.clear()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    # the model-level switch: one full step through HomographyModel(graph_tail=True) on a private stream
    torch.manual_seed(3)
    net = hm.VGGRegressor(P).to(dev).to(memory_format=torch.channels_last)
    with torch.cuda.stream(side):
        m = hm.HomographyModel(params(hm, 'train', B, P, W, H, 'l1_loss'), *synthetic.model_args(batch), net=net,
                               fused_patch=fused, graph_tail=True)
        m.l1_loss.backward()
    side.synchronize()
    assert np.isfinite(float(m.l1_loss.detach())) and all(torch.isfinite(p.grad).all() for p in net.parameters())
    assert m.pred_I2.shape == (B, P, P, 1) and m.H_mat.shape == (B, 3, 3)
    # the tail against the separate ops on IDENTICAL h4p (dropout makes two model constructions differ)
    h4p = torch.randn(B, 8, device=dev) * 3
    with torch.cuda.stream(side):
        a = h4p.clone().requires_grad_(True)
        Hm, th = ops.solve_dlt(batch['pts1'], a, W, H)
        if fused:
            la, pa = ops.warp_patch_l1(batch['I_aug'], th, batch['I2_aug'], batch['patch_indices'], P)
        else:
            wa, _ = ops.transformer(batch['I_aug'], th, (H, W), with_condition=False)
            pa = ops.gray_patch_gather(wa, batch['patch_indices'], P)
            la = ops.l1_loss(pa, batch['I2_aug'])
        la.backward()
        bgrad = []
        for _ in range(3):
            b = h4p.clone().requires_grad_(True)
            lb, pb, Hb, plan = ops.photometric_tail(batch['pts1'], b, batch['I_aug'], batch['I2_aug'],
                                                    batch['patch_indices'], P, fused_patch=fused, graph=True)
            (2.0 * lb).backward()
            bgrad.append(b.grad.clone())
    side.synchronize()
    assert float(la) == float(lb)
    assert torch.equal(pa, pb) and torch.equal(Hm, Hb)
    assert torch.equal(2.0 * a.grad, bgrad[0]) and torch.equal(bgrad[0], bgrad[2])
    if not fused:
        assert torch.equal(plan.warped, wa)
    st = plan.stats()
    assert st['captures'] >= 1 and st['launches'] >= 3 and st['captures'] <= st['launches'] - 2, st


def test_tail_plan_reentry_constants_and_address_churn(pkg, dev):
    """Three hazards of the cached tail plan (VERDICT r1 / ADVICE r1):
    (1) two forwards through the SAME plan before the first backward (second tower, gradient accumulation) must not
        corrupt the first one's loss / pred / H / gradient -- outputs are copies, not views of plan buffers;
    (2) the captured graph bakes M / Minv in as kernel arguments: re-running a plan with different M_host contents must
        re-capture, not replay stale constants;
    (3) a caller that hands in freshly allocated tensors every step must not pay a capture per step;
    (4) ... and an argument set that repeats after that is captured again."""
    hm, synthetic, _ = pkg
    from unsuperviseddeephomographyral2018_amd import ops
    batch = synthetic.make_batch(B, H, W, P, RHO, seed=21, device=dev)
    ops.TailPlan._cache.clear()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    args = (batch['I_aug'], batch['I2_aug'], batch['patch_indices'], P)
    g = torch.Generator(device='cpu').manual_seed(5)
    h1 = (torch.randn(B, 8, generator=g) * 3).to(dev); h2 = (torch.randn(B, 8, generator=g) * 3).to(dev)
    with torch.cuda.stream(side):
        # reference: each h4p alone
        ref = []
        for h in (h1, h2):
            x = h.clone().requires_grad_(True)
            l, pr, Hm, plan = ops.photometric_tail(batch['pts1'], x, *args, graph=True)
            l.backward()
            ref.append((float(l), pr.clone(), Hm.clone(), x.grad.clone()))
        # (1) both forwards first, then both backwards
        a = h1.clone().requires_grad_(True); b = h2.clone().requires_grad_(True)
        la, pa, Ha, plan_a = ops.photometric_tail(batch['pts1'], a, *args, graph=True)
        lb, pb, Hb, plan_b = ops.photometric_tail(batch['pts1'], b, *args, graph=True)
        assert plan_a is plan_b
        la.backward(); lb.backward()
        side.synchronize()
        assert float(la) == ref[0][0] and float(lb) == ref[1][0]
        assert torch.equal(pa, ref[0][1]) and torch.equal(Ha, ref[0][2]) and torch.equal(a.grad, ref[0][3])
        assert torch.equal(pb, ref[1][1]) and torch.equal(Hb, ref[1][2]) and torch.equal(b.grad, ref[1][3])
        # (2) same plan, same pointers, different M / Minv (those of a 2x wider frame): theta changes, so pred must
        M2, Minv2 = ops.m_and_minv(2 * W, H)
        keep = (plan.M, plan.Minv, plan._Mh, plan._Mih)
        plan._Mh = np.ascontiguousarray(M2.reshape(9), np.float32)
        plan._Mih = np.ascontiguousarray(Minv2.reshape(9).astype(np.float32))
        x = h1.clone().requires_grad_(True)
        l2, p2, _, _ = ops.photometric_tail(batch['pts1'], x, *args, graph=True)
        side.synchronize()
        plan.M, plan.Minv, plan._Mh, plan._Mih = keep
        assert not torch.equal(p2, ref[0][1])
        x = h1.clone().requires_grad_(True)
        l3, p3, _, _ = ops.photometric_tail(batch['pts1'], x, *args, graph=True)
        side.synchronize()
        assert torch.equal(p3, ref[0][1]) and float(l3) == ref[0][0]
        # (3) fresh addresses every call: the capture count stops growing after the thrash threshold
        c0 = plan.stats()['captures']
        held = []
        for i in range(24):
            Ui = batch['I_aug'].clone(); held.append(Ui)             # kept alive => a new address every time
            x = h1.clone().requires_grad_(True)
            li, pi, _, _ = ops.photometric_tail(batch['pts1'], x, Ui, batch['I2_aug'], batch['patch_indices'], P, graph=True)
            assert torch.equal(pi, ref[0][1])
        side.synchronize()
        assert plan.stats()['captures'] - c0 <= 9, plan.stats()
        # (4) ... but the detector does not switch the graph route off for good: an argument set that COMES BACK (a pool
        # of batches cycled by the trainer) is captured on its second sighting and replayed from then on
        c1 = plan.stats()['captures']
        for rep in range(4):
            for Ui in held[-3:]:                                   # (run eagerly above: their first sighting)
                x = h1.clone().requires_grad_(True)
                li, pi, _, _ = ops.photometric_tail(batch['pts1'], x, Ui, batch['I2_aug'], batch['patch_indices'], P, graph=True)
                li.backward()
                assert torch.equal(pi, ref[0][1]) and torch.equal(x.grad, ref[0][3])
        side.synchronize()
        assert plan.stats()['captures'] - c1 == 3, plan.stats()
        # (5) ... nor does a cycle LONGER than the old 8-entry cache thrash it: 12 sets cycled four times are
        # captured once each (second sighting) and replayed afterwards -- not re-captured and evicted on every step
        c2 = plan.stats()['captures']
        for rep in range(4):
            for Ui in held[:12]:
                x = h1.clone().requires_grad_(True)
                li, pi, _, _ = ops.photometric_tail(batch['pts1'], x, Ui, batch['I2_aug'], batch['patch_indices'], P, graph=True)
                assert torch.equal(pi, ref[0][1])
            side.synchronize()
            if rep == 1:
                c3 = plan.stats()['captures']
        assert plan.stats()['captures'] - c2 <= 12 and plan.stats()['captures'] == c3, (c2, c3, plan.stats())


@pytest.mark.parametrize('kind', ['collapsed', 'overflow'])
def test_degenerate_pair_does_not_poison_the_variables(pkg, dev, kind):
    """A 100 000-step run died at ~85 k steps: ONE pair with a degenerate corner prediction gave a non-finite theta, and
    NaN flowed through d loss / d pred_h4p into every variable (profiles/r02_train_long_100k_nan_at_85k.txt).  In train
    mode the DLT backward now zeroes the gradient of such a pair (UH_DLT_ZERO_NONFINITE_GRAD): the other pairs' gradients
    are untouched, the variables' gradients stay finite.  With the guard off the NaN is passed on, as the reference does."""
    hm, synthetic, _ = pkg
    from unsuperviseddeephomographyral2018_amd import ops
    batch = synthetic.make_batch(B, H, W, P, RHO, seed=31, device=dev)
    good = torch.randn(B, 8, device=dev) * 3
    bad = good.clone()
    if kind == 'collapsed':
        bad[1] = torch.tensor([100., 100.] * 4, device=dev) - batch['pts1'][1]      # all four p2 corners on one point
    else:
        bad[1] = 1e30                                                               # products overflow -> inf - inf
    from unsuperviseddeephomographyral2018_amd import _lib
    grads, zeroed = {}, {}
    for name, h4p0, guard in (('good', good, True), ('bad_guarded', bad, True), ('bad_raw', bad, False)):
        h = h4p0.clone().requires_grad_(True)
        _lib.dlt_zeroed_pairs(reset=True)
        _, theta = ops.solve_dlt(batch['pts1'], h, W, H, zero_nonfinite_grad=guard)
        _, pred = ops.warp_gather(batch['I_aug'], theta, batch['patch_indices'], P)
        loss = ops.patch_losses(pred, batch['I2_aug'], train='l1_loss')[2]
        loss.backward()
        grads[name] = h.grad.clone()
        zeroed[name] = _lib.dlt_zeroed_pairs(reset=True)
    assert torch.isfinite(grads['bad_guarded']).all()
    # the guard is never silent: the library counts the pairs it zeroed (the trainer prints the count)
    assert zeroed['good'] == 0 and zeroed['bad_raw'] == 0
    assert zeroed['bad_guarded'] == (0 if torch.isfinite(grads['bad_raw'][1]).all() else 1)
    keep = [0, 2, 3]
    assert torch.equal(grads['bad_guarded'][keep], grads['good'][keep])             # the other pairs are untouched
    if not torch.isfinite(grads['bad_raw'][1]).all():                               # the guard fired for pair 1
        assert float(grads['bad_guarded'][1].abs().max()) == 0.0
    if kind == 'overflow':
        assert not torch.isfinite(grads['bad_raw'][1]).all()                        # guard off: NaN goes on to the regressor
    # the model switches the guard on in train mode: a full step leaves every variable gradient finite
    torch.manual_seed(0)
    net = hm.VGGRegressor(P).to(dev).to(memory_format=torch.channels_last).eval()   # (no dropout: pred_h4p == bad exactly)
    with torch.no_grad():
        offset = bad - net(torch.cat([batch['I1_aug'], batch['I2_aug']], 3))
    for graph_tail in (False, True):
        net.zero_grad(set_to_none=True)
        m = hm.HomographyModel(params(hm, 'train', B, P, W, H, 'l1_loss'), *synthetic.model_args(batch), net=net,
                               h4p_offset=offset, graph_tail=graph_tail)
        assert m.zero_nonfinite_grad
        m.loss.backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters()), graph_tail


def test_conv_bias_relu_epilogue_equals_torch(pkg, dev):
    """ops.conv_bias_relu (MIOpen conv + HIP bias/ReLU epilogue) vs F.relu(F.conv2d(x, w, b)): same forward bits,
    gradients within f32 reduction-order noise; and the whole regressor with / without the fused epilogue."""
    hm, synthetic, _ = pkg
    from unsuperviseddeephomographyral2018_amd import ops
    import torch.nn.functional as F
    g = torch.Generator(device=dev).manual_seed(0)
    for (N, Ci, Co, S) in ((3, 2, 64, 32), (2, 64, 128, 16)):
        x = torch.randn(N, Ci, S, S, generator=g, device=dev).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(Co, Ci, 3, 3, generator=g, device=dev) * 0.1).contiguous(memory_format=torch.channels_last)
        b = torch.randn(Co, generator=g, device=dev)
        gy = torch.randn(N, Co, S, S, generator=g, device=dev).contiguous(memory_format=torch.channels_last)
        outs = []
        for fused in (False, True):
            xi, wi, bi = (t.clone().requires_grad_(True) for t in (x, w, b))
            y = ops.conv_bias_relu(xi, wi, bi, 1) if fused else F.relu(F.conv2d(xi, wi, bi, 1, 1))
            y.backward(gy)
            outs.append((y.detach(), xi.grad, wi.grad, bi.grad))
        assert torch.allclose(outs[0][0], outs[1][0], rtol=0, atol=1e-5)
        for k in (1, 2, 3):
            ref = outs[0][k]; got = outs[1][k]
            assert (got - ref).abs().max() <= 1e-4 * max(1.0, float(ref.abs().max())), k
    # + the 2x2/2 max-pool fused into the epilogue
    for (N, Ci, Co, S) in ((2, 64, 64, 32), (3, 64, 128, 8)):
        x = torch.randn(N, Ci, S, S, generator=g, device=dev).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(Co, Ci, 3, 3, generator=g, device=dev) * 0.1).contiguous(memory_format=torch.channels_last)
        b = torch.randn(Co, generator=g, device=dev)
        gp = torch.randn(N, Co, S // 2, S // 2, generator=g, device=dev).contiguous(memory_format=torch.channels_last)
        outs = []
        for fused in (False, True):
            xi, wi, bi = (t.clone().requires_grad_(True) for t in (x, w, b))
            y = ops.conv_bias_relu_pool(xi, wi, bi, 1) if fused else F.max_pool2d(F.relu(F.conv2d(xi, wi, bi, 1, 1)), 2, 2)
            y.backward(gp)
            outs.append((y.detach(), xi.grad, wi.grad, bi.grad))
        assert torch.allclose(outs[0][0], outs[1][0], rtol=0, atol=1e-5)
        for k in (1, 2, 3):
            ref = outs[0][k]; got = outs[1][k]
            assert (got - ref).abs().max() <= 1e-4 * max(1.0, float(ref.abs().max())), ('pool', k)
    batch = synthetic.make_batch(B, H, W, P, RHO, seed=5, device=dev)
    torch.manual_seed(1)
    net = hm.VGGRegressor(P).to(dev).to(memory_format=torch.channels_last).eval()
    xin = torch.cat([batch['I1_aug'], batch['I2_aug']], 3)
    res = []
    for fused in (True, False):
        net.fused_epilogue = fused
        net.zero_grad(set_to_none=True)
        o = net(xin)
        o.square().mean().backward()
        res.append((o.detach(), [p.grad.clone() for p in net.parameters()]))
    assert torch.allclose(res[0][0], res[1][0], rtol=1e-4, atol=1e-4)
    for (n, _), ga, gb in zip(net.named_parameters(), res[0][1], res[1][1]):
        # below a max-pool a near-tied winner may flip between the two forward passes (see the e2e test): loose there
        tight = n.split('.')[0] in ('fc1', 'fc2') or n.startswith(('convs.6', 'convs.7'))
        assert (ga - gb).abs().max() <= (1e-3 if tight else 5e-2) * max(1e-6, float(gb.abs().max())), n


def test_whole_step_hipgraph_trains(pkg, dev):
    """--step_graph: the whole training step (convs, epilogues, hot path, capturable fused Adam) captured into one
    hipGraph and replayed.  The replayed steps must keep training (loss on a fixed batch goes down) and new batch data
    copied into the static buffers must be what the graph consumes."""
    hm, synthetic, drv = pkg
    args = drv.build_parser().parse_args(['--batch_size', str(B), '--img_h', str(H), '--img_w', str(W), '--patch_size', str(P),
                                          '--rho', str(RHO), '--loss_type', 'l1_loss', '--step_graph', 'True', '--lr', '1e-4'])
    torch.manual_seed(0)
    step = drv.TrainStep(args, dev, 1)
    batch = synthetic.make_batch(B, H, W, P, RHO, seed=21, device=dev)
    losses = []
    for _ in range(40):
        m = step(batch)
        losses.append(float(m.l1_loss.detach()))
    assert step._graph is not None and step.global_step == 40          # 40 replays; the capture's eager warm-up is rolled back
    assert all(np.isfinite(losses)) and np.mean(losses[-5:]) < np.mean(losses[:5])
    other = synthetic.make_batch(B, H, W, P, RHO, seed=22, device=dev)
    m = step(other)
    assert torch.equal(step._static['I_aug'], other['I_aug']) and torch.equal(m.I, step._static['I_aug'])


def test_step_graph_resume_keeps_the_adam_state(pkg, dev):
    """A --step_graph run resumed from a checkpoint: the capture's eager warm-up must put back the RESTORED Adam moments and
    step counts (not zeros), so the first replayed steps match the eager steps of a run resumed from the same checkpoint.
    Dropout is what differs between the two routes' RNG consumption, so the comparison runs with it off (eval-mode net);
    with the state zeroed instead of restored, bias correction restarts and the first update is ~lr-sized in every
    coordinate: the two runs would differ by orders of magnitude more than the tolerance here."""
    hm, synthetic, drv = pkg
    common = ['--batch_size', str(B), '--img_h', str(H), '--img_w', str(W), '--patch_size', str(P), '--rho', str(RHO),
              '--loss_type', 'l1_loss', '--lr', '1e-4', '--tf_adam_epsilon', 'False']      # the graph route keeps torch's constant eps
    batch = synthetic.make_batch(B, H, W, P, RHO, seed=31, device=dev)
    torch.manual_seed(0)
    first = drv.TrainStep(drv.build_parser().parse_args(common), dev, 1)
    for _ in range(12):
        first(batch)
    sd = copy.deepcopy(first.state_dict())
    w_ckpt = first.net.fc2.weight.detach().clone()

    def resumed(graph):
        a = drv.build_parser().parse_args(common + ['--step_graph', 'True' if graph else 'False'])
        torch.manual_seed(5)
        st = drv.TrainStep(a, dev, 1)
        st.load_state_dict(copy.deepcopy(sd))
        import torch.nn.functional as F
        orig = F.dropout
        F.dropout = lambda x, p=0.5, training=True, inplace=False: x       # dropout off on both routes
        try:
            for _ in range(3):
                st(batch)
        finally:
            F.dropout = orig
        torch.cuda.synchronize()
        return st
    eager, graph = resumed(False), resumed(True)
    assert graph._graph is not None and graph.global_step == eager.global_step == 15
    # Adam's step counter continued from the checkpoint on both routes
    se = [float(s['step']) for s in eager.opt.state.values()]
    sg = [float(s['step']) for s in graph.opt.state.values()]
    assert se == sg and se[0] == 15.0
    moved = float((eager.net.fc2.weight - w_ckpt).abs().max())
    diff = float((eager.net.fc2.weight - graph.net.fc2.weight).abs().max())
    assert moved > 0 and diff <= 0.05 * moved, (moved, diff)
    for (n, pe), pg in zip(eager.net.named_parameters(), graph.net.parameters()):
        d = float((pe - pg).abs().max()); m = float((pe - dict(first.net.named_parameters())[n]).abs().max())
        assert d <= 0.1 * max(m, 1e-12), (n, d, m)


@pytest.mark.parametrize('loss_type', ['l1_loss', 'rec_loss'])
def test_default_model_path_gradient_values(pkg, dev, loss_type):
    """The DEFAULT model path -- f32 DLT solve (`solve_f64=False`), fused conv epilogues, full-frame warp, one-node
    transform + losses -- held to gradient VALUES at the model level (the other model-level gradient tests run
    with solve_f64=True or with the epilogue fusion off).  d loss / d pred_h4p from the HIP chain against f64 autograd of the
    reference-equivalent op graph at the SAME pred_h4p, under the mid-training theta law.  Observed on MI355X: 1e-5 per pair
    and for the whole tensor (the 120x160 / rho = 20 frames of this file keep the 8x8 system well conditioned; the golden
    chain test covers a badly conditioned one); bounds: 5e-4 for the worst pair, 2e-4 for the tensor."""
    hm, synthetic, _ = pkg
    batch = synthetic.make_batch(8, H, W, P, RHO, seed=41, device=dev)
    torch.manual_seed(3)
    net = hm.VGGRegressor(P).to(dev).to(memory_format=torch.channels_last).eval()
    assert net.fused_epilogue
    # a fresh regressor predicts ~0, i.e. the best-conditioned DLT there is: add the mid-training offset gt + N(0, 2 px)
    gen = torch.Generator(device=dev).manual_seed(9)
    off = batch['gt'] + 2.0 * torch.randn(8, 8, generator=gen, device=dev)
    m = hm.HomographyModel(params(hm, 'train', 8, P, W, H, loss_type), *synthetic.model_args(batch), net=net, h4p_offset=off)
    assert m.solve_f64 is False and m.fused_patch is False
    (g,) = torch.autograd.grad(m.loss, m.pred_h4p, retain_graph=True)
    m.loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
    h = m.pred_h4p.detach().cpu().double().requires_grad_(True)
    cpu = {k: v.detach().cpu() for k, v in batch.items()}
    ref_loss, _ = OT.photometric_loss(loss_type, cpu['I_aug'].double(), cpu['I2_aug'].double(), cpu['pts1'].double(), h,
                                      cpu['patch_indices'], W, H, P)
    ref_loss.backward()
    assert abs(float(m.loss.detach()) - float(ref_loss)) <= 1e-4 * max(1.0, abs(float(ref_loss)))
    got, ref = g.cpu().double(), h.grad
    per_pair = ((got - ref).abs().max(1).values / ref.abs().max(1).values).numpy()
    whole = float((got - ref).abs().max() / ref.abs().max())
    print('default-path gradient: per-pair relative error', np.round(per_pair, 5), 'whole tensor', whole)
    assert np.median(per_pair) < 1e-4 and per_pair.max() < 5e-4, per_pair
    assert whole < 2e-4, whole


def test_train_entry_point_from_generator_pool_and_disk(pkg, dev, tmp_path, capsys):
    """homography_CNN_synthetic.train() end to end for a few steps on each input route: the generator in the loop, a
    pre-generated pool, and the reference's on-disk layout through Dataloader with decode worker processes + prefetch.
    Log lines carry the running means and pairs/s; a checkpoint is written and --resume continues from it."""
    hm, synthetic, drv = pkg
    from unsuperviseddeephomographyral2018_amd import dataloader as dl
    common = ['--batch_size', '4', '--img_h', str(H), '--img_w', str(W), '--patch_size', str(P), '--rho', str(RHO),
              '--loss_type', 'l1_loss', '--num_total_steps', '6', '--log_every', '2', '--save_every', '100',
              '--model_dir', str(tmp_path / 'models')]
    for extra in ([], ['--data_pool', '3']):
        st = drv.train(drv.build_parser().parse_args(common + extra))
        assert st.global_step == 6
    out = capsys.readouterr().out
    assert out.count('Train: step') == 6 and 'pairs/s' in out and 'l1_loss' in out
    # on-disk route
    b = synthetic.make_batch(12, H, W, P, RHO, seed=77, device=dev)
    to_u8 = lambda t: (t * 50.0 + 128.0).clamp(0, 255).to(torch.uint8).cpu().numpy()
    ff, fp, fg = dl.write_dataset(str(tmp_path / 'data') + '/', to_u8(b['I_aug']), to_u8(b['I_prime_aug']), b['pts1'].cpu().numpy(),
                                  b['gt'].cpu().numpy(), fmt='jpg')
    disk = ['--data_path', str(tmp_path / 'data') + '/', '--filenames_file', ff, '--pts1_file', fp, '--gt_file', fg]
    for workers in ('0', '2'):
        st = drv.train(drv.build_parser().parse_args(common + disk + ['--decode_workers', workers]))
        assert st.global_step == 6 and all(torch.isfinite(p).all() for p in st.net.parameters())
    st = drv.train(drv.build_parser().parse_args(common + ['--resume', 'True']))
    assert st.global_step == 12                         # continued from the checkpoint of the previous run
    out = capsys.readouterr().out
    assert '===> Start step: 6' in out


def test_zeroed_pairs_counter_is_read_on_the_callers_stream_and_refuses_a_capture(dev):
    """uh_dlt_zeroed_pairs (ABI 7): synchronous on the given stream only -- a count taken on a side stream sees the backward
    launched there before it; reset clears; a capturing stream is refused with UH_E_CAPTURING and the capture survives."""
    import ctypes as C
    from unsuperviseddeephomographyral2018_amd import _lib, ops
    lib = _lib.load()
    _lib.dlt_zeroed_pairs(reset=True)
    pts1 = torch.tensor([[10., 10., 50., 10., 50., 50., 10., 50.]] * 3, device=dev)
    h4p = torch.zeros(3, 8, device=dev)
    h4p[1] = 1e30                                                          # products overflow -> inf - inf -> NaN gradient
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        hp = h4p.clone().requires_grad_(True)
        Hm, theta = ops.solve_dlt(pts1, hp, 64, 64, zero_nonfinite_grad=True)
        theta.sum().backward()
        n = _lib.dlt_zeroed_pairs(reset=False)                            # current stream = side
        assert n == 1 and torch.all(hp.grad[1] == 0) and torch.isfinite(hp.grad).all()
        assert _lib.dlt_zeroed_pairs(reset=True) == 1 and _lib.dlt_zeroed_pairs() == 0
        g = torch.cuda.CUDAGraph()
        buf = torch.zeros(4, device=dev)
        with torch.cuda.graph(g, stream=side):
            buf += 1
            cnt = C.c_ulonglong(7)
            rc = lib.uh_dlt_zeroed_pairs(C.byref(cnt), 0, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            assert rc == -6 and cnt.value == 7                            # refused, nothing written, capture intact
        g.replay()
    torch.cuda.synchronize(dev)
    assert torch.all(buf == 1)


def test_use_batch_norm_flag_keeps_the_reference_quirk(pkg, dev):
    """--use_batch_norm True (homography_model.py:92-93,356): slim.batch_norm(out, self.is_training) passes is_training as the DECAY,
    so the moving statistics never move (decay 1 in train mode; their update ops are never run either) -- training normalises with
    batch statistics, test mode with the initial mean 0 / variance 1.  The mirror keeps that: momentum 0, centre only (no scale),
    eps 1e-3, BN after the ReLU; the fused epilogues step aside for the torch ops."""
    hm, synthetic, drv = pkg
    args = drv.build_parser().parse_args(['--batch_size', str(B), '--img_h', str(H), '--img_w', str(W), '--patch_size', str(P),
                                          '--rho', str(RHO), '--use_batch_norm', 'True'])
    assert args.use_batch_norm is True
    torch.manual_seed(0)
    step = drv.TrainStep(args, dev, world=1)
    net = step.net
    assert net.use_batch_norm and len(net.bns) == 8 and all(not bn.weight.requires_grad and bn.bias.requires_grad for bn in net.bns)
    batch = synthetic.make_batch(B, H, W, P, RHO, seed=8, device=dev)
    losses = [float(step(batch).l1_loss.detach()) for _ in range(5)]
    assert np.isfinite(losses).all()
    for bn in net.bns:                                          # decay = 1: the moving statistics are still the initial ones
        assert float(bn.running_mean.abs().max()) == 0.0 and float((bn.running_var - 1).abs().max()) == 0.0
        assert float(bn.bias.grad.abs().max()) > 0              # beta trains
    x = torch.cat([batch['I1_aug'], batch['I2_aug']], 3)
    with torch.no_grad():
        net.train(); torch.manual_seed(1); y_train = net(x)
        net.eval(); y_test = net(x)
        net.train()
    assert torch.isfinite(y_test).all() and float((y_train - y_test).abs().max()) > 1e-3      # batch statistics vs none
    res = drv.TestHomography(args, step_fn=step).run()
    assert res['mean_corner_error'] > 0
