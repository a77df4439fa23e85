"""CPU: the trainer's Adam update is tf.train.AdamOptimizer's (the reference's optimizer, /root/reference/code/
homography_CNN_synthetic.py:161-183), not torch.optim.Adam's with a constant eps.  TF 1.x documents its rule as
    lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t);  m_t = b1 m + (1 - b1) g;  v_t = b2 v + (1 - b2) g^2;  var -= lr_t m_t / (sqrt(v_t) + eps)
("epsilon hat" of the paper).  homography_CNN_synthetic.tf_adam_eps gives the per-step torch eps that reproduces it; TrainStep sets
it next to the staircase learning rate."""
import math

import numpy as np
import pytest

torch = pytest.importorskip('torch')


def tf_adam_numpy(x0, grads, lrs, b1=0.9, b2=0.999, eps=1e-8):
    x = x0.astype(np.float64).copy(); m = np.zeros_like(x); v = np.zeros_like(x)
    for t, (g, lr) in enumerate(zip(grads, lrs), 1):
        g = g.astype(np.float64)
        lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        m = b1 * m + (1 - b1) * g
        v = b2 * v + (1 - b2) * g * g
        x = x - lr_t * m / (np.sqrt(v) + eps)
    return x


def test_per_step_eps_reproduces_the_tensorflow_adam_update():
    from unsuperviseddeephomographyral2018_amd.homography_CNN_synthetic import staircase_lr, tf_adam_eps
    assert abs(tf_adam_eps(1) - 1e-8 / math.sqrt(1e-3)) < 1e-20 and tf_adam_eps(10 ** 6) == pytest.approx(1e-8, rel=1e-12)
    rs = np.random.RandomState(0)
    n, steps = 64, 60
    x0 = rs.randn(n)
    scale = np.concatenate([np.full(n // 2, 1.0), np.full(n // 4, 1e-8), np.full(n // 4, 1e-10)])     # ordinary, eps-sized and tiny gradients
    grads = [rs.randn(n) * scale for _ in range(steps)]
    lrs = [staircase_lr(1e-3, t, 20.5) for t in range(steps)]                                         # a staircase that decays inside the run
    want = tf_adam_numpy(x0, grads, lrs)

    def run(tf_rule):
        p = torch.nn.Parameter(torch.tensor(x0, dtype=torch.float64))
        opt = torch.optim.Adam([p], lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
        for t, (g, lr) in enumerate(zip(grads, lrs), 1):
            for grp in opt.param_groups:
                grp['lr'] = lr
                if tf_rule:
                    grp['eps'] = tf_adam_eps(t, 1e-8, grp['betas'][1])
            p.grad = torch.tensor(g, dtype=torch.float64)
            opt.step()
        return p.detach().numpy()
    got = run(True)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-15)
    # a constant eps is a different rule where gradients are eps-sized: visible on those coordinates, invisible on ordinary ones
    plain = run(False)
    moved = np.abs(want - x0)
    rel = np.abs(plain - want) / np.maximum(moved, 1e-300)
    assert rel[:n // 2].max() < 1e-4 and rel[n // 2:].max() > 1e-2


def test_trainstep_sets_the_rule_and_checkpoints_its_counter():
    """TrainStep on the CPU (construction and bookkeeping only: a step needs the MI355X): eps follows adam_t, the counter is
    saved / restored, --retrain resets the step counter but not Adam's t (TF restores the beta-power accumulators), and the
    whole-step graph mode keeps torch's constant eps."""
    from unsuperviseddeephomographyral2018_amd import homography_CNN_synthetic as drv
    args = drv.build_parser().parse_args(['--patch_size', '16', '--batch_size', '2'])
    assert args.tf_adam_epsilon is True
    ts = drv.TrainStep(args, torch.device('cpu'))
    assert ts.tf_adam_epsilon and ts.adam_t == 0
    ts.adam_t, ts.global_step = 41, 41
    sd = ts.state_dict()
    assert sd['adam_t'] == 41
    ts2 = drv.TrainStep(args, torch.device('cpu'))
    ts2.load_state_dict(sd, retrain=True)
    assert ts2.global_step == 0 and ts2.adam_t == 41
    del sd['adam_t']                                           # a checkpoint written before round 6: falls back to the optimizer state
    ts3 = drv.TrainStep(args, torch.device('cpu'))
    ts3.load_state_dict(sd)
    assert ts3.adam_t == 0 and ts3.global_step == 41          # (no optimizer state yet in this synthetic checkpoint)
    args_off = drv.build_parser().parse_args(['--patch_size', '16', '--batch_size', '2', '--tf_adam_epsilon', 'False'])
    assert drv.TrainStep(args_off, torch.device('cpu')).tf_adam_epsilon is False
