"""The PRODUCT data-parallel step on two ranks equals the one-rank step (VERDICT r3 "What's missing" #1).

Two processes share the one GPU of the test box (gloo: RCCL wants a device per rank; the collective's transport is not what
is under test, the step's construction is): each runs TrainStep(world=2) -- the real VGG regressor, HomographyModel over the
HIP hot path, GradAverager with its two flat buckets and post-accumulate hooks, fused Adam -- on ITS contiguous half of a
fixed batch, for three steps.  Rank 0 then replays the same three steps as ONE tower on the full batch, from the same
initial variables, and compares.  Reference semantics: tf.split of the batch over towers
(/root/reference/code/homography_CNN_synthetic.py:199-207), one model per tower sharing the variables (:229-257),
get_average_grads = per-variable mean of the tower gradients (utils/utils.py:380-403), ONE Adam update (:277-278).

  * l1_loss (photometric): mean over B*P*P pixels = mean of the shard means  =>  world 2 == world 1 up to f32 reduction order.
  * h_loss (supervised): an RMSE PER TOWER (homography_model.py:288)  =>  world 2 == the mean of the per-shard RMSE gradients
    and NOT the full-batch RMSE gradient; both halves of that statement are asserted.
  * dropout: every tower draws its own masks (one slim.dropout op per tower, homography_model.py:120-121,128): ranks seed
    the model RNG with seed + rank AFTER the variable broadcast; variables stay rank-equal all the same.
"""
import os
import socket

import pytest

torch = pytest.importorskip('torch')
import torch.multiprocessing as mp       # noqa: E402

pytestmark = pytest.mark.gpu

B, H, W, P, RHO, STEPS = 8, 240, 320, 128, 45, 3


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _args(loss_type, batch):
    from unsuperviseddeephomographyral2018_amd.homography_CNN_synthetic import build_parser
    return build_parser().parse_args(['--mode', 'train', '--loss_type', loss_type, '--batch_size', str(batch),
                                      '--img_h', str(H), '--img_w', str(W), '--patch_size', str(P), '--rho', str(RHO),
                                      '--seed', '3'])


def _worst(a, b, names, top=3):
    """the `top` tensors with the largest relative L2 difference: [(name, rel L2, max |d| / max |ref|)]"""
    rows = []
    for n, x, y in zip(names, a, b):
        x, y = x.double(), y.double()
        rows.append((n, float((x - y).norm() / y.norm().clamp_min(1e-30)), float((x - y).abs().max() / y.abs().max().clamp_min(1e-30))))
    rows.sort(key=lambda r: -r[1])
    return [(n, float('%.3g' % a_), float('%.3g' % b_)) for n, a_, b_ in rows[:top]]


def _diff(a, b):
    """(max |a-b| / max |b|, ||a-b|| / ||b||) over a list of tensors, in f64."""
    worst, num, den = 0.0, 0.0, 0.0
    for x, y in zip(a, b):
        x, y = x.double(), y.double()
        worst = max(worst, float((x - y).abs().max() / y.abs().max().clamp_min(1e-30)))
        num += float(((x - y) ** 2).sum()); den += float((y ** 2).sum())
    return worst, (num / max(den, 1e-300)) ** 0.5


def _private_miopen_db():
    """These comparisons need MIOpen's deterministic solvers.  In immediate mode (cudnn.benchmark off) MIOpen still consults the
    USER find-db: entries left there by an earlier process that ran the same conv shapes with find mode on (the trainer tests
    of this file do, at the same per-tower batch) make it pick the recorded fastest solver -- split-K with atomics -- and one
    gradient computed twice then differs in the last bits.  A fresh box has no such entries; a box that already ran the suite
    does.  An empty private user db per worker process makes both behave like the fresh box."""
    import tempfile
    os.environ['MIOPEN_USER_DB_PATH'] = tempfile.mkdtemp(prefix='uh_miopen_db_')
    # ... and the deterministic route needs MIOpen's reference solvers: a trainer that ran in the PARENT pytest process
    # (dist.skip_naive_conv_in_find: MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_*=0, inherited by spawned workers) leaves it with
    # "No suitable algorithm was found" for the 2-channel first layer
    for k in [k for k in os.environ if k.startswith('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_')]:
        del os.environ[k]


def _collect(q, procs, n, timeout):
    """n results from the workers' queue -- failing as soon as a worker has died instead of waiting out the whole timeout."""
    import queue as _queue
    import time as _time
    out, t0 = [], _time.time()
    while len(out) < n:
        try:
            out.append(q.get(timeout=2))
        except _queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead:
                raise AssertionError('a worker process died (exit codes %r) before reporting' % dead)
            if _time.time() - t0 > timeout:
                raise AssertionError('workers did not report within %d s' % timeout)
    return out


def _worker(rank, world, port, q, loss_type, dropout_p):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    _private_miopen_db()
    import copy
    from unsuperviseddeephomographyral2018_amd import _lib, dist as D, synthetic
    from unsuperviseddeephomographyral2018_amd.homography_CNN_synthetic import TrainStep, tf_adam_eps
    from unsuperviseddeephomographyral2018_amd.homography_model import HomographyModel, VGGRegressor
    _lib.load()
    # MIOpen's default conv solvers are not run-to-run reproducible (split-K / atomic accumulation): with
    # them, ONE gradient computed twice differs by ~1e-4 relative, which would drown what this test is after.
    # cudnn.deterministic restricts PyTorch-ROCm to MIOpen's GEMM algorithms; UH_TEST_NONDET=1 runs the default solvers.
    torch.backends.cudnn.benchmark = False
    torch.backends.cudnn.deterministic = os.environ.get('UH_TEST_NONDET', '0') != '1'
    r, w, local = D.init_from_env(backend='gloo')          # two ranks, one device
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(11 + rank)                           # DIFFERENT initial variables per rank: the broadcast must fix that
    net = VGGRegressor(P, dropout_p=dropout_p)
    step = TrainStep(_args(loss_type, B), dev, world, net=net)
    init = copy.deepcopy(step.net.state_dict())            # after the broadcast = rank 0's variables
    # the first thing a tower draws after construction: its dropout stream (seed + rank)
    probe = torch.nn.functional.dropout(torch.ones(4096, device=dev), 0.5, True)
    D.seed_tower_rng(step.args.seed, rank)                 # put the stream back where TrainStep left it
    full = synthetic.make_batch(B, H, W, P, RHO, seed=5, device=dev, kind=os.environ.get('UH_TEST_TEXTURE', 'white'))
    mine = {k: D.shard(v, rank, world).contiguous() for k, v in full.items()}
    t_find = step.prime_conv_finds(mine)                   # rank 0 alone, then rank 1: no collective may be issued in there
    D.seed_tower_rng(step.args.seed, rank)
    grads1 = None
    losses = []
    for it in range(STEPS):
        m = step(mine)
        losses.append(float(m.loss.detach()))
        if it == 0:
            grads1 = [p.grad.detach().clone() for p in step.net.parameters()]
    torch.cuda.synchronize(dev)
    var_dp = [p.detach().clone() for p in step.net.parameters()]
    # rank equality, exactly: the same averaged gradient and the same Adam on the same variables
    sums = torch.stack([v.double().sum() for v in var_dp] + [g.double().abs().sum() for g in grads1])
    gathered = [torch.empty_like(sums) for _ in range(world)]
    torch.distributed.all_gather(gathered, sums)
    rank_equal = all(torch.equal(gathered[0], g) for g in gathered)
    masks = [torch.empty_like(probe) for _ in range(world)]
    torch.distributed.all_gather(masks, probe)
    masks_differ = not torch.equal(masks[0], masks[1])
    mask_keep = [float(mk.ne(0).float().mean()) for mk in masks]
    out = {'rank': rank, 'rank_equal': rank_equal, 'masks_differ': masks_differ, 'mask_keep': mask_keep, 'losses': losses,
           't_find': t_find, 'buckets': [b['flat'].numel() for b in step.averager.buckets],
           'moved': max(float((v - init[n]).abs().max()) for v, (n, _) in zip(var_dp, step.net.named_parameters()))}
    torch.distributed.barrier()
    if rank == 0 and dropout_p == 0.0:
        def fresh(batch_size, world_=1):
            n = VGGRegressor(P, dropout_p=dropout_p)
            s = TrainStep(_args(loss_type, batch_size), dev, world_, net=n)
            s.net.load_state_dict(init)
            return s
        # (1) ONE tower on the full batch
        one = fresh(B)
        g_one = None
        for it in range(STEPS):
            one(full)
            if it == 0:
                g_one = [p.grad.detach().clone() for p in one.net.parameters()]
        var_one = [p.detach().clone() for p in one.net.parameters()]
        # the noise floor of the comparison: the SAME one-tower step computed a second time
        again = fresh(B)
        again(full)
        g_again = [p.grad.detach().clone() for p in again.net.parameters()]
        out['self_noise_grad'] = _diff(g_again, g_one)
        names = [n for n, _ in one.net.named_parameters()]
        out['worst_grad_vs_one_tower'] = _worst(grads1, g_one, names)
        out['grad_vs_one_tower'] = _diff(grads1, g_one)
        out['var_vs_one_tower'] = _diff(var_dp, var_one)
        # (2) the reference's in-process tower loop: per-shard losses, mean of the tower gradients, one Adam update
        two = fresh(B // world)                             # its model_params carry the per-tower batch size
        g_two = None
        for it in range(STEPS):
            for g in two.opt.param_groups:                  # what TrainStep._step sets: staircase lr + TF's epsilon-hat rule
                g['lr'] = two.learning_rate()
                g['eps'] = tf_adam_eps(it + 1, 1e-8, g['betas'][1])
            two.opt.zero_grad(set_to_none=True)
            tower_losses = []
            for k in range(world):
                sh = {kk: D.shard(v, k, world).contiguous() for kk, v in full.items()}
                tm = HomographyModel(two.model_params, *synthetic.model_args(sh), reuse_variables=True, net=two.net,
                                     zero_nonfinite_grad=True)
                tower_losses.append(tm.loss)
            (sum(tower_losses) / world).backward()
            two.opt.step(); two.global_step += 1
            if it == 0:
                g_two = [p.grad.detach().clone() for p in two.net.parameters()]
        var_two = [p.detach().clone() for p in two.net.parameters()]
        out['grad_vs_tower_loop'] = _diff(grads1, g_two)
        out['var_vs_tower_loop'] = _diff(var_dp, var_two)
        out['grad_one_vs_tower_loop'] = _diff(g_one, g_two)
    q.put(out)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _run(loss_type, dropout_p):
    if not torch.cuda.is_available():
        pytest.skip('needs the MI355X')
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, loss_type, dropout_p)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        out = sorted(_collect(q, procs, world, 600), key=lambda o: o['rank'])
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    for o in out:
        assert o['rank_equal'], 'variables / averaged gradients differ between the ranks'
        assert o['buckets'][0] == 1024 * 16 * 16 * 128 + 1024 + 8 * 1024 + 8      # fc1 + fc2: the first bucket of DESIGN section 6
        assert all(l == l for l in o['losses'])
        assert o['moved'] > 1e-4                                # three Adam steps (lr 1e-4 each) did move the variables
    return out


def test_product_dp_step_l1_equals_one_tower_full_batch():
    """TrainStep(world=2) on the two halves == TrainStep(world=1) on the whole batch, photometric l1_loss, dropout off."""
    out = _run('l1_loss', 0.0)
    r0 = out[0]
    print('l1_loss world 2 vs one tower:', {k: v for k, v in r0.items() if 'vs' in k or 'noise' in k}, 'find pass %.1f s' % r0['t_find'])
    # first-step averaged gradient: f32 reduction-order noise only (conv weight gradients of batch 4 + 4 vs batch 8)
    assert r0['grad_vs_one_tower'][1] <= 1e-5 and r0['grad_vs_one_tower'][0] <= 1e-4
    assert r0['grad_vs_tower_loop'][1] <= 1e-5
    # variables after three Adam steps (lr 1e-4 each): relative L2 over all 34.19 M variables
    assert r0['var_vs_one_tower'][1] <= 1e-5, r0['var_vs_one_tower']
    assert r0['var_vs_tower_loop'][1] <= 1e-5, r0['var_vs_tower_loop']


def test_product_dp_step_h_loss_is_the_mean_of_per_tower_rmse_gradients():
    """Supervised h_loss is an RMSE per tower (homography_model.py:288): two ranks reproduce the reference's tower loop and
    DIFFER from one tower on the full batch -- on purpose."""
    out = _run('h_loss', 0.0)
    r0 = out[0]
    print('h_loss world 2:', {k: v for k, v in r0.items() if 'vs' in k or 'noise' in k})
    assert r0['grad_vs_tower_loop'][1] <= 1e-5 and r0['var_vs_tower_loop'][1] <= 1e-5
    # sqrt(mean) over 8 pairs vs the mean of two sqrt(mean) over 4: a different gradient unless the shard RMSEs coincide
    assert r0['grad_one_vs_tower_loop'][1] >= 1e-3
    assert r0['grad_vs_one_tower'][1] >= 1e-3


def test_product_dp_towers_draw_different_dropout_masks_and_stay_in_step():
    """Dropout on (the product default): the two towers' first masks differ (seed + rank), keep-rate ~ 0.5 on both, and the
    variables are still bit-equal across ranks after three steps."""
    out = _run('l1_loss', 0.5)
    assert out[0]['masks_differ'] and out[1]['masks_differ']
    for keep in out[0]['mask_keep']:
        assert 0.45 < keep < 0.55


def _rccl_worker(q, port):
    """ONE rank, backend "nccl" (= RCCL): the collectives are identities, the CODE PATH is the multi-GPU one."""
    os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', LOCAL_WORLD_SIZE='1', MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    _private_miopen_db()
    import copy
    import torch.distributed as dist
    from unsuperviseddeephomographyral2018_amd import _lib, dist as D, synthetic
    from unsuperviseddeephomographyral2018_amd.homography_CNN_synthetic import TrainStep
    from unsuperviseddeephomographyral2018_amd.homography_model import VGGRegressor
    _lib.load()
    torch.backends.cudnn.benchmark = False
    torch.backends.cudnn.deterministic = True
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend='nccl', rank=0, world_size=1)
    out = {'backend': dist.get_backend()}
    try:
        out['rccl_version'] = '.'.join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:                                  # noqa: BLE001
        out['rccl_version'] = 'unknown (%s)' % e
    # the averaging collective itself
    t = torch.arange(8, dtype=torch.float32, device=dev)
    h = dist.all_reduce(t, op=dist.ReduceOp.AVG, async_op=True); h.wait()
    torch.cuda.synchronize(dev)
    out['avg_identity'] = bool(torch.equal(t.cpu(), torch.arange(8, dtype=torch.float32)))
    # the product step with the data-parallel machinery ON (world = 2 as far as TrainStep is concerned: variable broadcast,
    # per-tower seed, staggered find pass with its barriers, flat buckets, hooks issuing async AVG all-reduces on RCCL's
    # stream, finish() waiting on them before fused Adam) against the plain one-tower step
    Bt = 4
    torch.manual_seed(21)
    net = VGGRegressor(P, dropout_p=0.0)
    dp = TrainStep(_args('l1_loss', 2 * Bt), dev, 2, net=net)          # per-tower batch = 2*Bt / 2
    out['avg_in_collective'] = dp.averager._avg_in_collective
    init = copy.deepcopy(dp.net.state_dict())
    batch = synthetic.make_batch(Bt, H, W, P, RHO, seed=9, device=dev, kind='white')
    out['t_find'] = dp.prime_conv_finds(batch)
    for _ in range(STEPS):
        dp(batch)
    torch.cuda.synchronize(dev)
    var_dp = [p.detach().clone() for p in dp.net.parameters()]
    net1 = VGGRegressor(P, dropout_p=0.0)
    one = TrainStep(_args('l1_loss', Bt), dev, 1, net=net1)
    one.net.load_state_dict(init)
    for _ in range(STEPS):
        one(batch)
    torch.cuda.synchronize(dev)
    var_one = [p.detach().clone() for p in one.net.parameters()]
    out['var_diff'] = _diff(var_dp, var_one)
    out['moved'] = max(float((v - init[n]).abs().max()) for v, (n, _) in zip(var_dp, dp.net.named_parameters()))
    out['buckets_ms'] = [b['ms'] for b in dp.averager.time_buckets(iters=3)]
    dist.barrier()
    dist.destroy_process_group()
    q.put(out)


def test_product_step_over_rccl_one_rank_equals_plain_step():
    """The RCCL leg of the data-parallel step has never met hardware with more than one GPU (every box of the pool has one).
    What CAN run here: a one-rank "nccl" group, where every collective is an identity -- so TrainStep with its data-parallel
    machinery switched on (broadcast, barriers of the staggered find pass, bucket hooks issuing asynchronous
    ReduceOp.AVG all-reduces on RCCL's own stream, finish() ordering them before fused Adam) must reproduce the plain step
    bit for bit.  Catches an RCCL build without AVG, a missing stream dependency between the collective and Adam, a
    barrier that needs a device it was not given."""
    if not torch.cuda.is_available():
        pytest.skip('needs the MI355X')
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(q, _free_port()))
    p.start()
    try:
        out = _collect(q, [p], 1, 600)[0]
    finally:
        p.join(timeout=120)
        if p.is_alive():
            p.kill()
    print('one-rank RCCL step:', out)
    assert p.exitcode == 0
    assert out['backend'] == 'nccl' and out['avg_identity'] and out['avg_in_collective']
    assert out['moved'] > 1e-4
    assert out['var_diff'][1] <= 1e-7 and out['var_diff'][0] <= 1e-5, out['var_diff']     # measured: (0.0, 0.0)


def _train_worker(rank, world, port, q, tmp):
    """train() itself on two gloo ranks sharing the GPU: 21 steps, --log_every 10."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port), UH_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    import contextlib
    import io
    import torch.distributed as dist
    from unsuperviseddeephomographyral2018_amd import homography_CNN_synthetic as drv
    seen = []
    real = dist.all_reduce

    def counting(t, *a, **k):
        seen.append((str(t.dtype), int(t.numel())))
        return real(t, *a, **k)
    dist.all_reduce = counting
    args = drv.build_parser().parse_args(
        ['--mode', 'train', '--loss_type', 'l1_loss', '--batch_size', '8', '--num_gpus', '2', '--num_total_steps', '21',
         '--log_every', '10', '--save_every', '1000', '--model_dir', os.path.join(tmp, 'models'), '--tunable_gemm', 'False',
         '--seed', '5'])
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        step_fn = drv.train(args)
    flat = torch.cat([p.detach().flatten() for p in step_fn.net.parameters()])
    both = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    q.put({'rank': rank, 'monitor_collectives': sum(1 for d, n in seen if d == 'torch.float64' and n == 14),
           'small_f32_collectives': sum(1 for d, n in seen if d == 'torch.float32' and n <= 16),
           'stdout': buf.getvalue(), 'in_step': bool(torch.equal(both[0], both[1])), 'global_step': step_fn.global_step})
    dist.barrier()
    dist.destroy_process_group()


def test_train_two_ranks_reduces_its_monitors_at_log_steps_only(tmp_path):
    """train() at world 2: the six loss monitors are summed per rank and all-reduced ONLY at log steps (one
    14-float collective at steps 0, 10, 20 of a 21-step run -- SURVEY 8e, reference :279-284,345-352), never per step; the final
    line carries dist.exchange_report's object; and the extra no-exchange steps it runs leave no trace: both ranks hold the same
    variables afterwards and the step counter says 21."""
    if not torch.cuda.is_available():
        pytest.skip('needs the MI355X')
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    try:
        out = sorted(_collect(q, procs, world, 900), key=lambda o: o['rank'])
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    import json
    for o in out:
        assert o['monitor_collectives'] == 3, o['monitor_collectives']
        assert o['small_f32_collectives'] == 0                      # the per-step 6-float mean of rounds 2-5 is gone
        assert o['in_step'] and o['global_step'] == 21
    log = out[0]['stdout']
    assert log.count('Train: step') == 3 and 'Train: step 20' in log
    line = [l for l in log.splitlines() if l.startswith('===> exchange (world 2, gloo): ')]
    assert len(line) == 1
    rep = json.loads(line[0].split(': ', 1)[1])
    assert rep['ms_per_step_with_exchange'] > 0 and rep['ms_per_step_no_exchange'] > 0 and len(rep['buckets']) == 2
    assert abs(rep['exchange_cost_ms_per_step'] - (rep['ms_per_step_with_exchange'] - rep['ms_per_step_no_exchange'])) < 2e-3
    assert out[1]['stdout'].count('Train: step') == 0              # rank 0 alone prints


def test_readme_command_line_with_num_gpus_2_runs_end_to_end(tmp_path):
    """The reference README's first command line (README.md:125: `--mode train --lr 5e-4 --loss_type h_loss --visual True`) plus
    `--num_gpus 2`, as a user would type it: the program re-executes itself under torch.distributed.run with two ranks (sharing
    this box's GPU over gloo, and saying so), trains, logs from rank 0 and writes the checkpoint.  VERDICT r5 item 1."""
    if not torch.cuda.is_available():
        pytest.skip('needs the MI355X')
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get('PYTHONPATH', ''))
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'unsuperviseddeephomographyral2018_amd.homography_CNN_synthetic',
           '--mode', 'train', '--lr', '5e-4', '--loss_type', 'h_loss', '--visual', 'True',          # README.md:125, verbatim
           '--num_gpus', '2', '--batch_size', '8', '--num_total_steps', '3', '--log_every', '1', '--exchange_report', 'False',
           '--tunable_gemm', 'False', '--model_dir', str(tmp_path / 'models')]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert 'world size: 2' in r.stdout and r.stdout.count('Train: step') == 3 and 'Train: step 2 ' in r.stdout
    assert '--visual True' in r.stderr
    if torch.cuda.device_count() < 2:
        assert 'share the visible device(s) over gloo' in r.stderr
    assert os.path.exists(str(tmp_path / 'models' / 'h_loss_normalize' / 'model.ckpt.pt'))
