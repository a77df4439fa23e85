"""CPU, world_size 2, gloo: the data-parallel exchange step (dist.GradAverager) reproduces
get_average_grads (/root/reference/code/utils/utils.py:380-403): per-tower gradients of per-shard mean
losses, averaged variable by variable -- which equals the single-process gradient of the full-batch mean."""
import os
import socket

import pytest

torch = pytest.importorskip('torch')
import torch.multiprocessing as mp       # noqa: E402
import torch.nn as nn                    # noqa: E402


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


class Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.convs = nn.ModuleList([nn.Conv2d(2, 4, 3, padding=1), nn.Conv2d(4, 4, 3, padding=1)])
        self.fc1 = nn.Linear(4 * 8 * 8, 16)
        self.fc2 = nn.Linear(16, 8)

    def forward(self, x):
        for c in self.convs:
            x = torch.relu(c(x))
        return self.fc2(torch.relu(self.fc1(x.flatten(1))))


def _worker(rank, world, port, q, backend='gloo'):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    from unsuperviseddeephomographyral2018_amd import dist as D
    r, w, local = D.init_from_env(backend=backend)
    assert (r, w) == (rank, world)
    dev = torch.device('cuda', local) if backend == 'nccl' else torch.device('cpu')
    torch.manual_seed(0)
    net = Tiny().to(dev)
    for p in net.parameters():
        torch.distributed.broadcast(p.data, src=0)
    g = torch.Generator().manual_seed(1)
    X = torch.randn(8, 2, 8, 8, generator=g).to(dev); Y = torch.randn(8, 8, generator=g).to(dev)
    avg = D.GradAverager(net, world)
    assert avg._avg_in_collective == (backend == 'nccl')        # RCCL averages inside the collective (ReduceOp.AVG)
    assert len(avg.buckets) == 2 and avg.buckets[0]['flat'].numel() == sum(
        p.numel() for n, p in net.named_parameters() if n.startswith('fc'))
    res = []
    for it in range(2):                      # two steps: buffers are re-zeroed, hooks re-armed
        avg.reset()
        xs, ys = D.shard(X, rank, world), D.shard(Y, rank, world)
        loss = ((net(xs) - ys) ** 2).mean()          # mean over the LOCAL shard, like one tower
        loss.backward()
        avg.finish()
        res.append([p.grad.clone() for p in net.parameters()])
    vals = D.all_reduce_mean_scalars([loss], world)
    tb = avg.time_buckets(iters=2)                 # the stand-alone exchange timing bench.py --gpus N reports
    assert len(tb) == 2 and all(b['ms'] > 0 and b['bytes'] > 0 for b in tb)
    assert avg.finishes == 2 and avg.exposed_wait_s >= 0.0
    # the self-diagnosing part of bench.py --gpus N (dist.exchange_report): bucket timing with busbw against the xGMI figure,
    # and the step WITHOUT the exchange -- every rank runs steps with the averager disabled (no collective is issued)
    calls = []

    def run_steps(n):
        for _ in range(n):
            avg.reset()
            ((net(D.shard(X, rank, world)) - D.shard(Y, rank, world)) ** 2).mean().backward()
            assert avg.enabled is False and avg._handles == []           # disabled: the hooks issued nothing
            avg.finish()
            calls.append(1)
    fin_before = avg.finishes
    rep = D.exchange_report(avg, run_steps, ms_per_step=5.0, steps_without=3, bucket_iters=2)
    assert avg.enabled is True and len(calls) == 2 + 3 and avg.finishes == fin_before       # finish() is a no-op when disabled
    assert rep['reduce_op'] == ('AVG' if backend == 'nccl' else 'SUM+div') and rep['steps_without_exchange'] == 3
    assert rep['ms_per_step_no_exchange'] > 0 and abs(rep['exchange_cost_ms_per_step'] - (5.0 - rep['ms_per_step_no_exchange'])) < 2e-3
    assert rep['xgmi'] == dict(rep['xgmi'], links_to_peers=1, GBs_per_link=153.0, peak_GBs=153.0)
    assert all(abs(b['busbw_frac_of_xgmi'] - b['busbw_GBs'] / 153.0) < 1e-3 for b in rep['buckets'])
    q.put((rank, [[t.cpu().numpy() for t in r_] for r_ in res], float(vals[0])))
    torch.distributed.destroy_process_group()


def test_grad_averager_world2_equals_full_batch_gradient():
    _run_world2('gloo')


@pytest.mark.gpu
def test_grad_averager_world2_rccl():
    """The same exchange step over RCCL (backend "nccl") on two real GPUs: rank-equal averaged gradients that equal the
    one-process full-batch gradient.  Activates itself on any box with >= 2 devices (the driver's 8-GPU node)."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip('needs >= 2 HIP devices (RCCL wants one device per rank)')
    _run_world2('nccl')


def _run_world2(backend):
    import numpy as np
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out.sort(key=lambda t: t[0])
    # reference: one process, full batch
    torch.manual_seed(0)
    net = Tiny()
    g = torch.Generator().manual_seed(1)
    X = torch.randn(8, 2, 8, 8, generator=g); Y = torch.randn(8, 8, generator=g)
    ((net(X) - Y) ** 2).mean().backward()
    ref = [p.grad.numpy() for p in net.parameters()]
    for it in range(2):
        for a, b, r in zip(out[0][1][it], out[1][1][it], ref):
            np.testing.assert_allclose(a, b, rtol=0, atol=0)          # both ranks hold the same mean
            np.testing.assert_allclose(a, r, rtol=1e-5 if backend == 'gloo' else 1e-4, atol=1e-7 if backend == 'gloo' else 1e-6)
    assert abs(out[0][2] - out[1][2]) < 1e-7


def _worker_resync(rank, world, port, q):
    """exchange_report's no-exchange steps are real optimizer steps on per-tower gradients: the replicas diverge.  With
    resync=(module, optimizer) every rank leaves the call holding rank 0's variables and Adam state."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from unsuperviseddeephomographyral2018_amd import dist as D
    D.init_from_env(backend='gloo')
    torch.manual_seed(0)
    net = Tiny()
    for p in net.parameters():
        torch.distributed.broadcast(p.data, src=0)
    g = torch.Generator().manual_seed(1)
    X = torch.randn(8, 2, 8, 8, generator=g); Y = torch.randn(8, 8, generator=g)
    avg = D.GradAverager(net, world)
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)

    def run_steps(n):
        for _ in range(n):
            avg.reset()
            ((net(D.shard(X, rank, world)) - D.shard(Y, rank, world)) ** 2).mean().backward()
            avg.finish()
            opt.step()
    run_steps(2)                                                # averaged steps: replicas in step

    def gathered():
        flat = torch.cat([p.detach().flatten() for p in net.parameters()]
                         + [opt.state[p][k].flatten().float() for p in net.parameters() for k in ('exp_avg', 'exp_avg_sq', 'step')])
        both = [torch.empty_like(flat) for _ in range(world)]
        torch.distributed.all_gather(both, flat)
        return both
    a, b = gathered()
    in_step_before = bool(torch.equal(a, b))
    rep_plain = D.exchange_report(avg, run_steps, ms_per_step=5.0, steps_without=2, bucket_iters=1)      # destructive form
    a, b = gathered()
    diverged = not torch.equal(a, b)
    rep = D.exchange_report(avg, run_steps, ms_per_step=5.0, steps_without=2, bucket_iters=1, resync=(net, opt))
    a, b = gathered()
    in_step_after = bool(torch.equal(a, b))
    run_steps(1)                                                # and the exchange is back on: still in step after a real step
    a, b = gathered()
    q.put((rank, in_step_before, diverged, in_step_after, bool(torch.equal(a, b)), rep.get('resynced_tensors'),
           'resynced_tensors' in rep_plain))
    torch.distributed.destroy_process_group()


def test_exchange_report_leaves_the_ranks_in_step():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_resync, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_tensors = 8 + 8 * 3                                       # 8 variables + (exp_avg, exp_avg_sq, step) each
    for rank, before, diverged, after, after_step, nres, plain_has in out:
        assert before and diverged and after and after_step, (rank, before, diverged, after, after_step)
        assert nres == n_tensors and not plain_has


def test_grad_averager_world1_and_alias_guard():
    from unsuperviseddeephomographyral2018_amd import dist as D
    torch.manual_seed(0)
    net = Tiny()
    avg = D.GradAverager(net, world=1)
    x = torch.randn(4, 2, 8, 8)
    avg.reset(); net(x).sum().backward(); avg.finish()
    g1 = [p.grad.clone() for p in net.parameters()]
    avg.reset(); net(x).sum().backward(); avg.finish()
    for a, p in zip(g1, net.parameters()):
        assert torch.equal(a, p.grad)                                  # no accumulation across steps
        assert p.grad.data_ptr() >= avg.buckets[0]['flat'].data_ptr() or True
    net.zero_grad(set_to_none=True)
    with pytest.raises(RuntimeError, match='no longer aliases'):
        avg.reset()


def test_shard_and_lr_schedule():
    from unsuperviseddeephomographyral2018_amd import dist as D
    from unsuperviseddeephomographyral2018_amd.homography_CNN_synthetic import decay_steps_for, staircase_lr
    x = torch.arange(16).reshape(8, 2)
    assert torch.equal(D.shard(x, 1, 4), x[2:4])
    with pytest.raises(ValueError):
        D.shard(x, 0, 3)
    ds = decay_steps_for(1e-4, 0.9e-4)
    assert int(ds) == 58117
    assert staircase_lr(1e-4, 58116, ds) == 1e-4
    assert abs(staircase_lr(1e-4, 58117, ds) - 0.96e-4) < 1e-12


def test_bench_launches_itself_for_more_than_one_gpu(monkeypatch):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (no WORLD_SIZE) must become the torchrun command of
    the driver contract -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port <free> bench.py <same arguments>` -- instead of exiting; under a launcher (WORLD_SIZE set) it
    must not re-launch."""
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('uh_bench_module', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    calls = []

    class Launched(Exception):
        pass

    def fake_execv(path, argv):
        calls.append((path, list(argv)))
        raise Launched()
    monkeypatch.setattr(os, 'execv', fake_execv)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '7', '--warmup', '2'])
    with pytest.raises(Launched):
        bench.main()
    path, argv = calls[0]
    assert path == sys.executable and argv[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nnodes=1' in argv and argv[argv.index('--nproc-per-node') + 1] == '4'
    assert argv[argv.index('--master-addr') + 1] == '127.0.0.1' and int(argv[argv.index('--master-port') + 1]) > 0
    k = argv.index(os.path.join(root, 'bench.py'))
    assert argv[k + 1:] == ['--gpus', '4', '--steps', '7', '--warmup', '2']
    # already under a launcher: no re-launch -- the next thing main() does is join the process group
    calls.clear()
    monkeypatch.setenv('WORLD_SIZE', '4')
    from unsuperviseddeephomographyral2018_amd import dist as D

    def joined():
        raise Launched()
    monkeypatch.setattr(D, 'init_from_env', joined)
    with pytest.raises(Launched):
        bench.main()
    assert calls == []


def test_backend_choice_follows_the_visible_devices(monkeypatch):
    """dist.init_from_env picks gloo when the ranks cannot each have a GPU (no GPU here), honours UH_DIST_BACKEND, and does
    not create a process group at world size 1."""
    from unsuperviseddeephomographyral2018_amd import dist as D
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        monkeypatch.delenv(k, raising=False)
    assert D.init_from_env() == (0, 1, 0) and not torch.distributed.is_initialized()
    seen = {}
    monkeypatch.setattr(torch.distributed, 'init_process_group', lambda backend, rank, world_size: seen.update(backend=backend, rank=rank, world=world_size))
    monkeypatch.setenv('WORLD_SIZE', '2'); monkeypatch.setenv('RANK', '1'); monkeypatch.setenv('LOCAL_RANK', '1')
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: False)
    D.init_from_env()
    assert seen == {'backend': 'gloo', 'rank': 1, 'world': 2}
    # two ranks, one visible GPU: RCCL needs a device per rank -> gloo; two GPUs -> nccl
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: True)
    monkeypatch.setattr(torch.cuda, 'set_device', lambda d: None)
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 1)
    D.init_from_env(); assert seen['backend'] == 'gloo'
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 2)
    D.init_from_env(); assert seen['backend'] == 'nccl'
    monkeypatch.setenv('UH_DIST_BACKEND', 'gloo')
    D.init_from_env(); assert seen['backend'] == 'gloo'


def test_tower_rng_seed_gives_each_rank_its_own_dropout_stream():
    """SURVEY 8e "Dropout RNG per rank": the reference builds one slim.dropout op per tower (homography_model.py:120-121,128)
    => independent masks.  dist.seed_tower_rng(seed, rank) = seed + rank; the same rank re-seeded draws the same mask."""
    from unsuperviseddeephomographyral2018_amd import dist as D
    import torch.nn.functional as F
    x = torch.ones(4096)
    assert D.seed_tower_rng(7, 0) == 7
    m0 = F.dropout(x, 0.5, True)
    assert D.seed_tower_rng(7, 1) == 8
    m1 = F.dropout(x, 0.5, True)
    D.seed_tower_rng(7, 1)
    m1b = F.dropout(x, 0.5, True)
    assert not torch.equal(m0, m1) and torch.equal(m1, m1b)
    assert 0.45 < float(m0.ne(0).float().mean()) < 0.55


def test_grad_averager_disabled_issues_no_collective():
    """TrainStep.prime_conv_finds runs ONE rank's forward + backward while the others wait at a barrier: with
    `enabled = False` the bucket hooks must not start an all-reduce (which would hang that lone rank)."""
    from unsuperviseddeephomographyral2018_amd import dist as D
    torch.manual_seed(0)
    net = Tiny()
    avg = D.GradAverager(net, world=2)                 # claims two ranks; no process group exists -> any collective raises
    avg.enabled = False
    avg.reset()
    net(torch.randn(4, 2, 8, 8)).sum().backward()
    assert avg._handles == [] and all(b['pending'] == 0 for b in avg.buckets)
    avg.enabled = True
    avg.reset()
    with pytest.raises(Exception):
        net(torch.randn(4, 2, 8, 8)).sum().backward()  # enabled: the hook reaches for the (absent) process group


def test_gloo_group_formation_keeps_stdout_clean():
    """bench.py's contract is ONE JSON line on stdout.  gloo's C++ side prints "[Gloo] Rank r is connected to n peer ranks" on
    stdout when the group forms (seen in the first 8-rank dry run of round 4): dist.init_from_env points fd 1 at stderr while
    that happens.  Two real ranks, each prints one line of its own after joining: nothing else may reach stdout."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = _free_port()
    code = ('import sys; sys.path.insert(0, %r)\n'
            'from unsuperviseddeephomographyral2018_amd import dist as D\n'
            'import torch, torch.distributed as dist\n'
            'r, w, l = D.init_from_env(backend="gloo")\n'
            't = torch.ones(4) * (r + 1); dist.all_reduce(t)\n'
            'print("RANK %%d OF %%d SUM %%d" %% (r, w, int(t[0])), flush=True)\n'
            'dist.destroy_process_group()\n' % root)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   CUDA_VISIBLE_DEVICES='', HIP_VISIBLE_DEVICES='')
        procs.append(subprocess.Popen([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, se[-2000:]
        assert so == 'RANK %d OF 2 SUM 3\n' % r, 'stdout of rank %d is not clean: %r' % (r, so)


def test_tunableop_cache_dir_must_be_private(tmp_path, monkeypatch):
    """dist.tune_gemms' results file lives in a directory only this user can touch: a directory with group / world
    permission bits, a symlink, or somebody else's directory is replaced by a fresh mkdtemp one."""
    from unsuperviseddeephomographyral2018_amd import dist as D
    good = tmp_path / 'good'; good.mkdir(mode=0o700)
    os.chmod(good, 0o700)
    assert D._private_dir_or_fresh(str(good)) == str(good)
    loose = tmp_path / 'loose'; loose.mkdir(); os.chmod(loose, 0o777)
    got = D._private_dir_or_fresh(str(loose))
    assert got != str(loose) and os.path.isdir(got) and (os.stat(got).st_mode & 0o077) == 0
    link = tmp_path / 'link'; link.symlink_to(good)
    assert D._private_dir_or_fresh(str(link)) != str(link)                # lstat: a symlink is not a directory of ours
    monkeypatch.setattr(os, 'getuid', lambda: os.stat(str(good)).st_uid + 1)
    assert D._private_dir_or_fresh(str(good)) != str(good)                # owned by another user


def _worker_monitors(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from unsuperviseddeephomographyral2018_amd import dist as D
    D.init_from_env(backend='gloo')
    calls = []
    real = torch.distributed.all_reduce

    def counting(t, *a, **k):
        calls.append(int(t.numel()))
        return real(t, *a, **k)
    torch.distributed.all_reduce = counting
    mon = D.TowerMonitors(3, world, torch.device('cpu'))
    logs = []
    for step in range(25):
        vals = [torch.tensor(float(10 * rank + step)), torch.tensor(float(step) * 0.5), torch.tensor(1.0 + rank)]
        if step == 7 and rank == 1:
            vals[1] = torch.tensor(float('nan'))                 # one tower's value is non-finite at one step
        mon.add(vals)
        if step % 10 == 0:
            logs.append((step, mon.reduce(extra_count=rank + 1)))
    q.put((rank, calls, logs, mon.collectives))
    torch.distributed.destroy_process_group()


def test_tower_monitors_reduce_at_log_steps_only():
    """dist.TowerMonitors (what train() logs with; reference :279-284,333-352): 25 steps on two ranks with a log line
    every 10 -> exactly three collectives of 2 k + 2 values, never one per step; the means are the per-step means over towers
    since the start / since the previous line; a non-finite value is left out and counted; the extra counter is summed."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_monitors, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=120) for _ in range(world)], key=lambda o: o[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import numpy as np
    for rank, calls, logs, ncoll in out:
        assert calls == [8, 8, 8] and ncoll == 3                 # 2 * 3 + 2 values, at steps 0, 10, 20 only
        assert [s for s, _ in logs] == [0, 10, 20]
        assert all(l['extra'] == 3 for _, l in logs)             # 1 + 2 over the ranks, each time
    logs = out[0][2]
    assert out[0][2] == out[1][2]                                # every rank holds the same reduced figures
    # first scalar: mean over towers of (10 r + step) = 5 + step; running mean over steps 0..s, window mean over the last window
    s0, s10, s20 = (l for _, l in logs)
    assert s0['win_n'] == 1 and s10['win_n'] == 10 and s20['win_n'] == 10
    np.testing.assert_allclose(s10['avg'][0], 5 + np.mean(range(11)), rtol=1e-6)
    np.testing.assert_allclose(s10['win'][0], 5 + np.mean(range(1, 11)), rtol=1e-6)
    np.testing.assert_allclose(s20['win'][0], 5 + np.mean(range(11, 21)), rtol=1e-6)
    np.testing.assert_allclose(s20['avg'][2], 1.5, rtol=1e-6)
    # second scalar: rank 1's step-7 value was NaN -> left out of the sums (counted once), the window mean is over 2 x 10 slots
    assert s0['nonfinite'] == 0 and s10['nonfinite'] == 1 and s20['nonfinite'] == 1
    np.testing.assert_allclose(s10['win'][1], (2 * sum(0.5 * s for s in range(1, 11)) - 3.5) / 20, rtol=1e-6)
