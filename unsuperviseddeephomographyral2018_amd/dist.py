"""Data parallelism: one process per GPU, gradient averaging over RCCL (xGMI).

Replaces the reference's in-graph tower loop + get_average_grads
(/root/reference/code/homography_CNN_synthetic.py:199-207,229-278, utils/utils.py:380-403): there the
batch is tf.split over --num_gpus towers in ONE process and per-variable tower gradients are
expand_dims -> concat -> reduce_mean on the default device.  Here every rank owns a contiguous shard
(B_local = B / world), computes the mean loss over ITS shard (so h_loss stays an RMSE per tower) and
the 34.19 M f32 gradients are averaged with two flat all-reduces:

    bucket 0 = fc2 + fc1   (134.3 MB; produced FIRST in backward  -> its all-reduce runs under the whole
                            conv backward)
    bucket 1 = conv1..8    (2.6 MB; produced last)

Gradients are persistent views into the flat bucket buffers, so there is no pack/unpack copy; the
all-reduce of a bucket is issued (async) by the post-accumulate hook of the LAST parameter of the bucket
to receive its gradient.  The hot path itself (DLT / warp / loss) has no parameters => no collective.
xGMI is point-to-point (7 links x ~153 GB/s): RCCL's direct reduce-scatter + all-gather moves 1/8 of
a bucket per link per phase (~0.11 ms for bucket 0) -- hidden under >= 3 ms of conv backward.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    Returns (rank, world, local_rank).  World size 1 => no process group at all."""
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC (the only kind this pool's driver supports) for RCCL
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if torch.cuda.is_available():
        # one process per GPU; UH_DIST_BACKEND=gloo lets several ranks share one GPU (functional test of the N > 1
        # code path on a 1-GPU box) -- RCCL itself needs one device per rank
        local = local % max(torch.cuda.device_count(), 1)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            # RCCL needs one device per rank; more ranks than visible GPUs (functional runs of the N > 1 path on a
            # 1-GPU box) fall back to gloo by themselves, as does a CPU-only process
            enough = torch.cuda.is_available() and torch.cuda.device_count() >= int(os.environ.get('LOCAL_WORLD_SIZE', world))
            backend = os.environ.get('UH_DIST_BACKEND') or ('nccl' if enough else 'gloo')
        if torch.cuda.is_available():                                      # "nccl" IS RCCL on ROCm
            torch.cuda.set_device(local)
        if backend == 'gloo':
            # gloo's C++ side prints "[Gloo] Rank r is connected to n peer ranks" on STDOUT when the group forms: a launcher
            # contract that reads ONE JSON line from stdout (bench.py) must not see it -> fd 1 points at stderr meanwhile
            import sys
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group(backend=backend, rank=rank, world_size=world)
                if dist.is_initialized():
                    dist.barrier()                               # the pairwise connections (and their message) happen here
            finally:
                os.dup2(saved, 1)
                os.close(saved)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def self_launch(nproc, target, argv, module=False):
    """A program asked for `nproc` > 1 GPUs (bench.py --gpus N; the trainer's --num_gpus N, the reference's tower count,
    homography_CNN_synthetic.py:56,199-207) with no launcher around it (no WORLD_SIZE in the environment): become
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> <target>
    <same arguments>` -- one rank per GPU; the ranks pick RCCL when every rank has a device of its own and gloo otherwise
    (init_from_env), so the same command is a functional run on a 1-GPU box.  `target` is a script path, or a module name
    with module=True (torchrun's -m).  Replaces the process: does not return."""
    import socket
    import sys
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(int(nproc)),
           '--master-addr', '127.0.0.1', '--master-port', str(port)]
    cmd += (['-m', target] if module else [os.path.abspath(target)]) + list(argv)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    sys.stdout.flush(); sys.stderr.flush()
    os.execv(sys.executable, cmd)


def resolve_num_gpus(num_gpus, batch_size=None):
    """What the reference's --num_gpus (number of towers the batch is tf.split over, homography_CNN_synthetic.py:56,199-207)
    means with one process per GPU.  -> ('run', world) to go on in this process, or ('launch', N) when the caller must
    self_launch N ranks.  Never silently ignored:
      not given (None)            the launcher decides: WORLD_SIZE, or 1 without a launcher
      N, no launcher, N == 1      one rank, this process
      N, no launcher, N > 1       ('launch', N)
      N, under a launcher         must equal WORLD_SIZE, otherwise ValueError
    and the batch must split evenly over the towers, as tf.split demands."""
    env_world = os.environ.get('WORLD_SIZE')
    if num_gpus is not None and int(num_gpus) < 1:
        raise ValueError('--num_gpus must be >= 1, got %s' % num_gpus)
    if env_world is not None:
        world = int(env_world)
        if num_gpus is not None and int(num_gpus) != world:
            raise ValueError('--num_gpus %d contradicts the launcher: WORLD_SIZE=%d (one process per GPU; start %d ranks, or '
                             'drop --num_gpus)' % (int(num_gpus), world, int(num_gpus)))
    else:
        world = 1 if num_gpus is None else int(num_gpus)
    if batch_size is not None and int(batch_size) % world:
        raise ValueError('--batch_size %d does not split over %d towers (tf.split would fail the same way)' % (int(batch_size), world))
    if env_world is None and world > 1:
        return 'launch', world
    return 'run', world


def skip_naive_conv_in_find():
    """MIOpen's find mode (torch.backends.cudnn.benchmark) also BENCHMARKS its reference solvers -- naive_conv_*: one thread per
    output, 0.3 s per launch for this network's weight gradients; 48 + 48 + 40 launches = 16.7 of the ~40 s the first step of a
    rank costs (profiles/r03_bench_kernel_stats.csv) -- which never win.  The documented MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_*
    switches take them out of the candidate list; MIOpen reads them when it first looks for solvers, so this must run before
    the first convolution of the process.  An explicit setting in the environment wins."""
    for k in ('FWD', 'BWD', 'WRW'):
        os.environ.setdefault('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_' + k, '0')


def tune_gemms(max_ms_per_solution=15, max_iterations=20, filename=None):
    """PyTorch's TunableOp for the regressor's fully connected GEMMs -- what cudnn.benchmark / MIOpen find is for its convs.
    fc1 is a skinny GEMM (batch x 32768 x 1024: 134 MB of weights for 64 rows); the library heuristic's pick runs the forward
    in 142 us where the weights could stream in ~25 (profiles/r04_step_breakdown.txt).  With TunableOp on, the first call of
    each GEMM shape benchmarks the rocBLAS / hipBLASLt candidates and keeps the fastest; results are cached in `filename`
    (default: <per-user cache dir>/tunableop[_r<local rank>_]<device>.csv -- ~/.cache/uh_hotpath, or $TMPDIR/uh_hotpath_uid<uid>
    when the home directory is not writable; never a world-shared path another user could have planted -- validated against the
    library versions), so later processes of the same user on the same host skip the search.  Returns True when TunableOp was
    switched on.
    Reproducibility: the pick is timing-dependent, so two runs (or two ranks) may choose different GEMM kernels for the same shape
    -- results then differ in the last bits (reduction order), like MIOpen's find mode does for the convs; test mode does not
    tune (it uses the library heuristic).  --tunable_gemm False gives the library heuristic everywhere."""
    try:
        import torch.cuda.tunable as tun
        # one results file per (local rank, device): ranks that share a GPU (gloo dry runs) must not write the same file at exit
        lr = os.environ.get('LOCAL_RANK')
        if filename is None:
            d = os.path.join(os.environ.get('XDG_CACHE_HOME') or os.path.join(os.path.expanduser('~'), '.cache'), 'uh_hotpath')
            try:
                os.makedirs(d, mode=0o700, exist_ok=True)
                if not os.access(d, os.W_OK):
                    raise OSError(d)
            except OSError:
                d = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'uh_hotpath_uid%d' % os.getuid())
                os.makedirs(d, mode=0o700, exist_ok=True)
            d = _private_dir_or_fresh(d)
            filename = os.path.join(d, 'tunableop%s.csv' % ('_r' + lr + '_' if lr is not None else ''))
        tun.set_filename(filename, insert_device_ordinal=True)
        tun.set_max_tuning_duration(int(max_ms_per_solution))
        tun.set_max_tuning_iterations(int(max_iterations))
        tun.enable(True)
        tun.tuning_enable(True)
        return True
    except Exception:                                        # noqa: BLE001 -- an optimisation, never a requirement
        return False


def _private_dir_or_fresh(d):
    """`d` if it is a real directory owned by this user with no group / world permission bits (0o700); otherwise a fresh mkdtemp directory (no
    caching across processes, but never a results file another user could have planted: makedirs(exist_ok=True) accepts a
    directory somebody else created first in a world-writable $TMPDIR)."""
    import stat
    import tempfile
    try:
        st = os.lstat(d)
        if stat.S_ISDIR(st.st_mode) and st.st_uid == os.getuid() and (st.st_mode & 0o077) == 0:
            return d
    except OSError:
        pass
    return tempfile.mkdtemp(prefix='uh_hotpath_')


def seed_tower_rng(seed, rank):
    """Every tower draws its OWN dropout masks: the reference builds one slim.dropout op per tower
    (homography_model.py:120-121,128, one model per tower at homography_CNN_synthetic.py:229-233), so the masks of two
    towers are independent.  One process per GPU => seed the model RNG (host + every device generator) with seed + rank.
    Call it AFTER the variables were created / broadcast, so that the initial variables stay rank-equal."""
    torch.manual_seed(int(seed) + int(rank))
    return int(seed) + int(rank)


def shard(tensor, rank, world):
    """Contiguous batch shard of this rank -- tf.split(x, num_gpus, 0)[rank]."""
    B = tensor.shape[0]
    if B % world:
        raise ValueError('batch %d not divisible by world size %d' % (B, world))
    n = B // world
    return tensor[rank * n:(rank + 1) * n]


class GradAverager(object):
    """Bucketed, overlapped gradient mean (== utils.get_average_grads across towers)."""

    def __init__(self, module, world=None, first_bucket=('fc2', 'fc1')):
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        params = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        b0 = [(n, p) for n, p in params if n.split('.')[0] in first_bucket]
        b1 = [(n, p) for n, p in params if n.split('.')[0] not in first_bucket]
        self.buckets = []
        self._handles = []
        self._hooks = []
        # RCCL ("nccl") averages in the collective itself (ReduceOp.AVG): no separate 137 MB divide pass after the wait.
        # gloo (CPU tests) has no AVG: SUM + div_.
        self._avg_in_collective = bool(self.world > 1 and dist.is_initialized() and dist.get_backend() == 'nccl')
        self._op = dist.ReduceOp.AVG if self._avg_in_collective else dist.ReduceOp.SUM
        self.enabled = True                # False: the hooks issue no collective (TrainStep.prime_conv_finds: one rank alone)
        self.exposed_wait_s = 0.0          # host time finish() spent blocked on the collectives (what backward did not hide)
        self.finishes = 0
        for plist in (b0, b1):
            if not plist:
                continue
            total = sum(p.numel() for _, p in plist)
            flat = torch.zeros(total, dtype=plist[0][1].dtype, device=plist[0][1].device)
            off = 0
            for _, p in plist:
                # persistent view with the PARAMETER's strides (conv weights are channels_last): no
                # pack/unpack, and fused Adam requires grad/param layouts to match
                dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
                if not dense:
                    raise RuntimeError('GradAverager needs dense (contiguous or channels_last) parameters')
                p.grad = torch.as_strided(flat, p.size(), p.stride(), off)
                off += p.numel()
            bucket = {'flat': flat, 'params': [p for _, p in plist], 'pending': 0}
            self.buckets.append(bucket)
            for _, p in plist:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bucket)))
        self.reset()

    def _make_hook(self, bucket):
        def hook(param):
            bucket['pending'] -= 1
            if bucket['pending'] == 0 and self.world > 1 and self.enabled:
                self._handles.append(dist.all_reduce(bucket['flat'], op=self._op, async_op=True))
        return hook

    def reset(self):
        """Call before each backward: zero the flat buffers (in place, views stay valid)."""
        self._handles = []
        for b in self.buckets:
            b['flat'].zero_()
            b['pending'] = len(b['params'])
            for p in b['params']:            # an optimizer may have detached .grad (set_to_none)
                if p.grad is None or p.grad.data_ptr() < b['flat'].data_ptr() or \
                        p.grad.data_ptr() >= b['flat'].data_ptr() + b['flat'].numel() * b['flat'].element_size():
                    raise RuntimeError('GradAverager: a .grad no longer aliases its bucket; use '
                                       'optimizer.zero_grad(set_to_none=False) or GradAverager.reset() only')

    def finish(self):
        """Call after backward, before optimizer.step(): wait for the all-reduces and divide by world.  With `enabled` False
        (no collective was issued) it does nothing: every rank keeps its own tower's gradient."""
        if self.world > 1 and self.enabled:
            import time
            for b in self.buckets:
                if b['pending'] != 0:          # a parameter received no gradient: reduce it now
                    self._handles.append(dist.all_reduce(b['flat'], op=self._op, async_op=True))
            t0 = time.perf_counter()
            for h in self._handles:
                h.wait()
            self.exposed_wait_s += time.perf_counter() - t0
            self.finishes += 1
            if not self._avg_in_collective:
                for b in self.buckets:
                    b['flat'].div_(self.world)
        self._handles = []

    def time_buckets(self, iters=5):
        """Stand-alone duration of each bucket's all-reduce (barrier + device sync either side, `iters` repeats): what the
        exchange step costs when nothing hides it.  -> [{'bytes', 'ms', 'algbw_GBs', 'busbw_GBs'}] (rank-local view)."""
        import time
        out = []
        for b in self.buckets:
            buf = torch.empty_like(b['flat'])
            dist.all_reduce(buf, op=self._op)                      # warm-up (communicator / buffers)
            if buf.is_cuda:
                torch.cuda.synchronize(buf.device)
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(iters):
                dist.all_reduce(buf, op=self._op)
            if buf.is_cuda:
                torch.cuda.synchronize(buf.device)
            dt = (time.perf_counter() - t0) / iters
            nbytes = buf.numel() * buf.element_size()
            out.append({'bytes': nbytes, 'ms': round(dt * 1e3, 4), 'algbw_GBs': round(nbytes / dt / 1e9, 1),
                        'busbw_GBs': round(nbytes / dt / 1e9 * 2 * (self.world - 1) / self.world, 1)})
        return out

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


XGMI_LINK_GBS = 153.0        # per direction per link, 7 links per GPU on an 8-GPU MI355X node


def exchange_report(averager, run_steps, ms_per_step=None, steps_without=10, bucket_iters=5, resync=None):
    """What the gradient exchange costs, readable from ONE line of the first real multi-GPU run.  Collective:
    every rank must call it, after the timed region.
      buckets                    stand-alone all-reduce time per bucket (nothing to hide under), algbw / busbw, and busbw as a
                                 fraction of the xGMI bandwidth a rank can use towards its world - 1 peers (links x 153 GB/s)
      ms_per_step_no_exchange    `steps_without` more steps with the averager DISABLED on every rank (each tower keeps its own
                                 gradient; barrier + device sync either side, MAX over ranks) -- the step without the exchange
      ms_per_step_with_exchange  only when ms_per_step is None: `steps_without` steps with the exchange ON, timed the same way
      exchange_cost_ms_per_step  ms_per_step - ms_per_step_no_exchange: the un-hidden part of the exchange PLUS what the RCCL
                                 kernels take from the conv backward they overlap (CU / HBM contention) -- the number
                                 host_blocked_ms_per_step cannot show on RCCL, whose wait() only enqueues a stream dependency
    run_steps(n) runs n training steps of the caller's TrainStep.
    The no-exchange steps are REAL steps (optimizer included) on per-tower gradients, so the replicas diverge: pass
    resync=(module, optimizer) and rank 0's variables, buffers and optimizer state are broadcast afterwards (broadcast_state) --
    ranks leave the call in step.  Without it the call is destructive and belongs at the very end of a run.  An exception on
    ONE rank inside a collective leaves the others waiting: call it where a hang costs nothing but the diagnostic."""
    import time
    world = averager.world
    out = {'reduce_op': 'AVG' if averager._avg_in_collective else 'SUM+div',
           'host_blocked_ms_per_step': round(averager.exposed_wait_s / max(averager.finishes, 1) * 1e3, 4)}
    try:
        buckets = averager.time_buckets(iters=bucket_iters)
        peak = max(world - 1, 1) * XGMI_LINK_GBS
        for b in buckets:
            b['busbw_frac_of_xgmi'] = round(b['busbw_GBs'] / peak, 4)
        out['buckets'] = buckets
        out['xgmi'] = {'links_to_peers': max(world - 1, 1), 'GBs_per_link': XGMI_LINK_GBS, 'peak_GBs': peak,
                       'note': 'busbw = algbw * 2 (n - 1) / n; point-to-point xGMI: a rank reaches each of its n - 1 peers over one '
                               'link, so a direct all-reduce is bound by (n - 1) x 153 GB/s per direction'}
    except Exception as e:                                  # noqa: BLE001 -- a diagnostic: must not lose the headline
        out['buckets'] = {'error': '%s: %s' % (type(e).__name__, e)}
    dev = averager.buckets[0]['flat'].device if averager.buckets else torch.device('cpu')
    sync = (lambda: torch.cuda.synchronize(dev)) if dev.type == 'cuda' else (lambda: None)

    def timed(n):                                           # barrier + device sync either side, MAX over ranks -> ms per step
        sync(); dist.barrier(); sync()
        t0 = time.perf_counter()
        run_steps(n)
        sync(); dist.barrier(); sync()
        tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item()) / n * 1e3
    try:
        if ms_per_step is None:                             # the caller has no like-for-like figure (a trainer whose loop also
            run_steps(2)                                    # draws inputs and logs): time the step WITH the exchange the same way
            ms_per_step = timed(steps_without)
            out['ms_per_step_with_exchange'] = round(ms_per_step, 3)
        averager.enabled = False
        run_steps(2)                                        # settle (the allocator sees a step without the collectives)
        ms_wo = timed(steps_without)
        out['ms_per_step_no_exchange'] = round(ms_wo, 3)
        out['exchange_cost_ms_per_step'] = round(ms_per_step - ms_wo, 3)
        out['steps_without_exchange'] = steps_without
    except Exception as e:                                  # noqa: BLE001
        out['ms_per_step_no_exchange'] = {'error': '%s: %s' % (type(e).__name__, e)}
    finally:
        averager.enabled = True
        # the steps above applied each tower's OWN gradient: variables (and optimizer moments) have diverged across ranks.
        # Put every rank back on rank 0's state so that a caller that goes on training keeps identical replicas.
        if resync is not None:
            try:
                out['resynced_tensors'] = broadcast_state(resync)
            except Exception as e:                              # noqa: BLE001
                out['resynced_tensors'] = {'error': '%s: %s' % (type(e).__name__, e)}
    out['note'] = ('bucket 0 = fc2+fc1 (issued first in backward, overlaps the conv backward), bucket 1 = conv; RCCL wait() only '
                   'enqueues a stream dependency, so host_blocked is ~0 by construction: exchange_cost_ms_per_step is the figure '
                   'that shows what the exchange really costs (un-hidden time + contention with the conv backward)')
    return out


def broadcast_state(module_and_optimizer, src=0):
    """Broadcast rank `src`'s variables, buffers and optimizer state tensors (Adam moments, step counts) to every rank, in a
    fixed order.  -> number of tensors sent.  Collective."""
    module, opt = module_and_optimizer
    n = 0
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)
        n += 1
    if opt is not None:
        for group in opt.param_groups:
            for p in group['params']:
                st = opt.state.get(p, {})
                for k in sorted(st):
                    v = st[k]
                    if torch.is_tensor(v):
                        if v.device.type == 'cpu' and dist.get_backend() == 'nccl':      # a host-side step counter: via the device
                            tmp = v.to(p.device)
                            dist.broadcast(tmp, src=src)
                            v.copy_(tmp.cpu())
                        else:
                            dist.broadcast(v, src=src)
                        n += 1
    return n


class TowerMonitors(object):
    """Running means of a tower's logging scalars with ONE collective per log line (reference: total_*_loss = reduce_mean over
    towers, homography_CNN_synthetic.py:279-284, accumulated and printed every 100 steps, :333-352; SURVEY 8e "loss scalars
    all-reduced only for logging").  Every rank adds its tower's values to device-side sums each step -- no collective, no host
    sync; at a log step the sums of all ranks meet in one all-reduce of 2 k + 2 doubles.  Means are linear, so the printed
    figures equal the per-step mean over towers.  A non-finite value (a degenerate pair makes that step's loss NaN; its gradient
    is zeroed in the DLT backward) is kept out of the sums and counted."""

    def __init__(self, k, world, device):
        self.k, self.world = int(k), int(world)
        self.totals = torch.zeros(self.k, device=device)
        self.window = torch.zeros(self.k, device=device)
        self.nonfinite = torch.zeros((), device=device)
        self.steps = 0
        self.win_n = 0
        self.collectives = 0

    def add(self, values):
        cur = torch.stack([v.detach().float().reshape(()) for v in values])
        finite = torch.isfinite(cur)
        self.nonfinite += (~finite.all()).float()
        cur = torch.where(finite, cur, torch.zeros_like(cur))
        self.totals += cur
        self.window += cur
        self.steps += 1
        self.win_n += 1

    def reduce(self, extra_count=0):
        """Collective (every rank, same step).  -> dict(avg = means since the start, win = means since the previous call, win_n,
        nonfinite = tower-steps with a non-finite value so far, extra = sum over ranks of `extra_count`); starts a new window."""
        mon = torch.cat([self.totals, self.window, self.nonfinite.reshape(1),
                         torch.tensor([float(extra_count)], device=self.totals.device)]).double()
        if self.world > 1:
            dist.all_reduce(mon)
            self.collectives += 1
        k = self.k
        out = {'avg': (mon[:k] / (self.world * max(self.steps, 1))).tolist(),
               'win': (mon[k:2 * k] / (self.world * max(self.win_n, 1))).tolist(), 'win_n': self.win_n,
               'nonfinite': int(round(float(mon[2 * k]))), 'extra': int(round(float(mon[2 * k + 1])))}
        self.window.zero_()
        self.win_n = 0
        return out


def all_reduce_mean_scalars(values, world):
    """Mean over ranks of a few logging scalars (total_*_loss = reduce_mean over towers, :279-284)."""
    if world == 1:
        return values
    t = torch.stack([v.detach().float().reshape(()) for v in values])
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    t = t / world
    return [t[i] for i in range(len(values))]
