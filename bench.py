#!/usr/bin/env python
"""bench.py -- image-pairs/s of the unsupervised-homography TRAIN STEP on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1 without a launcher: re-executes itself under
                                                                  torch.distributed.run, one rank per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the whole hot loop over one batch of synthetic pairs already resident in HBM
(reference homography_CNN_synthetic.py:333-353): VGG forward (stock PyTorch-ROCm, f32) -> Tensor-DLT ->
full-frame projective warp -> gray/patch gather -> photometric L1 -> full backward (HIP warp/DLT
backward kernels + conv backward) -> RCCL gradient mean (N > 1) -> Adam.  Workload at N=1 =
BASELINE.json configs[1]: batch 64, 240x320 I/I', 128x128 patch, RHO=45, photometric L1 (reference flag
--loss_type l1_loss).  N > 1: weak scaling, 64 pairs per GPU (configs[2]: 8 x 64 = 512).

Rank 0 prints ONE JSON line; extra objects:
  config4_point / north_star_point -- the warp kernels alone (dense dOut) at BASELINE configs[3] (480x640, rho 64,
                  batch 128: HBM-resident) and at north_star's batch-128 240x320 point, with the path mix of the tiles.
  roofline     -- dominant hot-path kernel (full-frame warp, forward or backward, whichever took more
                  time): algorithmic bytes per launch / average launch duration measured with HIP
                  events on the launch stream (uh_profile_*) inside the timed region, vs 8 TB/s HBM.
  cpu_baseline -- the reference-equivalent op graph on torch-CPU (oracle/hotpath_torch.py) timed on this
                  box's host cores on a bounded sample of the same workload: the GPU leg's batch and corner offsets,
                  copied to the host (rank 0, N=1 only).
  exchange     -- N > 1 only (dist.exchange_report): each bucket's stand-alone all-reduce time with busbw against (n - 1) x 153 GB/s
                  of xGMI, ms_per_step_no_exchange (10 more steps with the averager disabled on every rank) and
                  exchange_cost_ms_per_step = ms_per_step - that: what the gradient exchange really costs, contention included;
                  the ranks are put back in step afterwards (resynced_tensors).
  headline_roofline -- the whole step against the f32 MFMA peak (the conv stack is stock MIOpen); hot_path_share_of_step = the
                  library kernels timed in this run (avg_us x launches / steps) over ms_per_step -- computed, never quoted.
  quality      -- the metric's second half, "mean corner error": a from-scratch unsupervised l1_loss training with the
                  reference's hyper-parameters for a fixed step budget OUTSIDE the timed region, then the reference's
                  test statistics (homography_CNN_synthetic.py:391-401,573-579) on held-out pairs (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12           # B/s, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--per_gpu_batch', type=int, default=64)
    ap.add_argument('--img_h', type=int, default=240)
    ap.add_argument('--img_w', type=int, default=320)
    ap.add_argument('--patch_size', type=int, default=128)
    ap.add_argument('--rho', type=int, default=45)
    ap.add_argument('--loss_type', default='l1_loss')
    ap.add_argument('--fused_patch', type=int, default=0, help='1: fused patch kernel instead of the full-frame warp')
    ap.add_argument('--tunable_gemm', type=int, default=1,
                    help='1 (the trainer default): PyTorch TunableOp for the fully connected GEMMs (dist.tune_gemms): the first call of '
                         'each GEMM shape benchmarks the rocBLAS / hipBLASLt candidates, like MIOpen find does for the convs; 0: off')
    ap.add_argument('--graph_tail', type=int, default=0, help='1: DLT->warp->loss and backward as one hipGraph launch')
    ap.add_argument('--step_graph', type=int, default=0, help='1: whole training step as one hipGraph replay (implies --profile 0)')
    ap.add_argument('--profile', type=int, default=1, help='0: no per-kernel events in the timed region (no roofline object); 1: time the warp kernels; 2: time every library kernel')
    ap.add_argument('--cpu_baseline', type=int, default=1)
    ap.add_argument('--north_star', type=int, default=1, help='0: skip the extra warp-only measurement at batch 128')
    ap.add_argument('--config4', type=int, default=1, help='0: skip the extra warp-only measurement at BASELINE configs[3] (480x640, rho 64, batch 128)')
    ap.add_argument('--only_points', default='', help='"config4", "north_star" or "config4,north_star": measure only those '
                    'warp-only points and print them as one JSON line (used under rocprofv3 so that the kernel-trace stats '
                    'hold one shape per kernel name)')
    ap.add_argument('--mid_training_theta', type=int, default=1,
                    help='1: add gt + N(0, 2 px) to the regressor output inside the timed steps, so that theta follows SURVEY '
                         '8(d)\'s mid-training law (perspective, clipped and far-field tiles) instead of the near-identity a '
                         '25-step-old regressor predicts; 0: the raw regressor')
    ap.add_argument('--traffic', type=int, default=1,
                    help='1 (N = 1 only): measure roofline.traffic IN THIS RUN -- two child runs of this script under `rocprofv3 '
                         '--pmc FETCH_SIZE|WRITE_SIZE --kernel-trace` (separate passes) after the headline; 0: read the committed file')
    ap.add_argument('--traffic_child', type=int, default=0, help='(internal) this process is one of those child runs')
    ap.add_argument('--quality', type=int, default=1, help='0: skip the mean-corner-error training run')
    ap.add_argument('--quality_steps', type=int, default=8000, help='training steps of the quality run')
    ap.add_argument('--quality_pool', type=int, default=384, help='in-HBM pool of pre-generated training batches (118 MB each)')
    ap.add_argument('--quality_texture', default='multiscale', choices=['smooth', 'multiscale'])
    ap.add_argument('--cpu_sample_pairs', type=int, default=64, help='CPU leg: batch size (default = the GPU batch)')
    ap.add_argument('--cpu_sample_steps', type=int, default=12, help='CPU leg: timed steps after one warm-up step')
    ap.add_argument('--cpu_threads', type=int, default=32, help='host threads for the CPU leg (capped at nproc); 32 is the best of the '
                    'sweep in profiles/r05_cpu_threads.txt')
    ap.add_argument('--cpu_threads_sweep', default='', help='e.g. "8,32,128,256": time ONLY the CPU leg once per thread count on the '
                    'default workload, print one JSON line per count, and exit (profiles/r05_cpu_threads.txt)')
    return ap.parse_args()


def cpu_baseline(args, batch, h4p_offset):
    """Reference-equivalent op graph on torch-CPU: full train step (same VGG, TF-graph-shaped hot path with autograd
    backward, Adam) on the GPU leg's OWN workload: its batch (seed 100) and its per-pair corner offsets (the mid-training
    theta law) are copied to the host, the first `cpu_sample_pairs` pairs are stepped `cpu_sample_steps` times."""
    from oracle import hotpath_torch as OT
    from unsuperviseddeephomographyral2018_amd.homography_model import VGGRegressor
    cores = min(os.cpu_count() or 1, args.cpu_threads)     # small ops: more threads only add contention
    torch.set_num_threads(cores)
    B, H, W, P = min(args.cpu_sample_pairs, batch['I_aug'].shape[0]), args.img_h, args.img_w, args.patch_size
    cb = {k: v[:B].detach().cpu().contiguous() for k, v in batch.items() if torch.is_tensor(v)}
    off = h4p_offset[:B].detach().cpu() if h4p_offset is not None else None
    torch.manual_seed(1234)
    net = VGGRegressor(P).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    OT.train_step_cpu(net, opt, cb, W, H, P, args.loss_type, h4p_offset=off)   # warm-up (allocator, threads)
    t0 = time.perf_counter()
    for _ in range(args.cpu_sample_steps):
        OT.train_step_cpu(net, opt, cb, W, H, P, args.loss_type, h4p_offset=off)
    dt = time.perf_counter() - t0
    return {'value': round(B * args.cpu_sample_steps / dt, 2), 'unit': 'image-pairs/s', 'cores': cores,
            'kind': 'port',
            'sample': '%d train steps of the GPU leg\'s own batch (synthetic.make_batch seed 100, first %d pairs, %dx%d, P=%d) '
                      'under the same theta law (%s), copied to the host: torch-CPU op-graph restatement of the TF graph + '
                      'same VGG + Adam, %d threads of %d host cores, %.1f s' % (
                          args.cpu_sample_steps, B, H, W, P,
                          'regressor + gt + N(0,2px), generator seed 4321' if off is not None else 'raw regressor',
                          cores, os.cpu_count() or 1, dt)}


def quality_run(device, args):
    """Mean corner error (BASELINE.json's metric, second half): unsupervised photometric l1_loss from scratch with the
    reference's hyper-parameters (Adam, lr 1e-4, staircase decay with decay_steps_for(), batch 64, 240x320, P=128,
    RHO=45; homography_CNN_synthetic.py:161-183) for a FIXED budget of steps, then the reference's test loop
    (TestHomography, :391-401,573-579: 3 passes over `num_test_data` held-out pairs, bounded RMSE, failure rate).
    Training pairs cycle through an in-HBM pool drawn with the law of gen_synthetic_data.py:42-64; the test pairs come
    from disjoint seeds.  Runs outside the timed region, on its own freshly initialised variables."""
    import contextlib
    from unsuperviseddeephomographyral2018_amd import synthetic
    from unsuperviseddeephomographyral2018_amd.homography_CNN_synthetic import TestHomography, TrainStep, build_parser
    B = args.per_gpu_batch
    targs = build_parser().parse_args([
        '--mode', 'train', '--loss_type', 'l1_loss', '--batch_size', str(B), '--img_h', str(args.img_h),
        '--img_w', str(args.img_w), '--patch_size', str(args.patch_size), '--rho', str(args.rho), '--lr', '1e-4',
        '--min_lr', '.9e-4', '--texture', args.quality_texture, '--num_test_data', '1024', '--seed', '0'])
    torch.manual_seed(0)
    t0 = time.perf_counter()
    step_fn = TrainStep(targs, device, 1)
    pool = [synthetic.make_batch(B, args.img_h, args.img_w, args.patch_size, args.rho, seed=1000 + i, device=device,
                                 kind=args.quality_texture) for i in range(max(1, args.quality_pool))]
    gen = torch.Generator().manual_seed(7919)
    order, first, last = [], [], []
    for s in range(args.quality_steps):
        if not order:
            order = torch.randperm(len(pool), generator=gen).tolist()
        m = step_fn(pool[order.pop()])
        if s < 100:
            first.append(m.h_loss.detach())
        if s >= args.quality_steps - 100:
            last.append(m.h_loss.detach())
    torch.cuda.synchronize(device)
    t_train = time.perf_counter() - t0
    with contextlib.redirect_stdout(sys.stderr):           # the test loop prints the reference's result lines
        res = TestHomography(targs, step_fn=step_fn).run()
    # what an un-trained predictor scores on the same held-out law: pred_h4p = 0 -> RMSE of gt itself
    ident = []
    for step in range(8):
        b = synthetic.make_batch(B, args.img_h, args.img_w, args.patch_size, args.rho, seed=10_000_000 + step, device=device,
                                 kind=args.quality_texture)
        ident.append(torch.sqrt(torch.mean(b['gt'] ** 2, dim=1)).mean())
    del pool
    torch.cuda.empty_cache()
    return {'mean_corner_error_px': round(res['mean_corner_error'], 3), 'fail_percent': round(res['fail_percent'], 3),
            'identity_corner_error_px': round(float(torch.stack(ident).mean()), 3),
            'train_steps': args.quality_steps, 'train_seconds': round(t_train, 1), 'test_pairs': res['num_pairs'],
            'train_corner_error_px_first_100_steps': round(float(torch.stack(first).mean()), 3) if first else None,
            'train_corner_error_px_last_100_steps': round(float(torch.stack(last).mean()), 3) if last else None,
            'percentiles_px': {str(k): round(v, 3) for k, v in res['percentiles'].items()},
            'hyper_parameters': 'reference defaults: Adam lr 1e-4, exponential_decay 0.96 every %d steps (staircase), batch %d, '
                                'dropout 0.5, loss_type l1_loss (photometric)' % (int(step_fn.decay_steps), B),
            'data': 'synthetic %s texture, pool of %d batches cycled in random order; test: %d held-out pairs x 3 passes '
                    '(reference: MS-COCO crops, 150 000 steps)' % (args.quality_texture, args.quality_pool, 1024),
            'note': 'fixed step budget (%d of the reference\'s 150 000 steps): not a converged model; longer runs under profiles/'
                    % args.quality_steps,
            'reference_schedule_run': reference_schedule_result()}


def regressor_flops(P):
    """Forward FLOPs per pair of the VGG regressor (homography_model.py:107-133) on a P x P patch: 3x3 convs 2->64->64 at P,
    64->64->64 at P/2, 64->128->128 at P/4, 128->128->128 at P/8, fc (P/8)^2*128 -> 1024 -> 8.  2.52 GFLOP at P = 128."""
    f, side = 0, P
    for chans in ([(2, 64), (64, 64)], [(64, 64), (64, 64)], [(64, 128), (128, 128)], [(128, 128), (128, 128)]):
        for ci, co in chans:
            f += 2 * side * side * ci * co * 9
        side //= 2
    side *= 2                                                 # no pool after the last block
    return f + 2 * side * side * 128 * 1024 + 2 * 1024 * 8


def reference_schedule_result():
    """The committed results of the runs at the reference's own schedule (150 000 steps, lr 1e-4) -- read from profiles/, NOT
    measured in this run (14 GPU-minutes each): the clean in-HBM pool (tools/train_reference_schedule.sh) and, since round 5, the
    reference's real input route -- JPEG files + joint augmentation through Dataloader / uh_prepare_inputs -- tested with and
    without the reference's disjoint test augmentation (tools/train_from_disk.py)."""
    import re
    out = None
    f = os.path.join(ROOT, 'profiles', 'r06_train_reference_schedule.txt')
    try:
        txt = open(f).read()
        m = re.search(r'Average error: ([0-9.]+) \|Fail percent: ([0-9.]+)', txt)
        out = {'mean_corner_error_px': float(m.group(1)), 'fail_percent': float(m.group(2)), 'train_steps': 150000,
               'file': 'profiles/r06_train_reference_schedule.txt', 'note': 'read from the committed log, not measured in this run',
               'data': 'clean in-HBM pool of 65 536 synthetic pairs, augment_list = [normalize]'}
    except Exception:
        return None
    f2 = os.path.join(ROOT, 'profiles', 'r06_train_from_disk_reference_schedule.txt')
    try:
        rows = [json.loads(l[len('RESULT '):]) for l in open(f2) if l.startswith('RESULT ')]
        out['from_jpeg_files_with_augmentation'] = {
            'file': 'profiles/r06_train_from_disk_reference_schedule.txt', 'note': 'read from the committed log, not measured in this run',
            'results': [{k: r[k] for k in ('files', 'train_do_augment', 'test_do_augment', 'steps', 'train_pairs', 'mean_corner_error_px',
                                           'fail_percent', 'per_pair_median_px')} for r in rows]}
    except Exception:
        pass
    return out


def tile_paths(theta, H, W, lds_bytes=5120, tile=16):
    """Which path each 16x16 wave tile of uh_warp_forward takes under `theta` [B,9] (csrc/uh_warp.hip): A = interior,
    tap rectangle fits the wave's LDS slice (LDS-DMA staged); B = interior, rectangle too large (gather); C1 / C2 = some
    tap clipped, rectangle of the clipped taps fits / does not fit.  Re-derived here with torch ops in f32 (statistics
    only: a tile on a rounding edge may be classified differently from the kernel)."""
    B = theta.shape[0]
    th = theta.reshape(B, 3, 3).float()
    dev = theta.device
    gx = -1.0 + (2.0 / (W - 1)) * torch.arange(W, device=dev, dtype=torch.float32)
    gy = -1.0 + (2.0 / (H - 1)) * torch.arange(H, device=dev, dtype=torch.float32)
    GX, GY = gx[None, None, :], gy[None, :, None]
    row = lambda r: th[:, r, 0, None, None] * GX + th[:, r, 1, None, None] * GY + th[:, r, 2, None, None]
    t = row(2)
    t = torch.where(t.abs() >= 1e-7, t, t + 1e-6)
    x = (row(0) / t + 1.0) * W * 0.5
    y = (row(1) / t + 1.0) * H * 0.5
    fx, fy = torch.floor(x), torch.floor(y)
    big = 2147483648.0
    fx = torch.where((fx < big) & (fx >= -big), fx, torch.full_like(fx, -1.0)); fy = torch.where((fy < big) & (fy >= -big), fy, torch.full_like(fy, -1.0))
    Ht, Wt = H // tile, W // tile
    T = lambda a: a[:, :Ht * tile, :Wt * tile].reshape(B, Ht, tile, Wt, tile)
    mn = lambda a: T(a).amin(dim=(2, 4)); mx = lambda a: T(a).amax(dim=(2, 4))
    interior = (mn(fx) >= 0) & (mx(fx) + 1 <= W - 1) & (mn(fy) >= 0) & (mx(fy) + 1 <= H - 1)

    def fits(x0, x1, y0, y1):
        rw, rh = x1 - x0 + 1, y1 - y0 + 1
        cpr = torch.ceil(rw * 12 / 16)
        return (cpr <= 64) & (cpr * rh * 16 <= lds_bytes)
    fitA = fits(mn(fx), mx(fx) + 1, mn(fy), mx(fy) + 1)
    cx0, cx1 = fx.clamp(0, W - 1), (fx + 1).clamp(0, W - 1)
    cy0, cy1 = fy.clamp(0, H - 1), (fy + 1).clamp(0, H - 1)
    fitC = fits(mn(cx0), mx(cx1), mn(cy0), mx(cy1))
    n = float(interior.numel())
    return {'A_staged_interior': round(float((interior & fitA).sum()) / n, 4),
            'B_gather_interior': round(float((interior & ~fitA).sum()) / n, 4),
            'C1_staged_clipped': round(float((~interior & fitC).sum()) / n, 4),
            'C2_gather_clipped': round(float((~interior & ~fitC).sum()) / n, 4)}


def warp_point(device, B, H, W, P, rho, iters=30, seed=7, warm=30):
    """Warp forward + backward alone (dense dOut) under the mid-training law theta = DLT(gt + N(0, 2 px)): HIP-event
    kernel durations via uh_profile_*, algorithmic bytes 4*B*H*W*C*4, fraction of the 8 TB/s roofline.
    `warm` untimed iterations first: the first ~10 launches on freshly allocated gigabyte tensors run ~25 % slower than
    the steady state (measured in round 3: 3 warm-up + 20 timed launches average 187 / 200 us where the same process
    settles at 170 / 185 us; rounds 1-2 reported the former)."""
    from unsuperviseddeephomographyral2018_amd import _lib, ops, synthetic
    b = synthetic.make_batch(B, H, W, P, rho, seed=seed, device=device)
    g = torch.Generator(device=device).manual_seed(3)
    pred = b['gt'] + 2.0 * torch.randn(B, 8, generator=g, device=device)
    _, theta = ops.solve_dlt(b['pts1'], pred, img_w=W, img_h=H)
    dOut = torch.randn(B, H, W, 3, generator=g, device=device)
    U = b['I_aug']
    del b

    def once():
        t = theta.detach().clone().requires_grad_(True)
        out, _ = ops.transformer(U, t, (H, W), with_condition=False)
        out.backward(dOut)

    for _ in range(warm):
        once()
    torch.cuda.synchronize(device)
    _lib.profile_enable(True)
    for _ in range(iters):
        once()
    torch.cuda.synchronize(device)
    prof = _lib.profile_read()
    _lib.profile_enable(False)
    us = {k: (prof[k][0] / prof[k][1] * 1e3 if prof[k][1] else 0.0) for k in ('warp_forward', 'warp_backward', 'warp_backward_finish')}
    nbytes = 4 * B * H * W * 3 * 4
    t_total = (us['warp_forward'] + us['warp_backward'] + us['warp_backward_finish']) * 1e-6
    return {'workload': 'warp fwd+bwd(dTheta), batch %d, %dx%d, C=3, rho=%d, theta = DLT(gt + N(0,2px))' % (B, H, W, rho),
            'fwd_us': round(us['warp_forward'], 2), 'bwd_us': round(us['warp_backward'], 2),
            'bwd_finish_us': round(us['warp_backward_finish'], 2), 'algorithmic_MB': round(nbytes / 1e6, 1),
            'achieved_GBs': round(nbytes / t_total / 1e9, 1), 'frac_of_8TBs': round(nbytes / t_total / HBM_PEAK, 4),
            'fwd_frac': round(nbytes / 2 / (us['warp_forward'] * 1e-6) / HBM_PEAK, 4),
            'bwd_frac': round(nbytes / 2 / (us['warp_backward'] * 1e-6) / HBM_PEAK, 4),
            'tile_paths': tile_paths(theta.detach(), H, W),
            'timing': 'HIP start/stop events of each dispatch (uh_profile_*), %d launches after %d untimed ones' % (iters, warm)}


def forward_temperatures(device, U, theta, H, W, iters=30, evict_mb=1024):
    """The in-step forward launch replayed outside the step on the step's own `U` and theta: `warm` =
    back-to-back launches (the input is served by the 256 MB Infinity Cache / L2), `cold` = a 1 GiB device copy before every
    launch, i.e. the state the train step leaves the caches in (a whole conv stack runs between two warp launches).
    Durations are dispatch start/stop events (uh_profile_*), like the in-step reading.  tools/cold_forward.py sweeps the batch
    size around this point: t(B) is a straight line (residuals <= 0.4 us) -- no round-of-blocks quantisation -- with a fixed
    cost of 9.5 us cold / 7 us warm and an asymptotic streaming rate of 0.78 x 8 TB/s cold (the float4-copy yardstick)."""
    import ctypes as C
    from unsuperviseddeephomographyral2018_amd import _lib
    lib = _lib.load()
    B = U.shape[0]
    out = torch.empty_like(U)
    theta = theta.detach().contiguous()
    ev_a = torch.empty(evict_mb * (1 << 20) // 4, device=device).normal_()
    ev_b = torch.empty_like(ev_a)
    p = lambda t: C.c_void_p(t.data_ptr())
    res = {}
    for temp in ('warm', 'cold'):
        def fwd():
            _lib.check(lib.uh_warp_forward(p(U), p(theta), p(out), None, B, H, W, 3, H, W,
                                           C.c_void_p(torch.cuda.current_stream(device).cuda_stream)), 'uh_warp_forward')
        for _ in range(10):
            fwd()
        torch.cuda.synchronize(device)
        _lib.profile_enable(True, only=('warp_forward',))
        for _ in range(iters):
            if temp == 'cold':
                ev_b.copy_(ev_a)
            fwd()
        torch.cuda.synchronize(device)
        pr = _lib.profile_read()
        _lib.profile_enable(False)
        us = pr['warp_forward'][0] / max(pr['warp_forward'][1], 1) * 1e3
        res[temp + '_us'] = round(us, 2)
        res[temp + '_frac'] = round(2 * B * H * W * 3 * 4 / (us * 1e-6) / HBM_PEAK, 4)
    del ev_a, ev_b, out
    torch.cuda.empty_cache()
    return res


def library_fingerprint():
    """sha256 over the kernel sources + build flags the loaded library was built from (build._fingerprint(); the stamp beside
    the .so when it exists -- the GPU box has no need to re-hash -- else computed)."""
    from unsuperviseddeephomographyral2018_amd import build as uh_build
    try:
        return open(uh_build.LIB + '.sha256').read().strip()
    except Exception:
        return uh_build._fingerprint()


TRAFFIC_FILES = ('traffic_r06.json', 'traffic_r05.json', 'traffic_r04.json', 'traffic_r03.json', 'traffic_r02.json', 'traffic_r01.json')


def committed_traffic(key):
    """HBM bytes per launch from the newest committed PMC file (profiles/traffic_rNN.json) -- NOT measured in this run.
    `stale` says whether the file was measured on other kernel sources than the loaded library's (fingerprint stored in the
    file by tools/gpu_session.sh traffic; files from before round 4 carry none and count as stale)."""
    for name in TRAFFIC_FILES:
        f = os.path.join(ROOT, 'profiles', name)
        if os.path.exists(f):
            try:
                tr = json.load(open(f))
            except Exception:
                continue
            if key in tr:
                fp = tr.get('_fingerprint')
                return {'hbm_bytes_per_launch': tr[key]['hbm_bytes_per_launch'], 'file': 'profiles/' + name,
                        'provenance': tr.get('_provenance', 'no provenance recorded'),
                        'stale': (fp is None) or (fp != library_fingerprint()),
                        'note': 'read from the committed file, not measured in this run'}
    return None


def measure_traffic(args, timeout_s=180):
    """roofline.traffic measured IN THIS RUN: this script re-runs itself twice as a child under
    `rocprofv3 --pmc <counter> --kernel-trace` -- FETCH_SIZE and WRITE_SIZE in their own passes, nothing but --kernel-trace
    beside --pmc, as /opt/skills/guides/MI355X_MICROARCH.md prescribes -- for 5 + 3 steps of the same workload followed by
    the config-4 point, and reads every warp kernel's counters back.  The two shapes share kernel names; they are told apart
    by grid size.  HBM bytes per launch = FETCH_SIZE KiB x 1024 x 2 + WRITE_SIZE KiB x 1024 (gfx950 tallies a 128-byte fabric
    read as 64 B: the guide's correction, re-calibrated in this repo on a device copy of known size, profiles/r01a_pmc_*).
    -> {(kernel, blocks): bytes} keyed as '<kernel>@<blocks>' plus provenance; {'error': ...} when rocprofv3 is unavailable."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    tool = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(tool):
        return {'error': 'rocprofv3 not found'}
    nested = sorted(k for k in os.environ if k.startswith(('ROCPROF', 'ROCP_', 'ROCTRACER', 'RPD_')))
    if nested:                                               # this process is itself being profiled: no profiler inside a profiler
        return {'error': 'running under a profiler (%s): PMC child passes skipped' % ', '.join(nested[:3])}
    child = [sys.executable, os.path.abspath(__file__), '--steps', '5', '--warmup', '3', '--cpu_baseline', '0', '--north_star', '0',
             '--config4', '1' if args.config4 else '0', '--quality', '0', '--traffic', '0', '--traffic_child', '1', '--profile', '0',
             '--per_gpu_batch', str(args.per_gpu_batch), '--img_h', str(args.img_h), '--img_w', str(args.img_w),
             '--patch_size', str(args.patch_size), '--rho', str(args.rho), '--loss_type', args.loss_type,
             '--fused_patch', str(args.fused_patch), '--mid_training_theta', str(args.mid_training_theta), '--tunable_gemm', '0']
    vals = {}
    t0 = time.perf_counter()
    with tempfile.TemporaryDirectory(dir='/tmp') as td:
        env = dict(os.environ, TMPDIR='/tmp')
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            out = os.path.join(td, counter)
            cmd = [tool, '--pmc', counter, '--kernel-trace', '-d', out, '-o', 'p', '--output-format', 'csv', '--'] + child
            try:
                pr = subprocess.Popen(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                      start_new_session=True)
                try:
                    pr.wait(timeout=timeout_s)
                except subprocess.TimeoutExpired:
                    os.killpg(pr.pid, signal.SIGKILL)        # exactly the process group started here
                    pr.wait()
                    return {'error': 'rocprofv3 --pmc %s pass exceeded %d s' % (counter, timeout_s)}
            except Exception as e:                              # noqa: BLE001
                return {'error': 'rocprofv3 --pmc %s: %s' % (counter, e)}
            rows = 0
            for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
                for r in csv.DictReader(open(f)):
                    k = r.get('Kernel_Name', '')
                    if 'uh::warp_' not in k or r.get('Counter_Name') != counter:
                        continue
                    name = k.replace('void ', '').split('(')[0].replace('uh::', '')
                    wg = int(float(r.get('Workgroup_Size', 256) or 256))
                    blocks = int(float(r.get('Grid_Size', 0) or 0)) // max(wg, 1)
                    vals.setdefault((name, blocks), {}).setdefault(counter, []).append(float(r['Counter_Value']))
                    rows += 1
            if rows == 0:
                return {'error': 'rocprofv3 --pmc %s pass produced no warp-kernel rows (rc %s)' % (counter, pr.returncode)}
    res = {'_seconds': round(time.perf_counter() - t0, 1),
           '_how': 'measured in this run: 2 child runs of bench.py under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (own passes, '
                   '--kernel-trace only); bytes = median FETCH_SIZE KiB x 1024 x 2 + median WRITE_SIZE KiB x 1024 per launch '
                   '(gfx950 correction of the microarchitecture guide, calibrated on a device copy)',
           '_fingerprint': library_fingerprint()}
    for (name, blocks), cs in vals.items():
        if 'FETCH_SIZE' in cs and 'WRITE_SIZE' in cs:
            med = lambda v: sorted(v)[len(v) // 2]
            res['%s@%d' % (name, blocks)] = {'hbm_bytes_per_launch': int(med(cs['FETCH_SIZE']) * 2048 + med(cs['WRITE_SIZE']) * 1024),
                                             'FETCH_SIZE_KiB_median': med(cs['FETCH_SIZE']), 'WRITE_SIZE_KiB_median': med(cs['WRITE_SIZE']),
                                             'launches_seen': len(cs['FETCH_SIZE'])}
    return res


def traffic_lookup(measured, family, blocks):
    """bytes per launch of the `family` kernel ('warp_forward_kernel' / 'warp_backward_kernel', dense instantiation) launched
    with `blocks` workgroups, from measure_traffic()'s result; None if absent."""
    if not measured or 'error' in measured:
        return None
    for k, v in measured.items():
        if (k.startswith(family + '<') or k.startswith(family + '@')) and k.endswith('@%d' % blocks) and isinstance(v, dict):
            if family == 'warp_backward_kernel' and not k.split('@')[0].endswith(', false>'):
                continue                                       # PATCH-mode instantiation: the sparse backward of the step
            return v
    return None


def north_star_point(device, args):
    """north_star's point: batch 128, 240x320 (working set 354 MB: partly Infinity-Cache resident)."""
    return warp_point(device, 128, args.img_h, args.img_w, args.patch_size, args.rho, iters=40, warm=40)


def config4_point(device, args):
    """BASELINE.json configs[3]: full-frame 480x640 warp, rho = 64, batch 128 -- 1.9 GB working set, far beyond the
    256 MB Infinity Cache: the HBM-resident roofline point."""
    r = warp_point(device, 128, 480, 640, args.patch_size, 64, iters=30, warm=30)
    r['traffic_fwd'] = committed_traffic('warp_forward_B128_480x640')      # replaced by this run's own PMC passes in main()
    r['traffic_bwd'] = committed_traffic('warp_backward_B128_480x640')
    return r


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: dist.self_launch re-executes this file under
    torch.distributed.run with the same arguments (the driver contract's command line)."""
    from unsuperviseddeephomographyral2018_amd import dist as uh_dist
    uh_dist.self_launch(args.gpus, os.path.abspath(__file__), sys.argv[1:])


def main():
    args = parse()
    # the pool's host driver only supports dmabuf IPC: without this RCCL / device-tensor sharing across processes fails with
    # hipIpcGetMemHandle "invalid argument".  Read by the HSA runtime when the first device is opened -- i.e. after this line.
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        self_launch(args)                                   # does not return
    from unsuperviseddeephomographyral2018_amd import _lib, dist as uh_dist, synthetic
    from unsuperviseddeephomographyral2018_amd.homography_CNN_synthetic import TrainStep, build_parser
    _lib.load()                                             # fail loudly if the HIP library is missing
    rank, world, local = uh_dist.init_from_env()
    args.gpus = world                                       # the launcher's world size is authoritative
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback)')
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)
    # MIOpen find: pick the fastest f32 conv solvers.  (The PMC child runs of measure_traffic() only need the warp kernels'
    # counters: immediate mode there, so that no find-mode trial kernel is profiled.)
    torch.backends.cudnn.benchmark = not args.traffic_child
    uh_dist.skip_naive_conv_in_find()                       # ... without benchmarking MIOpen's reference (naive) solvers
    tuned_gemms = uh_dist.tune_gemms() if args.tunable_gemm else False
    torch.manual_seed(1234)

    if args.cpu_threads_sweep:
        B = args.per_gpu_batch
        batch = synthetic.make_batch(B, args.img_h, args.img_w, args.patch_size, args.rho, seed=100, device=device)
        g = torch.Generator(device=device).manual_seed(4321)
        off = batch['gt'] + 2.0 * torch.randn(B, 8, generator=g, device=device)
        for n in (int(v) for v in args.cpu_threads_sweep.split(',') if v):
            args.cpu_threads = n
            r = cpu_baseline(args, batch, off)
            print(json.dumps({'cpu_threads_requested': n, 'cores_used': r['cores'], 'host_cores': os.cpu_count(),
                              'pairs_per_s': r['value'], 'sample': r['sample']}), flush=True)
        return
    if args.only_points:
        out = {}
        if 'north_star' in args.only_points:
            out['north_star_point'] = north_star_point(device, args)
        if 'config4' in args.only_points:
            out['config4_point'] = config4_point(device, args)
        print(json.dumps(out), flush=True)
        return

    B = args.per_gpu_batch
    targs = build_parser().parse_args([
        '--mode', 'train', '--loss_type', args.loss_type, '--batch_size', str(B * world),
        '--img_h', str(args.img_h), '--img_w', str(args.img_w), '--patch_size', str(args.patch_size),
        '--rho', str(args.rho), '--fused_patch', 'True' if args.fused_patch else 'False',
        '--graph_tail', 'True' if args.graph_tail else 'False', '--step_graph', 'True' if args.step_graph else 'False'])
    if args.step_graph:
        args.profile = 0                                    # per-kernel events cannot be inserted into a replayed graph
    step_fn = TrainStep(targs, device, world)
    # synthetic pairs, generated once, resident in HBM before the timed region; each rank its own shard
    batch = synthetic.make_batch(B, args.img_h, args.img_w, args.patch_size, args.rho, seed=100 + rank,
                                 device=device)
    # A freshly initialised regressor predicts small deltas, i.e. theta ~ identity: the easy case for the warp (every
    # tile interior and staged).  --mid_training_theta 1 (default) adds gt + N(0, 2 px) to its output so that the timed
    # steps run the warp under SURVEY section 8(d)'s mid-training law (clipped and far-field tiles included); the step
    # (conv fwd/bwd, hot path fwd/bwd, Adam) is otherwise unchanged.  `in_step_theta` in the line says which law ran.
    if args.mid_training_theta:
        g = torch.Generator(device=device).manual_seed(4321 + rank)
        step_fn.h4p_offset = batch['gt'] + 2.0 * torch.randn(B, 8, generator=g, device=device)
    find_pass_s = step_fn.prime_conv_finds(batch)           # world > 1: MIOpen find by rank 0 first (untimed, before the warm-up)
    t_w = time.perf_counter()
    for _ in range(args.warmup):
        model = step_fn(batch)
    torch.cuda.synchronize(device)
    warmup_s = time.perf_counter() - t_w                    # MIOpen find (kernel builds + trials) lives here at N = 1
    if world > 1:
        torch.distributed.barrier()
    # profile 1 (default): time only the kernels the roofline object reports -- every timed dispatch costs a small
    # pipeline bubble, and with ~25 library launches per step timing all of them would cost ~10 % of the step
    hot = ('warp_forward', 'warp_backward', 'warp_backward_finish', 'warp_patch_l1_fused', 'warp_patch_l1_finish')
    _lib.profile_enable(bool(args.profile), only=None if args.profile == 2 else hot)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model = step_fn(batch)
    torch.cuda.synchronize(device)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    prof = _lib.profile_read()
    _lib.profile_enable(False)
    from unsuperviseddeephomographyral2018_amd import ops as _ops
    with torch.no_grad():
        _, th_last = _ops.solve_dlt(batch['pts1'], model.pred_h4p.detach(), img_w=args.img_w, img_h=args.img_h)
        in_step_theta = {'law': 'regressor + gt + N(0,2px)' if args.mid_training_theta else 'raw regressor (near identity)',
                         'tile_paths': tile_paths(th_last, args.img_h, args.img_w)}
    # the same in-step forward under the OTHER law, a few extra (untimed) steps after the timed region: lets a reader
    # compare with round 1's line, whose timed steps ran the raw near-identity regressor
    other_law = None
    if args.profile and args.mid_training_theta and not args.step_graph and not args.fused_patch:
        keep = step_fn.h4p_offset
        step_fn.h4p_offset = None
        for _ in range(2):
            step_fn(batch)
        torch.cuda.synchronize(device)
        _lib.profile_enable(True, only=('warp_forward',))
        for _ in range(8):
            m2 = step_fn(batch)
        torch.cuda.synchronize(device)
        p2 = _lib.profile_read()
        _lib.profile_enable(False)
        step_fn.h4p_offset = keep
        if p2['warp_forward'][1]:
            us2 = p2['warp_forward'][0] / p2['warp_forward'][1] * 1e3
            with torch.no_grad():
                _, th2 = _ops.solve_dlt(batch['pts1'], m2.pred_h4p.detach(), img_w=args.img_w, img_h=args.img_h)
            other_law = {'law': 'raw regressor (near identity), 8 untimed steps after the timed region',
                         'warp_forward_avg_us': round(us2, 2),
                         'frac': round(2 * B * args.img_h * args.img_w * 3 * 4 / (us2 * 1e-6) / HBM_PEAK, 4),
                         'tile_paths': tile_paths(th2, args.img_h, args.img_w)}
    # The extra objects are measured after the headline's timed region and must not be able to lose it: a failure in one of
    # them is reported in its place.
    def guarded(fn, *a):
        try:
            return fn(*a)
        except Exception as e:                              # noqa: BLE001 -- reported, not swallowed
            import traceback
            traceback.print_exc(file=sys.stderr)
            return {'error': '%s: %s' % (type(e).__name__, e)}
    # the two warp-only points first, in the allocator / cache state rounds 1-3 measured them in (the warm / cold replay below
    # cycles 2 GiB of evicting buffers, which the Infinity-Cache-assisted north-star point would feel)
    ns_point = guarded(north_star_point, device, args) if (rank == 0 and world == 1 and args.north_star) else None
    c4_point = guarded(config4_point, device, args) if (rank == 0 and world == 1 and args.config4) else None
    temps = None
    if args.profile and world == 1 and not args.step_graph and not args.fused_patch and not args.traffic_child:
        try:
            temps = forward_temperatures(device, batch['I_aug'], th_last, args.img_h, args.img_w)
        except Exception as e:                              # noqa: BLE001 -- an extra, must not lose the headline
            temps = {'error': '%s: %s' % (type(e).__name__, e)}
    loss_val = float(model.loss.detach())

    tt = torch.tensor([dt], dtype=torch.float64, device=device)
    exchange = None
    if world > 1:
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        # the exchange step on its own and the step WITHOUT it (collective: every rank takes part) -- dist.exchange_report
        def run_steps(n):
            for _ in range(n):
                step_fn(batch)
        exchange = uh_dist.exchange_report(step_fn.averager, run_steps, float(tt.item()) / args.steps * 1e3,
                                            resync=(step_fn.net, step_fn.opt))
    dt = float(tt.item())
    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    global_batch = B * world
    value = global_batch * args.steps / dt
    frame_bytes = 2 * B * args.img_h * args.img_w * 3 * 4          # fwd: read U + write out; bwd: read dOut + read U
    kern = {}
    for k, (ms, n) in prof.items():
        if n:
            kern[k] = {'avg_us': round(ms / n * 1e3, 2), 'launches': int(n)}
    if args.fused_patch and 'warp_patch_l1_fused' in kern:
        dom, alg = 'warp_patch_l1_fused', B * args.patch_size ** 2 * (4 * 12 + 12)
    else:
        cands = [k for k in ('warp_forward', 'warp_backward') if k in kern]
        dom = max(cands, key=lambda k: kern[k]['avg_us'] * kern[k]['launches']) if cands else None
        alg = frame_bytes
    if dom is None:                                        # --profile 0: no per-kernel events were taken
        roofline = {'bound': 'hbm', 'kernel': None, 'achieved': None, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
                    'frac': None, 'traffic': None, 'note': 'per-kernel timing disabled (--profile 0)'}
    else:
        t_dom = kern[dom]['avg_us'] * 1e-6
        achieved = alg / t_dom
        roofline = {'bound': 'hbm', 'kernel': dom, 'achieved': round(achieved / 1e9, 1), 'peak': HBM_PEAK / 1e9,
                    'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK, 4), 'traffic': None,
                    'algorithmic_bytes_per_launch': alg, 'avg_launch_us': kern[dom]['avg_us'],
                    'timing': 'HIP start/stop events of each dispatch on its launch stream (hipExtLaunchKernelGGL via '
                              'uh_profile_*), inside the timed region', 'kernels': kern}
    # HBM bytes per launch of the dominant kernel: from the committed PMC file first (flagged when it was measured on other
    # kernel sources than the loaded library's); replaced further down by THIS run's own PMC passes (--traffic 1, N = 1)
    if dom is not None:
        ct = committed_traffic('%s_B%d_%dx%d' % (dom, B, args.img_h, args.img_w))
        if ct is not None:
            roofline['traffic'] = ct['hbm_bytes_per_launch']
            roofline['traffic_stale'] = ct['stale']
            roofline['traffic_note'] = 'NOT measured in this run: read from the committed file %s (%s)%s' % (
                ct['file'], ct['provenance'], '; STALE: measured on other kernel sources than the loaded library' if ct['stale'] else '')
    out = {
        'metric': 'image-pairs/sec (train step) + mean corner error, 128x128 patch RHO=45', 'value': round(value, 1),
        'unit': 'image-pairs/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': '%s train step, batch %d/GPU, %dx%d I/I\', %dx%d patch, RHO=%d, full-frame warp%s' % (
                       'configs[1]: photometric-L1' if args.loss_type == 'l1_loss' else
                       ('configs[4]: supervised 4-pt (reference flag h_loss)' if args.loss_type == 'h_loss' else args.loss_type),
                       B, args.img_h, args.img_w, args.patch_size, args.patch_size, args.rho,
                       ' (fused patch kernel)' if args.fused_patch else ''),
                   'global_batch': global_batch, 'loss_type': args.loss_type, 'parallelism': 'dp%d' % world,
                   'final_loss': round(loss_val, 6),
                   'visible_devices': torch.cuda.device_count(),
                   'device': torch.cuda.get_device_name(device)},
        'roofline': roofline,
    }
    out['config']['in_step_theta'] = in_step_theta
    # where the HEADLINE stands against its own roofline: the step is conv GEMMs (stock MIOpen f32 igemm on f32-input MFMA)
    step_flops = 3 * regressor_flops(args.patch_size) * global_batch
    out['headline_roofline'] = {'bound': 'mfma', 'unit': 'TFLOP/s', 'peak': 157.3, 'achieved': round(step_flops / dt * args.steps / 1e12, 1),
                                'frac': round(step_flops / dt * args.steps / 1e12 / (157.3 * world), 4),
                                'flops_per_step': step_flops,
                                'note': 'whole train step (3 x forward FLOPs of the VGG regressor x global batch) / ms_per_step against the f32-input '
                                        'MFMA peak per GPU (v_mfma_f32_32x32x2_f32 = the f32 vector rate; gfx950 has no xf32 / TF32): the conv '
                                        'stack is stock MIOpen by north_star; hot_path_share_of_step = the library kernels timed in this run '
                                        '(roofline.kernels: avg_us x launches / steps) over ms_per_step; per-kernel step anatomy: '
                                        'profiles/r06_step_breakdown.txt'}
    if kern:
        timed_us = sum(v['avg_us'] * v['launches'] for v in kern.values()) / max(args.steps, 1)
        out['headline_roofline']['hot_path_share_of_step'] = round(timed_us / (dt / args.steps * 1e6), 4)
        out['headline_roofline']['hot_path_kernels_timed'] = sorted(kern)
    out['config']['warmup_seconds'] = round(warmup_s, 1)
    out['config']['tunable_gemm'] = ('on: torch.cuda.tunable picked the fully connected GEMMs (A/B: profiles/r04_tunable_gemm_ab.jsonl; '
                                     '--tunable_gemm 0 = the library heuristic)' if tuned_gemms else 'off')
    out['config']['miopen_find'] = ('cudnn.benchmark = True; reference solvers excluded from the trials '
                                    '(MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD/BWD/WRW=%s)' % os.environ.get('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW'))
    if other_law is not None and isinstance(out.get('roofline'), dict):
        out['roofline']['same_kernel_under_round1_law'] = other_law
    if temps is not None and dom == 'warp_forward':
        t_in = out['roofline']['avg_launch_us']
        temps['in_step_us'] = t_in
        if 'error' not in temps:
            span = max(temps['cold_us'] - temps['warm_us'], 1e-9)
            temps['in_step_position_between_warm_and_cold'] = round((t_in - temps['warm_us']) / span, 3)
            temps['reading'] = ('the in-step launch (%.1f us) lies %.0f %% of the way from the same launch back to back (warm, %.1f us: U served '
                                'by the Infinity Cache) to the same launch after a 1 GiB evicting copy (cold, %.1f us), all three measured in '
                                'this run.  What the cold launch is made of (dispatch overhead, the drain after the last block is '
                                'dispatched, XCDs finishing apart): profiles/r05_launch_anatomy.jsonl, profiles/r05_launch_anatomy_tail.jsonl, '
                                'profiles/r05_pmc_cold_warm.jsonl; batch sweep: profiles/r04_cold_forward.jsonl'
                                % (t_in, 100.0 * (t_in - temps['warm_us']) / span, temps['warm_us'], temps['cold_us']))
        out['roofline']['why_in_step_frac_is_below_the_warm_point'] = temps
    if world > 1:
        out['config']['world_size'] = torch.distributed.get_world_size()
        out['config']['dist_backend'] = torch.distributed.get_backend()
        if out['config']['dist_backend'] == 'nccl':         # only a run that really went over RCCL says so
            out['config']['rccl_world_size'] = torch.distributed.get_world_size()
            try:
                out['config']['rccl_version'] = '.'.join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                out['config']['rccl_version'] = None
        out['config']['conv_find_pass_s'] = round(find_pass_s, 1)
        out['config']['conv_find_staggered'] = os.environ.get('UH_FIND_STAGGER', '1') != '0'
        out['config']['model_rng'] = 'seed + rank per tower (independent dropout masks), variables broadcast from rank 0'
        out['exchange'] = exchange
    if ns_point is not None:
        out['north_star_point'] = ns_point
    if c4_point is not None:
        out['config4_point'] = c4_point
    if world == 1 and args.traffic and not args.traffic_child and dom is not None:
        mt = guarded(measure_traffic, args)
        out['roofline']['traffic_measurement'] = {k: v for k, v in mt.items() if k.startswith('_') or k == 'error'}
        blocks_of = lambda nb, hh, ww: nb * ((ww + 63) // 64) * ((hh + 15) // 16)
        if dom == 'warp_forward':
            hit = traffic_lookup(mt, 'warp_forward_kernel', blocks_of(B, args.img_h, args.img_w))
            if hit is not None:
                out['roofline']['traffic'] = hit['hbm_bytes_per_launch']
                out['roofline']['traffic_stale'] = False
                out['roofline']['traffic_note'] = mt['_how'] + '; %d launches of the in-step shape seen' % hit['launches_seen']
                out['roofline']['traffic_over_algorithmic'] = round(hit['hbm_bytes_per_launch'] / alg, 4)
        if isinstance(out.get('config4_point'), dict) and 'error' not in out['config4_point']:
            for side, fam in (('traffic_fwd', 'warp_forward_kernel'), ('traffic_bwd', 'warp_backward_kernel')):
                hit = traffic_lookup(mt, fam, blocks_of(128, 480, 640))
                if hit is not None:
                    out['config4_point'][side] = {'hbm_bytes_per_launch': hit['hbm_bytes_per_launch'], 'stale': False,
                                                  'launches_seen': hit['launches_seen'], 'note': 'measured in this run (PMC child passes)'}
    if world == 1 and args.cpu_baseline:
        out['cpu_baseline'] = guarded(cpu_baseline, args, batch, step_fn.h4p_offset)
    if world == 1 and args.quality and args.loss_type == 'l1_loss':
        del step_fn, model
        torch.cuda.empty_cache()
        out['quality'] = guarded(quality_run, device, args)
    out['timing_note'] = ('per-dispatch HIP events on the warp kernels inside the timed region (--profile 1) cost a small pipeline bubble '
                          'each; --profile 0 gives the headline without them.  The events are created with hipEventDisableSystemFence (a '
                          'default event writes back and invalidates the caches when it completes and lengthens the launches it '
                          'brackets; UH_PROF_FENCE=1 restores that; A/B: profiles/r04_prefetch_ab_library_side_stream.jsonl)') if args.profile else None
    print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
