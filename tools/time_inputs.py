#!/usr/bin/env python
"""SURVEY section 8 f3 -- do the GPU input producers keep up with a > 10^4 pairs/s train step?

  1. uh_prepare_inputs (uint8 frames -> the dataloader's 7 output tensors) at B=64, 240x320, with and without the
     photometric augmentation: kernel duration (HIP events of the dispatch, uh_profile_*), algorithmic bytes
     (2 x 3 B read + 2 x 12 B written per pixel, + 4 gray patches and the index patch: 5 x P^2 x 4 B per pair), fraction
     of the 8 TB/s HBM roofline; wall time of the whole host call (allocations + zero fills of the patch tensors included).
  2. The in-HBM synthetic generator (synthetic.make_batch: texture, DLT in f64, warp, gray patches) alone: pairs/s.
  3. Training pairs/s (full train step, photometric l1_loss, B=64) fed by
       (a) a pre-generated in-HBM pool        (--data_pool: what every training log of rounds 1-2 used)
       (b) the generator in the loop          (--fresh_data_every 1)
       (c) the reference's on-disk format     (--data_path: PNG / JPEG files written by dataloader.write_dataset, decoded
                                               with PIL on 20 host threads, uh_prepare_inputs on the GPU), without and
                                               with the prefetching producer thread
     and the disk loader alone (no training): pairs/s of decode + upload + uh_prepare_inputs.
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsuperviseddeephomographyral2018_amd import _lib, dataloader as D, synthetic  # noqa: E402
from unsuperviseddeephomographyral2018_amd.homography_CNN_synthetic import TrainStep, build_parser  # noqa: E402

PEAK = 8.0e12


def kernel_us(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    _lib.profile_enable(True, only=('prepare_inputs',))
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / iters
    ms, n = _lib.profile_read()['prepare_inputs']
    _lib.profile_enable(False)
    return ms / n * 1e3, wall * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--B', type=int, default=64)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--disk_pairs', type=int, default=1024)
    ap.add_argument('--jpg_only', type=int, default=1, help='0: also time a PNG copy of the set (30 s to write)')
    ap.add_argument('--kernel_only', type=int, default=0, help='1: only part 1 (the run rocprofv3 --kernel-trace --stats wraps)')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    torch.backends.cudnn.benchmark = True
    B, H, W, P, rho = a.B, 240, 320, 128, 45
    out = {'B': B, 'H': H, 'W': W, 'P': P, 'host_cores': os.cpu_count()}

    # ---- 1. the kernel
    g = torch.Generator(device=dev).manual_seed(0)
    I8 = torch.randint(0, 256, (B, H, W, 3), generator=g, device=dev, dtype=torch.uint8)
    Ip8 = torch.randint(0, 256, (B, H, W, 3), generator=g, device=dev, dtype=torch.uint8)
    x0 = torch.randint(rho, W - rho - P + 1, (B,), generator=g, device=dev)
    y0 = torch.randint(rho, H - rho - P + 1, (B,), generator=g, device=dev)
    pts1 = torch.stack([x0, y0, x0 + P, y0, x0 + P, y0 + P, x0, y0 + P], 1).float()
    alg = B * (H * W * (2 * 3 + 2 * 12) + 5 * P * P * 4)
    for name, aug in (('no_augmentation', None), ('joint_augmentation', D.sample_augmentation(B, 'train', 1.0))):
        us, wall_us = kernel_us(lambda: D.prepare_inputs(I8, Ip8, pts1, P, aug), 50)
        out['prepare_inputs_' + name] = {
            'kernel_us': round(us, 2), 'algorithmic_MB': round(alg / 1e6, 2), 'achieved_GBs': round(alg / us / 1e3, 1),
            'frac_of_8TBs': round(alg / (us * 1e-6) / PEAK, 4), 'pairs_per_s_kernel_only': round(B / (us * 1e-6)),
            'host_call_wall_us': round(wall_us, 1), 'pairs_per_s_host_call': round(B / (wall_us * 1e-6))}
    print(json.dumps({k: v for k, v in out.items() if k.startswith('prepare')}, indent=1), flush=True)

    if a.kernel_only:
        return
    # ---- 2. the generator alone
    for kind in ('smooth', 'multiscale'):
        for i in range(3):
            synthetic.make_batch(B, H, W, P, rho, seed=i, device=dev, kind=kind)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 100
        for i in range(n):
            synthetic.make_batch(B, H, W, P, rho, seed=10 + i, device=dev, kind=kind)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        out['generator_alone_' + kind] = {'ms_per_batch': round(dt * 1e3, 3), 'pairs_per_s': round(B / dt)}
    print(json.dumps({k: v for k, v in out.items() if k.startswith('generator')}, indent=1), flush=True)

    # ---- 3. training fed three ways
    targs = build_parser().parse_args(['--mode', 'train', '--loss_type', 'l1_loss', '--batch_size', str(B)])
    torch.manual_seed(0)
    step = TrainStep(targs, dev, 1)

    def train_rate(next_batch, steps, warm=10):
        for _ in range(warm):
            step(next_batch())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(next_batch())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return {'pairs_per_s': round(B * steps / dt, 1), 'ms_per_step': round(dt / steps * 1e3, 3), 'steps': steps}

    pool = [synthetic.make_batch(B, H, W, P, rho, seed=100 + i, device=dev) for i in range(32)]
    it = [0]

    def from_pool():
        it[0] += 1
        return pool[it[0] % len(pool)]
    out['train_data_pool'] = train_rate(from_pool, a.steps)

    def fresh():
        it[0] += 1
        return synthetic.make_batch(B, H, W, P, rho, seed=1000 + it[0], device=dev)
    out['train_fresh_data_every_1'] = train_rate(fresh, a.steps)
    print(json.dumps({k: v for k, v in out.items() if k.startswith('train')}, indent=1), flush=True)

    # on-disk set in the reference's layout, written from synthetic frames (uint8 round trip as gen_synthetic_data.py does)
    tmp = tempfile.mkdtemp(prefix='uh_disk_')
    try:
        for fmt in ('jpg',) if a.jpg_only else ('jpg', 'png'):
            root = os.path.join(tmp, fmt)
            n_pairs = a.disk_pairs
            frames, framesp, pts, gts = [], [], [], []
            for i in range(n_pairs // B):
                b = synthetic.make_batch(B, H, W, P, rho, seed=5000 + i, device=dev, kind='multiscale')
                to_u8 = lambda t: (t * 50.0 + 128.0).clamp(0, 255).to(torch.uint8).cpu().numpy()
                frames.append(to_u8(b['I_aug'])); framesp.append(to_u8(b['I_prime_aug']))
                pts.append(b['pts1'].cpu().numpy()); gts.append(b['gt'].cpu().numpy())
            t0 = time.perf_counter()
            ff, fp, fg = D.write_dataset(root, np.concatenate(frames), np.concatenate(framesp), np.concatenate(pts),
                                         np.concatenate(gts), fmt=fmt)
            t_write = time.perf_counter() - t0
            prm = D.dataloader_params(data_path=root, filenames_file=ff, pts1_file=fp, gt_file=fg, mode='train', batch_size=B,
                                      img_h=H, img_w=W, patch_size=P, augment_list=['normalize'], do_augment=0.5)
            res = {'pairs_on_disk': n_pairs, 'write_seconds': round(t_write, 1),
                   'bytes_per_image': int(os.path.getsize(os.path.join(root, 'I', '0.' + fmt)))}
            for pf, nw in ((0, 0), (4, 0), (4, 8), (4, 16), (4, 48)):
                loader = D.Dataloader(prm, shuffle=True, device=dev, seed=1, num_workers=nw)
                st = loader.stream(prefetch=pf)
                for _ in range(3):
                    next(st)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = 30
                for _ in range(n):
                    next(st)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n
                res['loader_alone_prefetch%d_workers%d' % (pf, nw)] = {'ms_per_batch': round(dt * 1e3, 2), 'pairs_per_s': round(B / dt)}
                st2 = D.Dataloader(prm, shuffle=True, device=dev, seed=2, num_workers=nw).stream(prefetch=pf)
                res['train_data_path_prefetch%d_workers%d' % (pf, nw)] = train_rate(lambda: next(st2), min(a.steps, 100), warm=5)
                st.close(); st2.close()
                del st, st2
            out['disk_' + fmt] = res
            print(json.dumps({'disk_' + fmt: res}, indent=1), flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print('SUMMARY ' + json.dumps(out))


if __name__ == '__main__':
    main()
