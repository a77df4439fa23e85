#!/usr/bin/env python
"""End-to-end run on the reference's ON-DISK format (DESIGN section 8 "real-data run"):
write a synthetic dataset in the layout of utils/gen_synthetic_data.py (I/, I_prime/ JPEG files -- the uint8 round trip
included --, filenames / pts1 / gt text files, train and test splits), then train the unsupervised photometric l1_loss
FROM THOSE FILES through dataloader.Dataloader (decode worker processes -> uh_prepare_inputs with the joint photometric
augmentation at do_augment = 0.5) with the reference's hyper-parameters, and evaluate with the reference's test loop on the
held-out split read from disk as well.  Prints pairs/s of the whole pipeline and the test statistics."""
import argparse
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsuperviseddeephomographyral2018_amd import dataloader as D, synthetic  # noqa: E402
from unsuperviseddeephomographyral2018_amd.homography_CNN_synthetic import TestHomography, build_parser, train  # noqa: E402


def write_split(root, prefix, n_pairs, seed0, dev, B=256):
    frames, framesp, pts, gts = [], [], [], []
    for i in range(n_pairs // B):
        b = synthetic.make_batch(B, 240, 320, 128, 45, seed=seed0 + i, device=dev, kind='multiscale')
        to_u8 = lambda t: (t * 50.0 + 128.0).clamp(0, 255).to(torch.uint8).cpu().numpy()
        frames.append(to_u8(b['I_aug'])); framesp.append(to_u8(b['I_prime_aug']))
        pts.append(b['pts1'].cpu().numpy()); gts.append(b['gt'].cpu().numpy())
    # (file names must differ between the splits: both live under I/ and I_prime/)
    I = np.concatenate(frames); Ip = np.concatenate(framesp)
    os.makedirs(os.path.join(root, 'I'), exist_ok=True); os.makedirs(os.path.join(root, 'I_prime'), exist_ok=True)
    from PIL import Image
    names = []
    for k in range(len(I)):
        name = '%s%d.jpg' % (prefix, k)
        Image.fromarray(I[k]).save(os.path.join(root, 'I', name)); Image.fromarray(Ip[k]).save(os.path.join(root, 'I_prime', name))
        names.append(name)
    ff, fp, fg = (os.path.join(root, prefix + s) for s in ('filenames.txt', 'pts1.txt', 'gt.txt'))
    with open(ff, 'w') as f:
        f.writelines('%s %s\n' % (n, n) for n in names)
    np.savetxt(fp, np.concatenate(pts), delimiter=' '); np.savetxt(fg, np.concatenate(gts), delimiter=' ')
    return ff, fp, fg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--train_pairs', type=int, default=16384)
    ap.add_argument('--test_pairs', type=int, default=1024)
    ap.add_argument('--steps', type=int, default=8000)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    root = tempfile.mkdtemp(prefix='uh_dataset_')
    try:
        t0 = time.time()
        tr = write_split(root, 'train_', a.train_pairs, 7000, dev)
        te = write_split(root, 'test_', a.test_pairs, 9000, dev)
        nbytes = sum(os.path.getsize(os.path.join(root, 'I', f)) for f in os.listdir(os.path.join(root, 'I')))
        print('dataset: %d + %d pairs as JPEG under %s, %.1f s to write, %.1f kB per image' % (
            a.train_pairs, a.test_pairs, root, time.time() - t0, nbytes / (a.train_pairs + a.test_pairs) / 1e3), flush=True)
        args = build_parser().parse_args([
            '--mode', 'train', '--loss_type', 'l1_loss', '--batch_size', '64', '--lr', '1e-4', '--do_augment', '0.5',
            '--data_path', root + '/', '--filenames_file', tr[0], '--pts1_file', tr[1], '--gt_file', tr[2],
            '--test_filenames_file', te[0], '--test_pts1_file', te[1], '--test_gt_file', te[2],
            '--num_total_steps', str(a.steps), '--log_every', '1000', '--save_every', '1000000',
            '--model_dir', os.path.join(root, 'models')])
        t0 = time.time()
        step_fn = train(args)
        torch.cuda.synchronize()
        dt = time.time() - t0
        print('trained %d steps from disk in %.1f s = %.0f pairs/s (MIOpen find and worker start-up included)' % (
            a.steps, dt, a.steps * 64 / dt), flush=True)
        res = TestHomography(args, step_fn=step_fn).run()
        print('held-out pairs READ FROM DISK: mean corner error %.3f px, %.2f %% failures, %d pairs (identity ~ 25.9 px)' % (
            res['mean_corner_error'], res['fail_percent'], res['num_pairs']), flush=True)
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == '__main__':
    main()
