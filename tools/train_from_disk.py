#!/usr/bin/env python
"""End-to-end runs on the reference's ON-DISK format: write a synthetic dataset
in the layout of utils/gen_synthetic_data.py (I/, I_prime/ image files -- the uint8 round trip included --, filenames / pts1 /
gt text files, train and test splits), train the unsupervised photometric l1_loss FROM THOSE FILES through
dataloader.Dataloader (decode worker processes -> uh_prepare_inputs with the reference's joint photometric augmentation) with
the reference's hyper-parameters, and evaluate with the reference's test loop on the held-out split read from disk as well --
once with the reference's DISJOINT test augmentation (dataloader.py:163-169) and once without it.

One process runs several ARMS so that datasets are written once per file format:

    --arms jpg:0.5,jpg:0,png:0.5,png:0      <format>:<train do_augment>; every arm starts from the same fresh initialisation
    --test_do_augment 0.5,0                 every arm is evaluated under each of these (test mode = disjoint draws)

Prints one JSON line per (arm, test setting) and a table at the end."""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsuperviseddeephomographyral2018_amd import synthetic  # noqa: E402
from unsuperviseddeephomographyral2018_amd.homography_CNN_synthetic import TestHomography, build_parser, train  # noqa: E402


def write_split(root, prefix, n_pairs, seed0, dev, fmt, B=256):
    from PIL import Image
    os.makedirs(os.path.join(root, 'I'), exist_ok=True); os.makedirs(os.path.join(root, 'I_prime'), exist_ok=True)
    to_u8 = lambda t: (t * 50.0 + 128.0).clamp(0, 255).to(torch.uint8).cpu().numpy()
    kw = {'compress_level': 1} if fmt == 'png' else {}
    names, pts, gts, k = [], [], [], 0
    with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 4)) as ex:       # PIL's encoders release the interpreter lock
        for i in range(-(-n_pairs // B)):
            b = synthetic.make_batch(B, 240, 320, 128, 45, seed=seed0 + i, device=dev, kind='multiscale')
            I, Ip = to_u8(b['I_aug']), to_u8(b['I_prime_aug'])
            pts.append(b['pts1'].cpu().numpy()); gts.append(b['gt'].cpu().numpy())
            jobs = []
            for j in range(len(I)):
                if k >= n_pairs:
                    break
                name = '%s%d.%s' % (prefix, k, fmt)              # (file names must differ between the splits: both live under I/, I_prime/)
                jobs.append(ex.submit(Image.fromarray(I[j]).save, os.path.join(root, 'I', name), **kw))
                jobs.append(ex.submit(Image.fromarray(Ip[j]).save, os.path.join(root, 'I_prime', name), **kw))
                names.append(name); k += 1
            for f in jobs:
                f.result()
    ff, fp, fg = (os.path.join(root, prefix + s) for s in ('filenames.txt', 'pts1.txt', 'gt.txt'))
    with open(ff, 'w') as f:
        f.writelines('%s %s\n' % (n, n) for n in names)
    np.savetxt(fp, np.concatenate(pts)[:n_pairs], delimiter=' '); np.savetxt(fg, np.concatenate(gts)[:n_pairs], delimiter=' ')
    return ff, fp, fg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--train_pairs', type=int, default=16384)
    ap.add_argument('--test_pairs', type=int, default=1024)
    ap.add_argument('--steps', type=int, default=8000)
    ap.add_argument('--lr', type=float, default=1e-4)
    ap.add_argument('--batch_size', type=int, default=64)
    ap.add_argument('--arms', default='jpg:0.5', help='comma list of <fmt>:<train do_augment>, fmt in jpg / png')
    ap.add_argument('--test_do_augment', default='0.5,0', help='comma list; every arm is evaluated under each (disjoint draws)')
    ap.add_argument('--log_every', type=int, default=1000)
    ap.add_argument('--root', default='', help='dataset directory (default: a fresh one under $TMPDIR, removed at the end)')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    arms = [(x.split(':')[0], float(x.split(':')[1])) for x in a.arms.split(',') if x]
    tests = [float(x) for x in a.test_do_augment.split(',') if x != '']
    root = a.root or tempfile.mkdtemp(prefix='uh_dataset_')
    fs = os.statvfs(root)
    print('dataset root %s: %.1f GB free' % (root, fs.f_bavail * fs.f_frsize / 1e9), flush=True)
    table, files = [], {}
    try:
        for fmt, aug in arms:
            if fmt not in files:
                t0 = time.time()
                d = os.path.join(root, fmt)
                tr = write_split(d, 'train_', a.train_pairs, 7000, dev, fmt)
                te = write_split(d, 'test_', a.test_pairs, 9000, dev, fmt)
                nbytes = sum(os.path.getsize(os.path.join(d, 'I', f)) for f in os.listdir(os.path.join(d, 'I')))
                print('dataset: %d + %d pairs as %s under %s, %.1f s to write, %.1f kB per image' % (
                    a.train_pairs, a.test_pairs, fmt.upper(), d, time.time() - t0, nbytes / (a.train_pairs + a.test_pairs) / 1e3), flush=True)
                files[fmt] = (d, tr, te)
            d, tr, te = files[fmt]
            argv = ['--mode', 'train', '--loss_type', 'l1_loss', '--batch_size', str(a.batch_size), '--lr', str(a.lr),
                    '--do_augment', str(aug), '--data_path', d + '/', '--filenames_file', tr[0], '--pts1_file', tr[1], '--gt_file', tr[2],
                    '--test_filenames_file', te[0], '--test_pts1_file', te[1], '--test_gt_file', te[2],
                    '--num_total_steps', str(a.steps), '--log_every', str(a.log_every), '--save_every', '1000000000',
                    '--model_dir', os.path.join(root, 'models_%s_%s' % (fmt, aug)), '--seed', '0']
            args = build_parser().parse_args(argv)
            print('=== arm: files %s, train do_augment %.2f (joint), %d steps, lr %g, batch %d' % (fmt.upper(), aug, a.steps, a.lr, a.batch_size), flush=True)
            t0 = time.time()
            step_fn = train(args)
            torch.cuda.synchronize()
            dt = time.time() - t0
            print('trained %d steps from disk in %.1f s = %.0f pairs/s (MIOpen find and worker start-up included)' % (
                a.steps, dt, a.steps * a.batch_size / dt), flush=True)
            for t_aug in tests:
                targs = argparse.Namespace(**dict(vars(args), do_augment=t_aug))
                res = TestHomography(targs, step_fn=step_fn).run()
                row = {'files': fmt, 'train_do_augment': aug, 'test_do_augment': t_aug, 'steps': a.steps, 'train_pairs': a.train_pairs,
                       'mean_corner_error_px': round(res['mean_corner_error'], 3), 'fail_percent': round(res['fail_percent'], 2),
                       'per_pair_median_px': round(res['percentiles'][50], 3),
                       'reference_percentile_intervals': np.round(np.array(res['reference_percentile_intervals']), 3).tolist(),
                       'test_pairs_x_passes': res['num_pairs'], 'train_pairs_per_s': round(a.steps * a.batch_size / dt)}
                table.append(row)
                print('RESULT ' + json.dumps(row), flush=True)
            del step_fn
            torch.cuda.empty_cache()
        print('\n| files | train do_augment (joint) | test do_augment (disjoint) | mean corner error px | failures % | per-pair median px |')
        print('|---|---|---|---|---|---|')
        for r in table:
            print('| %s | %.1f | %.1f | %.2f | %.2f | %.2f |' % (r['files'].upper(), r['train_do_augment'], r['test_do_augment'],
                                                              r['mean_corner_error_px'], r['fail_percent'], r['per_pair_median_px']))
        print('(identity predictor ~ 25.9 px; %d steps of the reference\'s 150 000; synthetic multiscale textures, not MS-COCO)' % a.steps)
    finally:
        if not a.root:
            shutil.rmtree(root, ignore_errors=True)


if __name__ == '__main__':
    main()
