#!/usr/bin/env python
"""What the reference's photometric augmentation (dataloader.py:323-375) does to [0, 255] images -- CPU only, numpy + the oracle.

    img ** gamma, gamma ~ U(0.8, 1.2);  * brightness ~ U(0.5, 2.0);  * colour ~ U(0.8, 1.2)^3;  clip to [0, 255]

The gamma shift is applied to RAW 8-bit intensities (tf.cast(image, tf.float32), dataloader.py:244 -- not to [0, 1] values): 128 ** 1.2
= 338, 200 ** 1.1 = 340, so for gamma > 1 most of an image is pushed past 255 and clipped; for gamma < 1 it is crushed into the
dark third (128 ** 0.8 = 48).  This script measures, over many draws, how much of an image survives: the share of pixels clipped
to 255, the share of images with more than half / 90 % of their pixels clipped, and -- for the DISJOINT law the test mode uses --
how often at least one image of a pair is destroyed.  Pixel populations: the on-disk synthetic dataset of tools/train_from_disk.py
(uint8 of 128 + 50 * N(0,1) textures) and a broad "photograph-like" population with the reference's own channel statistics
(mean 118.93 / 113.97 / 102.60, std 69.85 / 68.81 / 72.45, dataloader.py:99-100).

usage: python tools/augmentation_saturation.py > profiles/r05_augmentation_saturation.txt
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import hotpath_numpy as O  # noqa: E402

MEAN = np.array([118.93, 113.97, 102.60]); STD = np.array([69.85, 68.81, 72.45])


def population(kind, rs, n_img, hw=(24, 32)):
    H, W = hw
    if kind == 'synthetic':                 # tools/train_from_disk.py: (t * 50 + 128).clamp(0, 255), t unit-variance texture
        x = 128.0 + 50.0 * rs.randn(n_img, H, W, 3)
    else:                                   # photograph-like: per-image exposure + within-image spread, channel stats of the reference
        base = MEAN + 0.6 * STD * rs.randn(n_img, 1, 1, 3)
        x = base + 0.8 * STD * rs.randn(n_img, H, W, 3)
    return np.clip(np.round(x), 0, 255).astype(np.uint8)


def draw(rs, n):
    return np.concatenate([rs.uniform(0.8, 1.2, (n, 1)), rs.uniform(0.5, 2.0, (n, 1)), rs.uniform(0.8, 1.2, (n, 3))], 1)


def main():
    rs = np.random.RandomState(0)
    n = 4000
    print('reference augmentation law on [0,255] images: clip(img ** U(0.8,1.2) * U(0.5,2.0) * U(0.8,1.2)^3, 0, 255)   (dataloader.py:323-375)')
    print('%d images per population, %d draws; "clipped" = pixel channel value == 255 after the clip\n' % (n, n))
    for kind in ('synthetic', 'photograph-like'):
        I = population(kind, rs, n)
        a0, a1 = draw(rs, n), draw(rs, n)
        pts1 = np.zeros((n, 8), np.float32)
        aug = np.stack([a0, a1], 1).astype(np.float32)                       # [n, 2, 5]: image I gets a0, image I' gets a1
        out = O.prepare_inputs(I, I, pts1, 4, aug=aug, mean=(0, 0, 0), std=(1, 1, 1))
        A, Bq = out['I_aug'], out['I_prime_aug']                             # augmented, NOT standardised (mean 0, std 1)
        sat = (A >= 255.0).mean(axis=(1, 2, 3))
        sat_b = (Bq >= 255.0).mean(axis=(1, 2, 3))
        # information left: standard deviation of the augmented image relative to the plain one
        keep = A.std(axis=(1, 2, 3)) / np.maximum(I.astype(np.float32).std(axis=(1, 2, 3)), 1e-6)
        print('== %s population (mean %.1f, std %.1f, %.1f %% of raw pixels already at 0 or 255)' % (
            kind, I.mean(), I.std(), 100.0 * ((I == 0) | (I == 255)).mean()))
        print('   one augmented image: mean share of clipped pixels %.1f %%; images with > 50 %% clipped: %.1f %%; > 90 %% clipped: %.1f %%'
              % (100 * sat.mean(), 100 * (sat > 0.5).mean(), 100 * (sat > 0.9).mean()))
        print('   contrast kept (std after / std before): median %.2f, images keeping < 25 %% of their contrast: %.1f %%'
              % (np.median(keep), 100 * (keep < 0.25).mean()))
        for g_lo, g_hi in ((0.8, 0.9), (0.9, 1.0), (1.0, 1.1), (1.1, 1.2)):
            m = (a0[:, 0] >= g_lo) & (a0[:, 0] < g_hi)
            print('     gamma in [%.1f, %.1f): mean clipped share %.1f %%, > 50 %% clipped in %.1f %% of images, mean output level %.0f'
                  % (g_lo, g_hi, 100 * sat[m].mean(), 100 * (sat[m] > 0.5).mean(), A[m].mean()))
        either = (sat > 0.5) | (sat_b > 0.5)
        print('   DISJOINT pair (test mode, independent draws for I and I\'): at least one image > 50 %% clipped in %.1f %% of augmented pairs'
              % (100 * either.mean()))
        d = np.abs(A - Bq).mean(axis=(1, 2, 3))
        print('   DISJOINT pair: mean |I_aug - I\'_aug| of the SAME underlying image = %.1f gray levels (median %.1f); JOINT pair: 0 by construction'
              % (d.mean(), np.median(d)))
        for p in (0.5, 1.0):
            print('   with do_augment = %.1f: %.1f %% of all test pairs have an image that is more than half clipped' % (p, 100 * p * either.mean()))
        print()


if __name__ == '__main__':
    main()
