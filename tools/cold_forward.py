#!/usr/bin/env python
"""Why does the in-step warp forward (batch 64, 240x320) sit at 0.51-0.57 of the HBM roofline when the same kernel reaches
0.85 at batch 128 in a back-to-back loop?  ("Next round" #2.)  Developer tool, not bench.py.

Splits the gap into (i) COLD input -- inside the train step `U` was last touched a whole conv stack ago, so neither the 4 MB
L2 of an XCD nor the 256 MB Infinity Cache holds it -- (ii) GRID TAIL -- B=64 is 4 800 blocks over 256 CUs x 6 resident
blocks = 3.1 rounds -- and (iii) the FIXED cost of a launch (ramp-up, drain), by timing uh_warp_forward

    * warm  : back-to-back launches on the same tensors (what tools/microbench.py and the north-star point measure),
    * cold  : an evicting device copy (>= 1 GiB, four times the Infinity Cache) between launches,
    * cold+readU (+fill_out): the same, then `U` read once (and `out` written once) before the launch,
    * for B in a sweep that crosses 2.3 / 3.1 / 3.8 / 4.7 / 6.25 rounds of 1 536 resident blocks,
    * under both theta laws: `hard` = DLT(gt + N(0, 2 px)) (SURVEY 8d mid-training law: clipped / far-field tiles) and
      `easy` = DLT(N(0, 1 px)) (what a 25-step-old regressor predicts: every tile interior and staged).

Durations are the dispatch's own start/stop events (uh_profile_*).  A least-squares line t(B) = a + b*B over the sweep gives
the fixed cost a and the per-pair cost b for each (law, temperature); the residuals show whether whole rounds matter.
One JSON line per point, then one `fit` line per (law, temperature).
"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsuperviseddeephomographyral2018_amd import _lib, ops, synthetic  # noqa: E402

PEAK = 8.0e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batches', default='32,48,64,77,96,128')
    ap.add_argument('--iters', type=int, default=40)
    ap.add_argument('--evict_mb', type=int, default=1024)
    ap.add_argument('--h', type=int, default=240)
    ap.add_argument('--w', type=int, default=320)
    ap.add_argument('--rho', type=int, default=45)
    ap.add_argument('--tag', default='')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    lib = _lib.load()
    H, W, P = args.h, args.w, 128
    ev_a = torch.empty(args.evict_mb * (1 << 20) // 4, device=dev).normal_()
    ev_b = torch.empty_like(ev_a)
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    resident = 256 * 6
    sink = torch.zeros((), device=dev)
    fits = {}
    for B in [int(v) for v in args.batches.split(',')]:
        b = synthetic.make_batch(B, H, W, P, args.rho, seed=100, device=dev)
        U = b['I_aug']
        g = torch.Generator(device=dev).manual_seed(4321)
        noise = torch.randn(B, 8, generator=g, device=dev)
        laws = {'hard': b['gt'] + 2.0 * noise, 'easy': 1.0 * noise}
        out = torch.empty_like(U)
        blocks = B * ((H + 15) // 16) * ((W + 63) // 64)
        for law, h4p in laws.items():
            _, theta = ops.solve_dlt(b['pts1'], h4p, img_w=W, img_h=H)
            theta = theta.detach().contiguous()

            def fwd():
                _lib.check(lib.uh_warp_forward(p(U), p(theta), p(out), None, B, H, W, 3, H, W, st()), 'uh_warp_forward')
            # cold+readU: after the evicting copy, U is read once (a reduction) before the launch -- what a side-stream prefetch
            # issued under the fc layers would leave behind; cold+readU+fill_out additionally writes `out` once: together they
            # say how much of the cold penalty is U coming from HBM, how much the output lines / address translation
            for temp in ('warm', 'cold', 'cold+readU', 'cold+readU+fill_out'):
                for _ in range(10):
                    fwd()
                torch.cuda.synchronize()
                _lib.profile_enable(True, only=('warp_forward',))
                for _ in range(args.iters):
                    if temp != 'warm':
                        ev_b.copy_(ev_a)                 # 2 x evict_mb of traffic: L2 and Infinity Cache now hold the copy's lines
                    if temp.startswith('cold+readU'):
                        sink.copy_(U.sum())
                    if temp.endswith('fill_out'):
                        out.fill_(0.5)
                    fwd()
                torch.cuda.synchronize()
                prof = _lib.profile_read()
                _lib.profile_enable(False)
                us = prof['warp_forward'][0] / prof['warp_forward'][1] * 1e3
                alg = 2 * B * H * W * 3 * 4
                rec = {'B': B, 'law': law, 'temp': temp, 'us': round(us, 2), 'frac': round(alg / (us * 1e-6) / PEAK, 4),
                       'blocks': blocks, 'rounds_of_1536': round(blocks / resident, 3), 'us_per_pair': round(us / B, 4),
                       'lib': os.path.basename(_lib.LIB_PATH), 'tag': args.tag}
                print(json.dumps(rec), flush=True)
                fits.setdefault((law, temp), []).append((B, us))
        del b, U, out
        torch.cuda.empty_cache()
    for (law, temp), pts in fits.items():
        n = len(pts)
        sx = sum(x for x, _ in pts); sy = sum(y for _, y in pts)
        sxx = sum(x * x for x, _ in pts); sxy = sum(x * y for x, y in pts)
        slope = (n * sxy - sx * sy) / (n * sxx - sx * sx)
        icpt = (sy - slope * sx) / n
        res = {str(x): round(y - (icpt + slope * x), 2) for x, y in pts}
        print(json.dumps({'fit': '%s/%s' % (law, temp), 'fixed_us': round(icpt, 2), 'us_per_pair': round(slope, 4),
                          'asymptotic_frac': round(2 * H * W * 3 * 4 / (slope * 1e-6) / PEAK, 4), 'residual_us': res,
                          'tag': args.tag}), flush=True)


if __name__ == '__main__':
    main()
