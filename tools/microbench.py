#!/usr/bin/env python
"""Kernel-level micro-benchmark of the hot path on one MI355X (developer tool, not bench.py).

For each workload: times uh_warp_forward / uh_warp_backward / the fused patch kernel through the
C ABI with (a) the in-library HIP-event profiler and (b) torch events over a back-to-back loop,
next to a same-bytes device copy, and prints achieved algorithmic GB/s vs the 8 TB/s roofline.
"""
import argparse
import json
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsuperviseddeephomographyral2018_amd import ops, _lib  # noqa: E402

PEAK = 8.0e12


def timed(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def make_inputs(B, H, W, P, rho, dev, seed=0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    U = torch.randn(B, H, W, 3, generator=g).to(dev)
    x0 = torch.randint(rho, W - rho - P + 1, (B,), generator=g)
    y0 = torch.randint(rho, H - rho - P + 1, (B,), generator=g)
    pts1 = torch.stack([x0, y0, x0 + P, y0, x0 + P, y0 + P, x0, y0 + P], 1).float().to(dev)
    gt = torch.randint(-rho, rho + 1, (B, 8), generator=g).float()
    h4p = (gt + 2.0 * torch.randn(B, 8, generator=g)).to(dev)
    u = torch.arange(P)
    idx = ((u[None, :, None] + y0[:, None, None]) * W + (u[None, None, :] + x0[:, None, None])).reshape(B, P * P)
    I2 = torch.randn(B, P, P, 1, generator=g).to(dev)
    return U, pts1, h4p, idx.int().to(dev), I2


def run(B, H, W, P, rho, iters, dev):
    U, pts1, h4p, idx, I2 = make_inputs(B, H, W, P, rho, dev)
    _, theta = ops.solve_dlt(pts1, h4p, img_w=W, img_h=H)
    theta = theta.detach()
    if os.environ.get('UH_IDENTITY_THETA'):
        theta = torch.eye(3, device=dev).reshape(1, 9).repeat(B, 1).contiguous()
    kind = os.environ.get('UH_THETA_KIND')
    if kind:                                   # controlled footprints: which part of the law costs what
        import math
        def T9(m):
            return torch.tensor(m, dtype=torch.float32, device=dev).reshape(1, 9).repeat(B, 1).contiguous()
        rot = lambda deg: [[math.cos(math.radians(deg)), -math.sin(math.radians(deg)), 0],
                           [math.sin(math.radians(deg)), math.cos(math.radians(deg)), 0], [0, 0, 1]]
        theta = {'shift': T9([[1, 0, 0.0371], [0, 1, -0.0213], [0, 0, 1]]),
                 'rot5': T9(rot(5)), 'rot15': T9(rot(15)), 'rot45': T9(rot(45)),
                 'zoomin': T9([[0.8, 0, 0], [0, 0.8, 0], [0, 0, 1]]),
                 'zoomout': T9([[1.25, 0, 0], [0, 1.25, 0], [0, 0, 1]]),
                 'zoomout2': T9([[2.0, 0, 0], [0, 2.0, 0], [0, 0, 1]]),
                 'persp': T9([[1, 0, 0], [0, 1, 0], [0.15, 0.1, 1]])}[kind]
    dOut = torch.randn_like(U)
    bytes_fwd = 2 * B * H * W * 3 * 4
    res = {'B': B, 'H': H, 'W': W, 'alg_MB_fwd': bytes_fwd / 1e6}

    lib = _lib.load()
    import ctypes as C
    out = torch.empty_like(U)
    dT = torch.empty(B, 9, device=dev)
    nb = lib.uh_warp_backward_workspace_bytes(B, H, W, 3, H, W)
    ws = torch.empty(nb // 4, device=dev)
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())

    def fwd():
        lib.uh_warp_forward(p(U), p(theta), p(out), None, B, H, W, 3, H, W, st())

    def bwd():
        lib.uh_warp_backward(p(U), p(theta), p(dOut), p(dT), None, p(ws), nb, B, H, W, 3, H, W, st())

    def copy():
        out.copy_(U)

    pred = torch.empty(B, P * P, device=dev); loss = torch.empty(1, device=dev)
    nb2 = lib.uh_warp_patch_l1_workspace_bytes(B, P * P)
    ws2 = torch.empty(nb2 // 4, device=dev)
    I2f = I2.reshape(B, -1).contiguous()

    def fused():
        lib.uh_warp_patch_l1_fwdbwd(p(U), p(theta), p(I2f), p(idx), p(pred), p(loss), p(dT), p(ws2), nb2,
                                    B, H, W, 3, P * P, st())

    for name, fn, nbytes in (('copy', copy, bytes_fwd), ('warp_fwd', fwd, bytes_fwd), ('warp_bwd', bwd, bytes_fwd),
                             ('patch_fused', fused, B * P * P * (48 + 12))):
        t = timed(fn, iters)
        res[name + '_us'] = round(t * 1e6, 2)
        res[name + '_GBs'] = round(nbytes / t / 1e9, 1)
        res[name + '_frac'] = round(nbytes / t / PEAK, 3)
    # in-library profiler view (per-kernel, events on the launch stream)
    _lib.profile_enable(True)
    for _ in range(iters):
        fwd(); bwd(); fused()
    prof = _lib.profile_read()
    _lib.profile_enable(False)
    res['lib_prof_us'] = {k: round(v[0] / max(v[1], 1) * 1e3, 2) for k, v in prof.items() if v[1]}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=50)
    ap.add_argument('--configs', default='64,240,320,128,45;128,240,320,128,45;128,480,640,128,64',
                    help='semicolon-separated B,H,W,P,rho')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    for cfg in args.configs.split(';'):
        B, H, W, P, rho = (int(v) for v in cfg.split(','))
        r = run(B, H, W, P, rho, args.iters, dev)
        r['lib'] = os.path.basename(_lib.LIB_PATH); r['identity'] = bool(os.environ.get('UH_IDENTITY_THETA')); r['theta_kind'] = os.environ.get('UH_THETA_KIND', '')
        print(json.dumps(r), flush=True)


if __name__ == '__main__':
    main()
