#!/usr/bin/env python
"""Read-only / write-only / copy device bandwidth at the warp benchmark's sizes (developer baseline)."""
import json, sys, torch
def timed(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3
for (B, H, W) in ((128, 240, 320), (128, 480, 640)):
    x = torch.randn(B, H, W, 3, device='cuda'); y = torch.empty_like(x)
    n = x.numel() * 4
    r = {'B': B, 'H': H, 'W': W, 'MB': n / 1e6}
    t = timed(lambda: y.copy_(x)); r['copy_us'] = t * 1e6; r['copy_TBs'] = 2 * n / t / 1e12
    t = timed(lambda: y.fill_(1.0)); r['fill_us'] = t * 1e6; r['fill_TBs'] = n / t / 1e12
    t = timed(lambda: torch.sum(x)); r['sum_us'] = t * 1e6; r['sum_TBs'] = n / t / 1e12
    t = timed(lambda: torch.add(x, 1.0, out=y)); r['add_us'] = t * 1e6; r['add_TBs'] = 2 * n / t / 1e12
    print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items()}))
