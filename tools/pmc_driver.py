#!/usr/bin/env python
"""Minimal launch loop for rocprofv3 passes: N x (calibration copy, warp_fwd, warp_bwd[, fused patch]) at one
config.  The calibration copy (torch `out.copy_(U)`: a 16 B/lane streaming kernel moving exactly the bytes
warp_fwd's signature moves) gives the gfx950 correction factor of FETCH_SIZE / WRITE_SIZE for this run
(MI355X_MICROARCH.md, "HBM": FETCH_SIZE under-reports wide streaming reads by 2x; calibrate in your own run)."""
import argparse, ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsuperviseddeephomographyral2018_amd import ops, _lib
from tools.microbench import make_inputs

ap = argparse.ArgumentParser()
ap.add_argument('--cfg', default='128,240,320,128,45'); ap.add_argument('--iters', type=int, default=10)
ap.add_argument('--fused', type=int, default=1)
a = ap.parse_args()
B, H, W, P, rho = (int(v) for v in a.cfg.split(','))
dev = torch.device('cuda:0')
U, pts1, h4p, idx, I2 = make_inputs(B, H, W, P, rho, dev)
_, theta = ops.solve_dlt(pts1, h4p, img_w=W, img_h=H)
lib = _lib.load(); p = lambda t: C.c_void_p(t.data_ptr())
out = torch.empty_like(U); dOut = torch.randn_like(U); dT = torch.empty(B, 9, device=dev)
nb = lib.uh_warp_backward_workspace_bytes(B, H, W, 3, H, W); ws = torch.empty(nb // 4, device=dev)
pred = torch.empty(B, P * P, device=dev); loss = torch.empty(1, device=dev)
nb2 = lib.uh_warp_patch_l1_workspace_bytes(B, P * P); ws2 = torch.empty(nb2 // 4, device=dev)
I2f = I2.reshape(B, -1).contiguous()
torch.cuda.synchronize()
for _ in range(a.iters):
    out.copy_(U)
    lib.uh_warp_forward(p(U), p(theta), p(out), None, B, H, W, 3, H, W, None)
    lib.uh_warp_backward(p(U), p(theta), p(dOut), p(dT), None, p(ws), nb, B, H, W, 3, H, W, None)
    if a.fused:
        lib.uh_warp_patch_l1_fwdbwd(p(U), p(theta), p(I2f), p(idx), p(pred), p(loss), p(dT), p(ws2), nb2, B, H, W, 3, P * P, None)
torch.cuda.synchronize()
