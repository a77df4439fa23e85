#!/usr/bin/env python
"""Developer tool: per-wave phase timestamps of the forward warp (library built with -DUH_WARP_TRACE).
Prints, per path (A staged interior, B interior gather, C1 staged clipped, C2 clipped gather): share of waves, mean cycles
per phase, and the share of total wave-time -- i.e. which path the kernel's time goes to."""
import ctypes as C, os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsuperviseddeephomographyral2018_amd import ops, _lib
from tools.microbench import make_inputs
B, H, W, P, rho = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '128,480,640,128,64').split(','))
dev = torch.device('cuda:0')
U, pts1, h4p, idx, I2 = make_inputs(B, H, W, P, rho, dev)
_, theta = ops.solve_dlt(pts1, h4p, img_w=W, img_h=H); theta = theta.detach()
if os.environ.get('UH_IDENTITY_THETA'):
    theta = torch.eye(3, device=dev).reshape(1, 9).repeat(B, 1).contiguous()
lib = _lib.load()
nw = B * ((W + 63) // 64) * ((H + 15) // 16) * 4
tr = torch.zeros(nw * 16, dtype=torch.int64, device=dev)      # UH_TRACE_STRIDE
out = torch.empty_like(U)
p = lambda t: C.c_void_p(t.data_ptr())
BWD = bool(os.environ.get('UH_TRACE_BWD'))        # trace the backward (dense dOut) instead of the forward
if BWD:
    dOut = torch.randn_like(U); dT = torch.empty(B, 9, device=dev)
    nb = lib.uh_warp_backward_workspace_bytes(B, H, W, 3, H, W); ws = torch.empty(nb // 4, device=dev)
    run = lambda: lib.uh_warp_backward(p(U), p(theta), p(dOut), p(dT), None, p(ws), nb, B, H, W, 3, H, W, None)
else:
    run = lambda: lib.uh_warp_forward(p(U), p(theta), p(out), None, B, H, W, 3, H, W, None)
for _ in range(3):
    run()
torch.cuda.synchronize()
assert C.CDLL(_lib.LIB_PATH).uh_debug_set_trace(p(tr)) == 0
run()
torch.cuda.synchronize()
t = tr.cpu().numpy().reshape(nw, 16).astype(np.int64)
t = t[(t[:, 0] > 0) & (t[:, 7] > 0)]
span = t[:, 7].max() - t[:, 0].min()
life = t[:, 7] - t[:, 0]
res = {'kernel': 'backward' if BWD else 'forward', 'lib': os.path.basename(_lib.LIB_PATH), 'identity': bool(os.environ.get('UH_IDENTITY_THETA')), 'waves': int(len(t)), 'kernel_span_cycles': int(span), 'mean_wave_life': float(life.mean()),
       'avg_resident_waves': float(life.sum() / span)}
names = {0: 'A', 1: 'B', 2: 'C1', 3: 'C2'}
for k, n in names.items():
    m = t[:, 5] == k
    if not m.any():
        continue
    q = t[m]
    ph = {'coords+decide': (q[:, 1] - q[:, 0]).mean(), 'issue': (q[:, 2] - q[:, 1]).mean(), 'wait_data': (q[:, 3] - q[:, 2]).mean(),
          'consume+store_issue' if not BWD else 'accumulate': (q[:, 4] - q[:, 3]).mean(), 'store_drain' if not BWD else 'block_reduction': (q[:, 7] - q[:, 4]).mean()}
    res[n] = {'share_of_waves': round(float(m.mean()), 4), 'share_of_wave_time': round(float(life[m].sum() / life.sum()), 4),
              'mean_life': round(float(life[m].mean()), 1), 'p99_life': float(np.percentile(life[m], 99)),
              'phases': {a: round(float(b), 1) for a, b in ph.items()}, 'mean_dma_instr': round(float(q[:, 6].mean()), 2)}
# tail: when do the last waves of each path start / end relative to the kernel span
end = t[:, 7] - t[:, 0].min()
res['frac_of_span_after_which_95pct_waves_ended'] = round(float(np.percentile(end, 95) / span), 4)
print(json.dumps(res))
