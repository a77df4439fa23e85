#!/usr/bin/env python
"""Counters of the in-step forward COLD vs WARM (round 5, a footnote to the launch anatomy): address-translation traffic
(TCP_UTCL1_*) and the average latency of the vector L1's requests to L2 (TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ) for
uh_warp_forward at batch 64, 240x320, with a 1 GiB evicting copy before every launch (cold) or back to back (warm).

    python tools/pmc_cold_warm.py run   --cold 1      # the launch loop one rocprofv3 pass profiles
    python tools/pmc_cold_warm.py parse DIR           # -> one JSON line per (temperature, counter group)
    bash: see tools/gpu_session.sh stage `pmc_cold`"""
import csv
import ctypes as C
import glob
import json
import os
import sys

if sys.argv[1] == 'run':
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from unsuperviseddeephomographyral2018_amd import _lib, ops
    from tools.microbench import make_inputs
    cold = int(sys.argv[3])
    dev = torch.device('cuda:0')
    B, H, W, P, rho = 64, 240, 320, 128, 45
    U, pts1, h4p, idx, I2 = make_inputs(B, H, W, P, rho, dev)
    _, theta = ops.solve_dlt(pts1, h4p, img_w=W, img_h=H); theta = theta.detach().contiguous()
    lib = _lib.load(); p = lambda t: C.c_void_p(t.data_ptr())
    out = torch.empty_like(U)
    ev_a = torch.empty((1 << 30) // 4, device=dev).normal_(); ev_b = torch.empty_like(ev_a)
    for _ in range(12):
        if cold:
            ev_b.copy_(ev_a)
        lib.uh_warp_forward(p(U), p(theta), p(out), None, B, H, W, 3, H, W, None)
    torch.cuda.synchronize()
else:
    root = sys.argv[2]
    for d in sorted(glob.glob(root + '/*')):
        acc = {}
        for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
            for r in csv.DictReader(open(f)):
                if 'warp_forward_kernel' in r.get('Kernel_Name', ''):
                    acc.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
        med = {k: sorted(v)[len(v) // 2] for k, v in acc.items()}
        row = {'pass': os.path.basename(d), 'launches': max((len(v) for v in acc.values()), default=0), 'median_per_launch': med}
        if 'TCP_TCC_READ_REQ_LATENCY_sum' in med and med.get('TCP_TCC_READ_REQ_sum'):
            row['avg_L2_read_request_latency_cycles'] = round(med['TCP_TCC_READ_REQ_LATENCY_sum'] / med['TCP_TCC_READ_REQ_sum'], 1)
        if 'TCP_UTCL1_TRANSLATION_MISS_sum' in med and med.get('TCP_UTCL1_REQUEST_sum'):
            row['utcl1_miss_per_request'] = round(med['TCP_UTCL1_TRANSLATION_MISS_sum'] / med['TCP_UTCL1_REQUEST_sum'], 5)
        print(json.dumps(row))
