#!/usr/bin/env python
"""VERDICT r4 item 3, the bounded experiment: can a forward epilogue PASS be removed by PyTorch-ROCm's fused
aten::miopen_convolution_relu (MIOpen fusion plan: conv + bias + ReLU) on the regressor's five non-pooled convs (channels_last,
f32, batch 64)?  Per conv shape: the shipped route (F.conv2d without bias + uh_bias_relu_forward in place) against
miopen_convolution_relu, with cudnn.benchmark off (immediate mode) and on (find mode, what the trainer runs), bits compared,
torch-event timing of 50 back-to-back calls after 10 warm-up calls.  An exception is a result too: it is printed."""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsuperviseddeephomographyral2018_amd import dist as uh_dist, ops  # noqa: E402

uh_dist.skip_naive_conv_in_find()
dev = torch.device('cuda:0')
SHAPES = [(64, 2, 64, 128), (64, 64, 64, 64), (64, 64, 128, 32), (64, 128, 128, 16)]     # (N, Cin, Cout, side): convs 0, 2, 4, 6/7


def timed(fn, iters=50, warm=10):
    for _ in range(warm):
        y = fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        y = fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3, y


for bench_mode in (False, True):
    torch.backends.cudnn.benchmark = bench_mode
    for (N, Ci, Co, S) in SHAPES:
        g = torch.Generator(device=dev).manual_seed(0)
        x = torch.randn(N, Ci, S, S, generator=g, device=dev).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(Co, Ci, 3, 3, generator=g, device=dev) * 0.1).contiguous(memory_format=torch.channels_last)
        b = torch.randn(Co, generator=g, device=dev)
        row = {'cudnn_benchmark': bench_mode, 'shape': 'N%d %d->%d @%dx%d' % (N, Ci, Co, S, S)}
        with torch.no_grad():
            try:
                row['shipped_us'], y0 = timed(lambda: ops.conv_bias_relu(x, w, b, 1))
                row['conv_alone_us'], _ = timed(lambda: F.conv2d(x, w, None, 1, 1))
            except Exception as e:                          # noqa: BLE001
                row['shipped_error'] = '%s: %s' % (type(e).__name__, e); y0 = None
            for fmt in ('channels_last', 'contiguous'):
                xi = x if fmt == 'channels_last' else x.contiguous()
                wi = w if fmt == 'channels_last' else w.contiguous()
                try:
                    us, y1 = timed(lambda: torch.ops.aten.miopen_convolution_relu(xi, wi, b, [1, 1], [1, 1], [1, 1], 1))
                    row['miopen_convolution_relu_%s_us' % fmt] = us
                    if y0 is not None:
                        row['max_abs_diff_%s' % fmt] = float((y1 - y0).abs().max())
                except Exception as e:                      # noqa: BLE001
                    row['miopen_convolution_relu_%s_error' % fmt] = '%s: %s' % (type(e).__name__, str(e)[:120])
        print(json.dumps({k: (round(v, 2) if isinstance(v, float) and k.endswith('_us') else v) for k, v in row.items()}), flush=True)
