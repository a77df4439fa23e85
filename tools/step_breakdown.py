#!/usr/bin/env python
"""Steady-state per-step GPU time by kernel from a rocprofv3 --kernel-trace CSV of `bench.py --steps K --warmup W`: the K
TIMED steps (tools/_trace_steps.py delimits them), so neither the MIOpen find-mode trial kernels of the warm-up nor the
untimed replays bench.py appends enter.  First line: `# _fingerprint: <sha256>` of the traced library's kernel sources.
usage: step_breakdown.py kernel_trace.csv W K [top]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _trace_steps import fingerprint, load_steps  # noqa: E402

rows, spans = load_steps(sys.argv[1])
W, K = int(sys.argv[2]), int(sys.argv[3])
top = int(sys.argv[4]) if len(sys.argv) > 4 else 40
if len(spans) < W + K or K < 2:
    sys.exit('only %d steps found, need warm-up %d + timed %d (>= 2)' % (len(spans), W, K))
sel = spans[W:W + K]
acc = collections.defaultdict(lambda: [0.0, 0])
for a, b in sel:
    for r in rows[a:b]:
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        e = acc[r['Kernel_Name'][:120]]; e[0] += d; e[1] += 1
# wall per step between the first kernels of consecutive TIMED steps (the span after the last timed step holds bench.py's
# synchronize + read-back, not a step)
span = (int(rows[sel[-1][0]]['Start_Timestamp']) - int(rows[sel[0][0]]['Start_Timestamp'])) / 1e3 / (K - 1)
busy = sum(v[0] for v in acc.values()) / K
print('# _fingerprint: %s  (kernel sources of the traced library; W=%d K=%d)' % (fingerprint(), W, K))
print('steps averaged: %d (the timed region)   wall per step %.1f us   sum of kernel durations per step %.1f us' % (K, span, busy))
for name, (t, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:top]:
    print('%9.1f us/step %6.1f launches/step avg %8.1f us  %s' % (t / K, n / K, t / n, name))
