#!/usr/bin/env python
"""Steady-state per-step GPU time by kernel from a rocprofv3 --kernel-trace CSV of bench.py.
Steps are delimited by uh::dlt_forward_kernel<float> (once per step); the last K full steps are averaged, so the
MIOpen find-mode trial kernels of the warm-up never enter.  usage: step_breakdown.py kernel_trace.csv [K] [top]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'dlt_forward_kernel<float>' in r['Kernel_Name']]
lo, hi = marks[-K - 1], marks[-1]
sel = rows[lo:hi]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in sel:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    a = acc[r['Kernel_Name'][:120]]; a[0] += d; a[1] += 1
span = (int(rows[hi]['Start_Timestamp']) - int(rows[lo]['Start_Timestamp'])) / 1e3 / K
busy = sum(v[0] for v in acc.values()) / K
print('steps averaged: %d   wall per step %.1f us   sum of kernel durations per step %.1f us' % (K, span, busy))
for name, (t, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:top]:
    print('%9.1f us/step %6.1f launches/step avg %8.1f us  %s' % (t / K, n / K, t / n, name))
