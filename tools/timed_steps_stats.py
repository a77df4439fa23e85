#!/usr/bin/env python
"""Per-kernel statistics of the TIMED steps only, one block of rows per theta law, from a rocprofv3 --kernel-trace CSV of
`bench.py --steps K --warmup W` (the plain --stats CSV of that command is 98 % MIOpen find-mode trial kernels,
and its one warp_forward row averages three theta laws).

A step is the span from one uh::dlt_forward_kernel<float> launch to the next that also holds a uh::dlt_backward launch
(the stand-alone DLT solves bench.py makes for its statistics hold none).  Steps W .. W+K-1 are the timed region (mid-training
theta law); then come the untimed replays bench.py appends: the round-1 law, and (round 4) the frame prefetch on / off.

usage: timed_steps_stats.py kernel_trace.csv W K > profiles/rNN_bench_kernel_stats_timed_steps.csv
"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
W, K = int(sys.argv[2]), int(sys.argv[3])
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'dlt_forward_kernel<float>' in r['Kernel_Name']]
spans = []
for a, b in zip(marks, marks[1:] + [len(rows)]):
    if any('dlt_backward_kernel' in rows[i]['Kernel_Name'] for i in range(a, b)):
        # the step ends with its last optimizer / library kernel: cut at the next mark
        spans.append((a, b))
if len(spans) < W + K:
    sys.exit('only %d steps found, need warm-up %d + timed %d' % (len(spans), W, K))
# What bench.py runs after the timed region (all untimed): 2 + 8 steps under the round-1 law (raw regressor), then -- round 4 --
# 3 + 10 + 20 steps with the frame prefetch on and 3 + 20 with it off again (mid-training law).
blocks = [('timed steps (mid-training theta law: regressor + gt + N(0,2px))', spans[W:W + K])]
extra = spans[W + K:]
if len(extra) >= 10:
    blocks.append(('replay under the round-1 law (raw regressor, near identity), 8 steps', extra[2:10]))
if len(extra) >= 10 + 33 + 23:
    blocks.append(('mid-training law WITH the frame prefetch (--prefetch_frame True), 30 steps', extra[13:43]))
    blocks.append(('mid-training law, prefetch off again, 20 steps', extra[46:66]))
out = csv.writer(sys.stdout)
out.writerow(['Block', 'Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'MinNs', 'MaxNs', 'UsPerStep', 'Steps'])
for label, sp in blocks:
    acc = collections.defaultdict(list)
    for a, b in sp:
        for r in rows[a:b]:
            acc[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    n = len(sp)
    for name, d in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        out.writerow([label, name[:160], len(d), sum(d), round(sum(d) / len(d), 1), min(d), max(d), round(sum(d) / n / 1e3, 2), n])
