#!/usr/bin/env python
"""Per-kernel statistics of the TIMED steps only, one block of rows per theta law, from a rocprofv3 --kernel-trace CSV of
`bench.py --steps K --warmup W` (the plain --stats CSV of that command is 98 % MIOpen find-mode trial kernels,
and its one warp_forward row averages three theta laws).

A step is the span from one uh::dlt_forward_kernel<float> launch to the next that also holds a uh::dlt_backward launch
(the stand-alone DLT solves bench.py makes for its statistics hold none).  Steps W .. W+K-1 are the timed region (mid-training
theta law); the last 8 steps of the run are the "round-1 law" replay (raw regressor, near identity).  What follows the last
step (warm / cold replays of the forward) is reported as a third block.

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
blocks = [('timed steps (mid-training theta law: regressor + gt + N(0,2px))', spans[W:W + K])]
extra = spans[W + K:]
if len(extra) >= 8:
    blocks.append(('replay under the round-1 law (raw regressor, near identity), last 8 steps', extra[-8:]))
# a last block: whatever runs after the final step's span start + its own kernels -> only the warp forward replays matter
last_b = spans[-1][1]
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
