#!/usr/bin/env python
"""Per-kernel statistics of the TIMED steps only, one block of rows per theta law, from a rocprofv3 --kernel-trace CSV of
`bench.py --steps K --warmup W` (the plain --stats CSV of that command is 98 % MIOpen find-mode trial kernels,
and its one warp_forward row averages the theta laws).

Steps W .. W+K-1 are the timed region (mid-training theta law); then comes the untimed replay bench.py appends: 2 + 8 steps
under the round-1 law (raw regressor).  The first output line is `# _fingerprint: <sha256>` = build._fingerprint() of the
kernel sources the traced library was built from (tests/test_bench_helpers.py checks it against the tree).

usage: timed_steps_stats.py kernel_trace.csv W K > profiles/rNN_bench_kernel_stats_timed_steps.csv
"""
import collections
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _trace_steps import fingerprint, load_steps  # noqa: E402

rows, spans = load_steps(sys.argv[1])
W, K = int(sys.argv[2]), int(sys.argv[3])
if len(spans) < W + K:
    sys.exit('only %d steps found, need warm-up %d + timed %d' % (len(spans), W, K))
blocks = [('timed steps (mid-training theta law: regressor + gt + N(0,2px))', spans[W:W + K])]
extra = spans[W + K:]
if len(extra) >= 10:
    blocks.append(('replay under the round-1 law (raw regressor, near identity), 8 steps', extra[2:10]))
print('# _fingerprint: %s  (kernel sources of the traced library; W=%d K=%d)' % (fingerprint(), W, K))
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
