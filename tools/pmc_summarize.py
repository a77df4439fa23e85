#!/usr/bin/env python
"""Aggregate rocprofv3 counter_collection CSVs: mean counter value per kernel (uh:: kernels only)."""
import csv, glob, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '')
        if 'uh::' not in k: continue
        k = re.sub(r'\(.*', '', k).replace('void ', '')
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]; print('   %-40s mean %.4g  (n=%d)' % (c, sum(v) / len(v), len(v)))
