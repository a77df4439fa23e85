#!/usr/bin/env python
"""Aggregate rocprofv3 counter_collection CSVs -> mean counter value per kernel (uh:: kernels + the torch
calibration copy).  usage: pmc_summarize.py <dir> [out.json]"""
import csv, glob, json, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
names = set()
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '')
        names.add(k[:120])
        if 'uh::' in k:
            k = re.sub(r'\(.*', '', k).replace('void ', '')
        elif 'copyBuffer' in k or ('copy' in k.lower() and 'elementwise' in k.lower()):
            k = 'calibration_copy'
        else:
            continue
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
res = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
for k in sorted(res):
    print(k)
    for c in sorted(res[k]):
        print('   %-40s mean %.6g  (n=%d)' % (c, res[k][c], len(acc[k][c])))
print('all kernel names seen:')
for n in sorted(names):
    print('  ', n)
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], 'w'), indent=1, sort_keys=True)
