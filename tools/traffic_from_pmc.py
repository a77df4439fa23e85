#!/usr/bin/env python
"""rocprofv3 counter_collection CSVs of `bench.py` (FETCH_SIZE pass + WRITE_SIZE pass) -> profiles/traffic_rNN.json.

HBM bytes per launch = FETCH_SIZE[KiB] * 1024 * 2  +  WRITE_SIZE[KiB] * 1024
  * x2 on FETCH_SIZE: gfx950 tallies a 128-byte fabric read request as 64 bytes (MI355X_MICROARCH.md, "HBM");
    re-calibrated in this repo on a device copy of known size (profiles/r01a_pmc_*: copy of 471.9 MB -> FETCH_SIZE
    230.4 MiB, WRITE_SIZE 471.9 MB; TCC_EA0_RDREQ x 128 B and TCC_EA0_WRREQ x 64 B give the same bytes).
usage: traffic_from_pmc.py <pmc_dir> <B> <H> <W> <out.json> <label> [provenance] [library fingerprint]"""
import csv, glob, json, re, sys, collections
d, B, H, W, out, label = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6]
prov = sys.argv[7] if len(sys.argv) > 7 else ''
fingerprint = sys.argv[8] if len(sys.argv) > 8 else ''
import os
name = os.path.basename(out)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '')
        if 'uh::' not in k:
            continue
        k = re.sub(r'[<(].*', '', k.replace('void ', '')).replace('uh::', '').replace('_kernel', '')
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
try:
    res = json.load(open(out))
except Exception:
    res = {}
for k, cs in acc.items():
    if 'FETCH_SIZE' not in cs or 'WRITE_SIZE' not in cs:
        continue
    # launches of the timed region share one shape; the first launches (synthetic-data generation) may differ: use the median
    fs = sorted(cs['FETCH_SIZE'])[len(cs['FETCH_SIZE']) // 2]
    ws = sorted(cs['WRITE_SIZE'])[len(cs['WRITE_SIZE']) // 2]
    res['%s_B%d_%dx%d' % (k, B, H, W)] = {
        'hbm_bytes_per_launch': int(fs * 1024 * 2 + ws * 1024), 'FETCH_SIZE_KiB_median': fs, 'WRITE_SIZE_KiB_median': ws,
        'launches_seen': len(cs['FETCH_SIZE']),
        'source': 'profiles/' + name + ' <- rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `%s`; '
                  'bytes = FETCH_SIZE*2 KiB + WRITE_SIZE KiB (gfx950 correction, calibrated on a device copy)' % label}
if prov:
    res['_provenance'] = prov
if fingerprint:
    res['_fingerprint'] = fingerprint       # build._fingerprint() of the kernel sources measured: bench.py flags a mismatch as stale
json.dump(res, open(out, 'w'), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
