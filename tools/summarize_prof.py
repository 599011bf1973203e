#!/usr/bin/env python
"""Condense gpurun_out/prof_<workload>/ (rocprofv3 csv output) into
gpurun_out/prof_<workload>/summary.json: per-kernel average duration from the
kernel trace and HBM bytes per launch of the pruning kernel from the PMC passes
(FETCH_SIZE doubled for wide coalesced reads on gfx950, as
MI355X_MICROARCH.md section HBM prescribes; WRITE_SIZE as read)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

w = sys.argv[1]
base = os.path.join('gpurun_out', 'prof_' + w)
out = {'workload': w}


def find(pattern):
    hits = glob.glob(os.path.join(base, pattern), recursive=True)
    return hits[0] if hits else None


kt = find('trace/**/*kernel_trace.csv')
if kt:
    dur = defaultdict(list)
    for row in csv.DictReader(open(kt)):
        name = row.get('Kernel_Name') or row.get('kernel_name')
        s = int(row.get('Start_Timestamp') or row.get('start_timestamp'))
        e = int(row.get('End_Timestamp') or row.get('end_timestamp'))
        dur[name].append(e - s)
    rows = []
    for name, d in dur.items():
        d = sorted(d)
        rows.append(dict(kernel=name[:90], calls=len(d), avg_us=sum(d) / len(d) / 1e3,
                         median_us=d[len(d) // 2] / 1e3, min_us=d[0] / 1e3,
                         total_us=sum(d) / 1e3))
    rows.sort(key=lambda r: -r['total_us'])
    out['kernel_trace'] = rows

for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
    pc = find('pmc_%s/**/*counter_collection.csv' % counter)
    if not pc:
        continue
    vals = defaultdict(list)
    for row in csv.DictReader(open(pc)):
        name = row.get('Kernel_Name') or row.get('kernel_name')
        if (row.get('Counter_Name') or row.get('counter_name')) != counter:
            continue
        vals[name].append(float(row.get('Counter_Value') or row.get('counter_value')))
    out[counter] = dict((k[:90], dict(launches=len(v), avg_kb=sum(v) / len(v)))
                        for k, v in vals.items())

prune = None
for k in out.get('FETCH_SIZE', {}):
    if 'prune' in k and 'pack' not in k:
        prune = k
# the tree-specialised kernel is the measured path when the batch got one (the interpreter
# kernel's few launches in the same trace are bench.py's side measurement)
for k in out.get('FETCH_SIZE', {}):
    if k.startswith('rt_jit_prune'):
        prune = k
if prune:
    f = out['FETCH_SIZE'][prune]['avg_kb'] * 1024.0
    wr = out.get('WRITE_SIZE', {}).get(prune, {}).get('avg_kb', 0.0) * 1024.0
    out['prune_kernel'] = prune
    out['fetch_bytes_raw'] = f
    out['write_bytes'] = wr
    out['hbm_bytes_per_launch'] = 2.0 * f + wr
json.dump(out, open(os.path.join(base, 'summary.json'), 'w'), indent=1)
print(json.dumps(out, indent=1)[:3000])
