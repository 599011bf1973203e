#!/usr/bin/env python
"""Copy the judged parts of gpurun_out/prof_<w>/ (tools/profile.sh) into
profiles/ and write profiles/traffic_<w>.json (read by bench.py)."""
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
for w in ('c2', 'c3', 'c5', 'c6'):
    base = os.path.join('gpurun_out', 'prof_' + w)
    summ = os.path.join(base, 'summary.json')
    if not os.path.exists(summ):
        continue
    d = json.load(open(summ))
    shutil.copy(summ, 'profiles/%s_%s_summary.json' % (tag, w))
    ks = glob.glob(os.path.join(base, 'trace/**/*kernel_stats.csv'), recursive=True)
    if ks:
        shutil.copy(ks[0], 'profiles/%s_%s_kernel_stats.csv' % (tag, w))
    bt = os.path.join(base, 'bench_trace.json')
    if os.path.exists(bt):
        shutil.copy(bt, 'profiles/%s_%s_bench_under_rocprof.json' % (tag, w))
    if 'prune_kernel' in d:
        json.dump({
            'workload': w, 'kernel': d['prune_kernel'],
            'fetch_size_raw_bytes': d['fetch_bytes_raw'], 'write_size_bytes': d['write_bytes'],
            'hbm_bytes_per_launch': d['hbm_bytes_per_launch'],
            'method': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; '
                      'FETCH_SIZE x2 (gfx950 16 B/lane streaming reads, MI355X_MICROARCH.md HBM '
                      'section); per-launch average',
            'source': 'profiles/%s_%s_summary.json' % (tag, w)},
            open('profiles/traffic_%s.json' % w, 'w'), indent=1)
        print(w, d['prune_kernel'][:40], 'HBM bytes/launch %.1f MB' % (d['hbm_bytes_per_launch'] / 1e6),
              [(r['kernel'][:24], round(r['avg_us'], 1)) for r in d['kernel_trace'][:4]])
