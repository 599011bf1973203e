#!/usr/bin/env python
"""Print the GPU timeline of the last steps of a rocprofv3 kernel trace
(gpurun_out/prof_<workload>/trace): per kernel start offset, duration and the
idle gap before it."""
import csv, glob, sys
w = sys.argv[1]
f = glob.glob('gpurun_out/prof_%s/trace/**/*kernel_trace.csv' % w, recursive=True)[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:40])
               for r in csv.DictReader(open(f))))
tail = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -12:]
prev = None
for s, e, n in tail:
    print('%-40s start +%8.2f us  dur %7.2f us  gap %6.2f us' % (
        n, (s - tail[0][0]) / 1e3, (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3))
    prev = e
