#!/bin/bash
# rocprofv3 kernel-trace + HBM PMC passes for the three bench workloads (run through gpurun)
for w in ${RAOTEH_PROF_WORKLOADS:-c2 c3 c5 c6}; do
  bash tools/profile.sh $w 60 > gpurun_out/prof_$w.log 2>&1 || { tail -5 gpurun_out/prof_$w.log; exit 1; }
  python tools/timeline.py $w 7
done
