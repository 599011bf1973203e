#!/bin/bash
# A/B: site blocks per wave of the LDS-DMA lane kernel
set -e
mkdir -p gpurun_out
RAOTEH_LANE_BLOCKS=2 timeout -k 10 400 python -m pytest tests -m gpu -x -q --timeout=300 > gpurun_out/ab_tests_b2.log 2>&1 || { tail -30 gpurun_out/ab_tests_b2.log; exit 1; }
tail -2 gpurun_out/ab_tests_b2.log
for cfg in "1 4" "2 4" "2 2"; do
  set -- $cfg
  for rep in 1 2; do
    RAOTEH_JIT=0 RAOTEH_LANE_BLOCKS=$1 RAOTEH_LANE_WPB=$2 python bench.py --workload c2 --steps 400 --warmup 40 --no-cpu-baseline --also '' 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('B=$1 W=$2', d['value'], d['ms_per_step'], d['kernels_us'], d['roofline']['frac'])"
  done
done
