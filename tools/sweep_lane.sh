#!/bin/bash
# A/B sweep of the lane-kernel variants on the C2 workload (run through gpurun).
mkdir -p gpurun_out
for cfg in "reg 8" "reg 4" "dma 2" "dma 3" "dma 4" "dma 5"; do
  set -- $cfg; v=$1; r=$2
  RAOTEH_JIT=0 RAOTEH_LANE_VARIANT=$v RAOTEH_LANE_RING=$r timeout -k 10 120 python bench.py --workload c2 --steps 100 --warmup 10 --no-cpu-baseline --also '' > gpurun_out/sweep_${v}_$r.json 2> gpurun_out/sweep_${v}_$r.err || { echo FAIL $v $r; tail -3 gpurun_out/sweep_${v}_$r.err; continue; }
  python - <<PY
import json
d=json.load(open('gpurun_out/sweep_${v}_$r.json'))
print('$v R=$r', d['roofline']['kernel'], 'prune_us=%.1f frac=%.3f step_us=%.1f value=%.3g' % (d['roofline']['avg_kernel_us'], d['roofline']['frac'], d['ms_per_step']*1e3, d['value']))
PY
done
