#!/bin/bash
# usage (through gpurun): tools/sweep_c5_pd.sh -> gpurun_out/sweep_c5_pd.txt
# 4x4x4-block kernel of C5: tiles per wave x park delay x leaf prefetch distance
OUT=gpurun_out/sweep_c5_pd.txt
: > $OUT
run() {
  echo "== $*" >> $OUT
  env "$@" python bench.py --workload c5 --also '' --steps 40 --warmup 5 --no-cpu-baseline 2>>$OUT | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('   kernel %s avg %.1f us (n=%d) frac %.3f | step %.1f us expm %.1f us' % (r['kernel'], r['avg_kernel_us'], r['launches_timed'], r['frac'], d['ms_per_step']*1e3, d['kernels_us']['expm']))
" >> $OUT
}
for t in ${TILES:-2 4 1}; do
for pd in ${PDS:-0 1 2}; do
for d in ${DS:-2 3}; do
  run RAOTEH_JIT_QUAD=1 RAOTEH_JIT_TILES=$t RAOTEH_JIT_PARKDELAY=$pd RAOTEH_JIT_PREFETCH=$d
done
done
done
cat $OUT
