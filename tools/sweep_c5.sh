#!/bin/bash
# usage (through gpurun): tools/sweep_c5.sh  -> gpurun_out/sweep_c5.txt
OUT=gpurun_out/sweep_c5.txt
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
for q in 1 0; do
for t in 1 2 3 4; do
  run RAOTEH_JIT_QUAD=$q RAOTEH_JIT_TILES=$t
  run RAOTEH_JIT_QUAD=$q RAOTEH_JIT_TILES=$t RAOTEH_JIT_PREFETCH=4
done
done
run RAOTEH_JIT_QUAD=1 RAOTEH_JIT_TILES=2 RAOTEH_JIT_PREFETCH=3 RAOTEH_JIT_LOOKAHEAD=2
run RAOTEH_JIT_QUAD=1 RAOTEH_JIT_TILES=1 RAOTEH_JIT_PREFETCH=6
cat $OUT
