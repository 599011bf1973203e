#!/bin/bash
# usage (through gpurun): tools/sweep_c3.sh  -> gpurun_out/sweep_c3.txt
# A/B of the knobs of the split-M tree-specialised kernel on config 3 (10 000 codon sites)
OUT=gpurun_out/sweep_c3.txt
: > $OUT
run() {
  echo "== $*" >> $OUT
  env "$@" python bench.py --workload c3 --also '' --steps 40 --warmup 5 --no-cpu-baseline --also '' 2>>$OUT | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('   kernel %s avg %.1f us (n=%d) frac %.3f | step %.1f us expm %.1f us' % (r['kernel'], r['avg_kernel_us'], r['launches_timed'], r['frac'], d['ms_per_step']*1e3, d['kernels_us']['expm']))
" >> $OUT
}
run RAOTEH_JIT_TILES=1
run RAOTEH_JIT_TILES=1 RAOTEH_JIT_PREFETCH=2
run RAOTEH_JIT_TILES=1 RAOTEH_JIT_PREFETCH=3
run RAOTEH_JIT_TILES=1 RAOTEH_JIT_PREFETCH=2 RAOTEH_JIT_LOOKAHEAD=2
run RAOTEH_JIT_TILES=2
run RAOTEH_JIT_TILES=2 RAOTEH_JIT_PREFETCH=2
run RAOTEH_JIT_TILES=3
run RAOTEH_JIT_TILES=3 RAOTEH_JIT_PREFETCH=2
cat $OUT
