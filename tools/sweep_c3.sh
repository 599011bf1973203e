#!/bin/bash
# usage (through gpurun): tools/sweep_c3.sh  -> gpurun_out/sweep_c3.txt
# A/B of the split-M tree-specialised kernels on config 3 (10 000 codon sites) and on a
# config-4 shard (125 000 sites): serial / pipelined generator, tiles per workgroup, prefetch
OUT=gpurun_out/sweep_c3.txt
: > $OUT
run() {
  W=$1; shift
  echo "== $W $*" >> $OUT
  env "$@" python bench.py --workload $W --also '' --steps 30 --warmup 4 --no-cpu-baseline $EXTRA 2>>$OUT | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('   kernel %s avg %.1f us (n=%d) frac %.3f | step %.1f us expm %.1f us' % (r['kernel'], r['avg_kernel_us'], r['launches_timed'], r['frac'], d['ms_per_step']*1e3, d['kernels_us']['expm']))
" >> $OUT
}
EXTRA=""
for sp in serial pipelined; do
for t in 1 2 3; do
  run c3 RAOTEH_JIT_SPLIT=$sp RAOTEH_JIT_TILES=$t
done
done
run c3 RAOTEH_JIT_SPLIT=pipelined RAOTEH_JIT_TILES=3 RAOTEH_JIT_PREFETCH=2
run c3 RAOTEH_JIT_SPLIT=pipelined RAOTEH_JIT_TILES=3 RAOTEH_JIT_PREFETCH=3
EXTRA="--sites 125000"
for sp in serial pipelined; do
for t in 2 3; do
  run c4 RAOTEH_JIT_SPLIT=$sp RAOTEH_JIT_TILES=$t
done
done
run c4 RAOTEH_JIT_SPLIT=pipelined RAOTEH_JIT_TILES=1
cat $OUT
