#!/bin/bash
# A/B: tree-specialised (hiprtc) kernel against the interpreter kernel on C2
set -e
mkdir -p gpurun_out
RAOTEH_JIT=1 timeout -k 10 600 python -m pytest tests -m gpu -x -q --timeout=300 > gpurun_out/ab_tests_jit.log 2>&1 || { tail -30 gpurun_out/ab_tests_jit.log; exit 1; }
tail -2 gpurun_out/ab_tests_jit.log
run() {
  python bench.py --workload ${W:-c2} --steps 400 --warmup 40 --no-cpu-baseline --also '' 2>gpurun_out/ab_err.log | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('$1', '%.4g' % d['value'], '%.2f us/step' % (d['ms_per_step']*1e3), d['roofline']['kernel'], 'prune %.1f us' % d['kernels_us']['prune'], 'frac %.3f' % d['roofline']['frac'])" || tail -5 gpurun_out/ab_err.log
}
RAOTEH_JIT=0 run interp
RAOTEH_JIT=0 run interp
for LA in ${LAS:-1 2}; do
for D in ${DS:-4 6 8}; do
  RAOTEH_JIT=1 RAOTEH_JIT_PREFETCH=$D RAOTEH_JIT_LOOKAHEAD=$LA run jit_LA${LA}_D$D
  RAOTEH_JIT=1 RAOTEH_JIT_PREFETCH=$D RAOTEH_JIT_LOOKAHEAD=$LA run jit_LA${LA}_D$D
done
done
