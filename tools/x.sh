set -e
timeout -k 10 600 python -m pytest tests -m gpu -x -q --timeout=300 > gpurun_out/ab_tests_jit.log 2>&1 || { tail -30 gpurun_out/ab_tests_jit.log; exit 1; }
tail -1 gpurun_out/ab_tests_jit.log
run() { python bench.py --workload c2 --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('$1', '%.4g' % d['value'], '%.2f us/step' % (d['ms_per_step']*1e3), d['roofline']['kernel'], 'prune %.1f us' % d['kernels_us']['prune'], 'frac %.3f' % d['roofline']['frac'])"; }
for S in 64 0 49 56 61; do
if [ $S = 0 ]; then run jit_auto; else RAOTEH_JIT_BLOCK_SITES=$S run jit_S$S; fi
if [ $S = 0 ]; then run jit_auto; else RAOTEH_JIT_BLOCK_SITES=$S run jit_S$S; fi
done
