set -e
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout=600 > gpurun_out/ab_tests_jit.log 2>&1 || { tail -30 gpurun_out/ab_tests_jit.log; exit 1; }
tail -1 gpurun_out/ab_tests_jit.log
run() { python bench.py --workload $2 $3 --steps 200 --warmup 20 --no-cpu-baseline 2>gpurun_out/ab_err.log | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('$1', '%.4g' % d['value'], '%.2f us/step' % (d['ms_per_step']*1e3), d['roofline']['kernel'], 'prune %.1f us' % d['kernels_us']['prune'], 'frac %.3f' % d['roofline']['frac'])" || tail -3 gpurun_out/ab_err.log; }
run c2 c2; run c2 c2; run c3 c3; run c5 c5; run c4shard c3 "--sites 125000"
