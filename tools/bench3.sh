#!/bin/bash
# pytest -m gpu (stop at first failure) then bench the three workloads (run through gpurun)
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -m gpu -q --timeout=300 -x > gpurun_out/pytest.log 2>&1; rc=$?
echo pytest_rc=$rc; tail -${TAILN:-5} gpurun_out/pytest.log
[ $rc -eq 0 ] || exit $rc
for w in ${WORKLOADS:-c2 c3 c5}; do
  python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline --also '' > gpurun_out/b_$w.json 2>gpurun_out/b_$w.err || { tail -3 gpurun_out/b_$w.err; continue; }
  python - <<PY
import json
d=json.load(open("gpurun_out/b_$w.json"))
print("$w", d["roofline"]["kernel"], "prune_us=%.1f frac=%.3f step_us=%.1f value=%.4g" % (d["roofline"]["avg_kernel_us"], d["roofline"]["frac"], d["ms_per_step"]*1e3, d["value"]), {k:(round(v,1) if isinstance(v,float) else v) for k,v in d["kernels_us"].items()})
PY
done
