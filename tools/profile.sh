#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace stats + HBM PMC
# passes for one bench workload.  Outputs under gpurun_out/prof_<workload>/.
# usage: tools/profile.sh c2 [steps]
set -o pipefail
W=${1:-c2}
STEPS=${2:-50}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$W
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- \
    python3 $REPO/bench.py --workload $W --steps $STEPS --warmup 5 --no-cpu-baseline --also '' \
    > $OUT/bench_trace.json 2> $OUT/trace.err || exit 1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o pmc -- \
      python3 $REPO/bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline --also '' \
      > $OUT/bench_pmc_$C.json 2> $OUT/pmc_$C.err || exit 1
done
cd $REPO
python3 tools/summarize_prof.py $W
