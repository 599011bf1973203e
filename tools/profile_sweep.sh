#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace of Rao-Teh sweeps
# (tools/bench_sweep.py).  usage: tools/profile_sweep.sh c2 10000
set -o pipefail
W=${1:-c2}
N=${2:-10000}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_sweep_$W
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- \
    python3 $REPO/tools/bench_sweep.py $W $N 10 ${3:-device} > $OUT/run.log 2> $OUT/trace.err || exit 1
cd $REPO
cat $OUT/run.log
cut -c1-160 $OUT/trace/trace_kernel_stats.csv | head -12
