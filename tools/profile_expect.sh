#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace of the expected
# history statistics path (tests/soak/expect_check.py).  Outputs under
# gpurun_out/prof_expect_<workload>/.   usage: tools/profile_expect.sh c2 20000
set -o pipefail
W=${1:-c2}
N=${2:-20000}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_expect_$W
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- \
    python3 $REPO/tests/soak/expect_check.py $W $N 1 > $OUT/run.log 2> $OUT/trace.err || exit 1
cd $REPO
python3 tools/summarize_prof.py expect_$W
