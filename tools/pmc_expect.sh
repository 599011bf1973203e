#!/bin/bash
# usage (through gpurun): tools/pmc_expect.sh [c3] -> gpurun_out/pmc_expect_<tag>.txt
# SQ / L2 counters of the kernels of the resident expectation step, a few per pass
W=${1:-c3}
REPO=$(pwd)
for group in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum" \
             "FETCH_SIZE" "WRITE_SIZE" ; do
  tag=$(echo $group | cut -d' ' -f1)
  OUT=$REPO/gpurun_out/pmc_expect_$tag
  mkdir -p $OUT
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc $group --output-format csv -d $OUT -o pmc -- python3 $REPO/tools/time_expect_step.py $W 3 > $OUT/run.log 2> $OUT/err.log ) || { tail -3 $OUT/err.log; continue; }
  python3 - <<PY > gpurun_out/pmc_expect_$tag.txt
import csv, glob, collections
f = glob.glob('$OUT/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f)):
    acc[row['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:44]][row['Counter_Name']].append(float(row['Counter_Value']))
for k, d in acc.items():
    if any(w in k for w in ('expect_', 'prune_mfma', 'wide', 'frechet')):
        print(k)
        for c, v in sorted(d.items()):
            print('   %-28s avg %.5g  (n=%d)' % (c, sum(v)/len(v), len(v)))
PY
  cat gpurun_out/pmc_expect_$tag.txt
done
