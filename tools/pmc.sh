#!/bin/bash
# usage: tools/pmc.sh <workload> <tag> <counter> [<counter> ...]   (run through gpurun)
W=$1; TAG=$2; shift 2
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_${W}_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --output-format csv -d $OUT -o pmc -- python3 $REPO/bench.py --workload $W --steps 8 --warmup 2 --no-cpu-baseline --also '' > $OUT/bench.json 2> $OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
cd $REPO
python3 - <<PY
import csv, glob, collections
f = glob.glob('$OUT/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f)):
    acc[row['Kernel_Name'][:60]][row['Counter_Name']].append(float(row['Counter_Value']))
for k, d in acc.items():
    if 'prune' in k or 'expm' in k:
        print(k)
        for c, v in sorted(d.items()):
            print('   %-28s avg %.4g  (n=%d)' % (c, sum(v)/len(v), len(v)))
PY
