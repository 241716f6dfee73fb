#!/bin/bash
# usage (GPU box): bash tools/pmc.sh <tag> "<counters>" [bench args]   -- one --pmc pass of bench.py (short)
tag=$1; ctrs=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 0 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'][:28]; acc[k][r['Counter_Name']]+=float(r['Counter_Value']); 
    if r['Counter_Name']==list(acc[k].keys())[0]: n[k]+=1
for k in acc:
    print(k, 'dispatches', n[k], ' '.join(f"{c}={v/max(n[k],1):.4g}" for c,v in acc[k].items()))
PY
