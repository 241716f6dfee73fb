#!/bin/bash
# usage (GPU box): bash tools/pmc_tool.sh <tag> "<counter>" <tools/script.py> [args]  -- one --pmc pass of a side bench
tag=$1; ctrs=$2; script=$3; shift 3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -- python3 $GRAFT_REPO_ROOT/$script --no-cpu "$@" > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'][:48]; acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']==list(acc[k].keys())[0]: n[k]+=1
for k in acc:
    print(k, 'dispatches', n[k], ' '.join(f"{c}={v/max(n[k],1):.6g}" for c,v in acc[k].items()))
PY
