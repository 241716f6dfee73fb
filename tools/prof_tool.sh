#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/prof_tool.sh <tag> <tools/script.py> [args]
# rocprofv3 kernel trace + stats of one of the side benches; the summary lands in gpurun_out/<tag>_kernel_stats.csv
tag=$1; script=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python3 $GRAFT_REPO_ROOT/$script --no-cpu "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
grep '^{' $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])): print(f"{r['Name'][:60]:62s} {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}%")
PY
