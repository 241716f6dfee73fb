#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/prof.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 10 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
grep '^{' $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('value', d['value'], 'ms/step', d['ms_per_step'], 'eval ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'])"
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])): print(f"{r['Name'][:40]:42s} {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}%")
PY
# medians from the trace (the means above include the early-exit launches after ctrl->done)
python3 $GRAFT_REPO_ROOT/tools/kernel_medians.py $GRAFT_REPO_ROOT/gpurun_out/prof_$tag $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_medians.csv
