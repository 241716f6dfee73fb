#!/bin/bash
# usage (GPU box): bash tools/prof_shards.sh <tag> <config> <worlds>  -- rocprofv3 kernel medians of tools/bench_shards.py (the communicator path)
cd $GRAFT_REPO_ROOT
tag=$1; cfg=$2; w=$3
bash tools/prof_tool.sh ${tag} tools/bench_shards.py --config $cfg --worlds $w > gpurun_out/${tag}_prof.txt 2>&1
python3 tools/kernel_medians.py gpurun_out/prof_${tag} gpurun_out/${tag}_kernel_medians.csv
python3 - gpurun_out/${tag}_kernel_medians.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])): print(f"{r['Name'][:70]:72s} {r['Calls']:>5s} median {int(r['MedianNs'])/1e3:8.1f} us mean {int(r['MeanNs'])/1e3:8.1f}")
PY
