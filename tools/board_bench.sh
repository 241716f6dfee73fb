#!/bin/bash
# usage (GPU box): bash tools/board_bench.sh <tag> "<CxR> <CxR> ..." [bench args]
# The dominant kernel on boards other than BASELINE's 9x6 (round 6: k_eval_gram4<KS, MULTI> serves every board size): per
# board one rocprofv3 kernel trace of bench.py --board CxR (tools/prof.sh: medians over several hundred launches) and the
# bench line itself -> gpurun_out/<tag>_board_<CxR>_{bench.json,kernel_medians.csv}; one summary line per board.
tag=${1:?tag}; boards=${2:-"9x6 11x8 8x6 6x5"}; shift 2
o=$GRAFT_REPO_ROOT/gpurun_out
for b in $boards; do
  bash $GRAFT_REPO_ROOT/tools/prof.sh ${tag}_board_$b --board $b "$@" > $o/${tag}_board_${b}_prof.txt 2>&1
  grep '^{' $o/prof_${tag}_board_$b.log > $o/${tag}_board_${b}_bench.json
  python3 - $b $o/${tag}_board_${b}_bench.json $o/${tag}_board_${b}_kernel_medians.csv <<'PY'
import csv, json, sys
b, jf, mf = sys.argv[1:4]
d = json.load(open(jf)); r = d["roofline"]
n = r["alg_flop_per_launch"] / 1436.0
med = [(row["Name"], int(row["MedianNs"]), int(row["Calls"])) for row in csv.DictReader(open(mf)) if "k_eval_gram" in row["Name"]]
name, ns, calls = max(med, key=lambda x: x[2])
print(f"{b:6s} {int(n):8d} corners  {d['value']:8.0f} it/s  {1e3 * d['ms_per_step']:7.1f} us/step  {r['kernel']:28s} median {ns / 1e3:7.2f} us ({calls} launches)  "
      f"{1e3 * ns / n:6.2f} ps/corner  frac {r['alg_flop_per_launch'] / (ns * 1e-9) / 1e12 / r['peak']:.3f} (median)  {r['frac']:.3f} (HIP events)")
PY
done
