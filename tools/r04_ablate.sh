#!/bin/bash
# usage (GPU box): bash tools/r04_ablate.sh "<variant names>" <kernel name substring> [bench args]
# ablation builds (make variant VARIANT=<name> EXTRA=-DTSCM_ABLATE=<bits>: results invalid, the solves end early) under the
# kernel trace: median duration of the launches of one kernel that did work (early exits after ctrl->done filtered out)
names=$1; kern=$2; shift 2
d=$GRAFT_REPO_ROOT/tscm_calib_amd/csrc
cp $d/libtscm_hip.so /tmp/libtscm_release.so
trap 'cp /tmp/libtscm_release.so $d/libtscm_hip.so' EXIT
cd /tmp && export TMPDIR=/tmp
for v in $names; do
  cp $d/variants/lib$v.so $d/libtscm_hip.so
  rm -rf /tmp/abl_$v
  rocprofv3 --kernel-trace --output-format csv -d /tmp/abl_$v -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 10 --no-cpu-baseline "$@" > /tmp/abl_$v.log 2>&1
  python3 - /tmp/abl_$v "$kern" $v <<'PY'
import csv, glob, os, statistics, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[-1]
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if sys.argv[2] in r["Kernel_Name"]]
w = [x for x in d if x > 12000]
print(f"{sys.argv[3]}: {sys.argv[2]} {len(w)} working launches of {len(d)}, median {statistics.median(w) / 1e3:.1f} us" if w else f"{sys.argv[3]}: no working launch of {len(d)}")
PY
done
