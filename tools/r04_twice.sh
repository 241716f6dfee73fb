#!/bin/bash
# experiment (GPU box): k_solve_nd launched twice in a row per iteration (variants/libTW.so): durations of the odd / even launches from the kernel trace
cd $GRAFT_REPO_ROOT
d=tscm_calib_amd/csrc
cp $d/libtscm_hip.so /tmp/rel.so
trap 'cp /tmp/rel.so $d/libtscm_hip.so' EXIT
cp $d/variants/libTW.so $d/libtscm_hip.so
bash tools/prof.sh r04_twice --config 5 > gpurun_out/r04_twice_prof.txt 2>&1
python3 - <<'PY'
import csv,glob,statistics
f=sorted(glob.glob("gpurun_out/prof_r04_twice/**/*kernel_trace.csv", recursive=True))[-1]
rows=[r for r in csv.DictReader(open(f)) if "k_solve_nd" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
d=[int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows]
d=[x for x in d if x>20000]
print("launches", len(d), "first of a pair median %.1f us, second %.1f us" % (statistics.median(d[0::2])/1e3, statistics.median(d[1::2])/1e3))
PY
