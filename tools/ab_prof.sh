#!/bin/bash
# usage (GPU box): bash tools/ab_prof.sh "<name> <name> ..." [rounds] [bench args]
# alternates the experiment builds csrc/variants/lib<name>.so (make variant VARIANT=<name> EXTRA=...) under rocprofv3
# (tools/prof.sh: kernel trace of bench.py) and prints, per build and round, the LM rate and the MEDIAN duration of every
# kernel of the solver over the several hundred launches of the run -- the HIP-event figure of tools/variants.sh averages
# 13 launches and cannot tell builds apart that differ by less than a microsecond.  Restores the release build at the end.
names=$1; n=${2:-2}; shift; [ $# -gt 0 ] && shift
d=$GRAFT_REPO_ROOT/tscm_calib_amd/csrc
cp $d/libtscm_hip.so /tmp/libtscm_release.so
trap 'cp /tmp/libtscm_release.so $d/libtscm_hip.so' EXIT
for r in $(seq $n); do
  for v in $names; do
    cp $d/variants/lib$v.so $d/libtscm_hip.so
    rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_ab_$v
    bash $GRAFT_REPO_ROOT/tools/prof.sh ab_$v "$@" > /tmp/ab_prof.log 2>&1
    python3 - "$v" "$r" $GRAFT_REPO_ROOT/gpurun_out/ab_${v}_kernel_medians.csv /tmp/ab_prof.log <<'PY'
import csv, re, sys
v, r, f, log = sys.argv[1:5]
m = re.search(r"value ([0-9.]+)", open(log).read())
ks = [(row["Name"], int(row["Calls"]), int(row["MedianNs"])) for row in csv.DictReader(open(f)) if "tscm::" in row["Name"] and int(row["Calls"]) >= 100]
short = lambda n: re.sub(r"\(.*", "", n.replace("void ", "").replace("tscm::", ""))
print(f"{v} round {r}: {float(m.group(1)) if m else 0:.0f} it/s  " + "  ".join(f"{short(n)} {ns / 1e3:.2f}" for n, c, ns in ks))
PY
  done
done
