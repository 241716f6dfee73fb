#!/bin/bash
# usage (GPU box): bash tools/variants.sh "<name> <name> ..." [rounds] [bench args]
# alternates the experiment builds csrc/variants/lib<name>.so (make variant VARIANT=<name> EXTRA=...) under bench.py
# (no profiler: wall-clock rate and the HIP-event time of the dominant kernel); restores the release build at the end
names=$1; n=${2:-2}; shift 2
d=$GRAFT_REPO_ROOT/tscm_calib_amd/csrc
cp $d/libtscm_hip.so /tmp/libtscm_release.so
trap 'cp /tmp/libtscm_release.so $d/libtscm_hip.so' EXIT      # an interrupted run must not leave an experiment build as the release library
for r in $(seq $n); do
  for v in $names; do
    cp $d/variants/lib$v.so $d/libtscm_hip.so
    python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline "$@" 2> /tmp/variant_err.log | grep '^{' | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v round $r: %.0f it/s  %.1f us/step  eval %.2f us (%d launches timed)  frac %.3f' % (d['value'], 1e3*d['ms_per_step'], 1e3*r['avg_launch_ms'], r['launches'], r['frac']))" || tail -3 /tmp/variant_err.log
  done
done
