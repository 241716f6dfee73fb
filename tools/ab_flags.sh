#!/bin/bash
# usage (GPU box): bash tools/ab_flags.sh <config> "<exec_flags values>" [rounds]  -- bench.py --exec-flags A/B in one call (same box, alternating)
cd $GRAFT_REPO_ROOT
cfg=$1; flags=$2; n=${3:-2}
for r in $(seq $n); do for f in $flags; do
  python3 bench.py --config $cfg --steps 100 --warmup 10 --no-cpu-baseline --exec-flags $f 2>/dev/null | grep '^{' | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('config $cfg flags $f round $r: %.0f it/s  %.1f us/step  eval %.2f us' % (d['value'], 1e3*d['ms_per_step'], 1e3*r['avg_launch_ms']))"
done; done
