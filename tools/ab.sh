#!/bin/bash
# usage (GPU box): bash tools/ab.sh <kernel-name-fragment> [rounds]   -- alternates csrc/libA.so and csrc/libB.so under tools/prof.sh
frag=$1; n=${2:-2}
d=$GRAFT_REPO_ROOT/tscm_calib_amd/csrc
for r in $(seq $n); do
  for v in A B; do
    cp $d/lib$v.so $d/libtscm_hip.so; rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_ab_$v
    echo "== $v"; bash $GRAFT_REPO_ROOT/tools/prof.sh ab_$v 2>&1 | grep "value\|$frag"
  done
done
