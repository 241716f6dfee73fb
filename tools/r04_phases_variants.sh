#!/bin/bash
# usage (GPU box): bash tools/r04_phases_variants.sh "<variant names>" [config]  -- phase stamps of k_solve_nd for several -DTSCM_PHASE_PROFILE builds
cd $GRAFT_REPO_ROOT
d=tscm_calib_amd/csrc
cp $d/libtscm_hip.so /tmp/rel.so
trap 'cp /tmp/rel.so $d/libtscm_hip.so' EXIT
for v in $1; do
  cp $d/variants/lib$v.so $d/libtscm_hip.so
  echo "== $v config ${2:-5}"
  TSCM_BENCH_PREHEAT_MS=0 python3 bench.py --config ${2:-5} --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep -E "^solve_nd|^  phase |^  backsub" | tail -30 | awk '/^solve_nd/{print} /backsub/{print} /phase [0-9]/{printf "%s ", $3} END{print ""}'
done
