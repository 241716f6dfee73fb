#!/bin/bash
# usage (GPU box): bash tools/r04_phases.sh  -- phase stamps of the fused kernels (variants/libPH.so: -DTSCM_PHASE_PROFILE) at configs 4 and 5
cd $GRAFT_REPO_ROOT
d=tscm_calib_amd/csrc
cp $d/libtscm_hip.so /tmp/rel.so
trap 'cp /tmp/rel.so $d/libtscm_hip.so' EXIT
cp $d/variants/libPH.so $d/libtscm_hip.so
for c in ${CONFIGS:-4 5}; do
  TSCM_BENCH_PREHEAT_MS=0 python3 bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep -E "solve_nd|  phase |schur_gram wg|backsub_prep wg" | tail -${TAILN:-40} > gpurun_out/r04_phases_c$c.txt
  echo "== config $c"; cat gpurun_out/r04_phases_c$c.txt
done
