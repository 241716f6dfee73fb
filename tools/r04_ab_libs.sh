#!/bin/bash
# usage (GPU box): bash tools/r04_ab_libs.sh "<variant names>" [rounds]  -- config 5 on one GPU and on 8 local shards per variant library, alternating
cd $GRAFT_REPO_ROOT
d=tscm_calib_amd/csrc
cp $d/libtscm_hip.so /tmp/rel.so
trap 'cp /tmp/rel.so $d/libtscm_hip.so' EXIT
for r in $(seq ${2:-2}); do for v in $1; do
  cp $d/variants/lib$v.so $d/libtscm_hip.so
  a=$(python3 bench.py --config 5 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f us/step' % (1e3*d['ms_per_step']))")
  b=$(python3 tools/bench_shards.py --config 5 --worlds 8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f us/rank' % d['per_world']['8']['rank_us_per_iteration'])")
  c=$(python3 tools/bench_shards.py --config 4 --worlds 8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f us/rank' % d['per_world']['8']['rank_us_per_iteration'])")
  echo "$v round $r: config 5 one GPU $a | config 5 on 8 shards $b | config 4 on 8 shards $c"
done; done
