#!/bin/bash
# round 4, first call: is Ceres / Eigen on the GPU box, and the numbers of the round-3 release build on this box
cd $GRAFT_REPO_ROOT
o=gpurun_out
{
  echo "== probe for Ceres / Eigen on the GPU box =="
  ls -d /usr/include/ceres /usr/include/eigen3 /usr/lib/*/libceres* /usr/lib/*/cmake/Ceres /usr/local/include/ceres /usr/local/lib/libceres* 2>&1
  dpkg -l 2>/dev/null | grep -i -E 'ceres|eigen' || echo "dpkg: no ceres / eigen package"
  python3 -c "import pyceres" 2>&1 | tail -1
  find / -xdev \( -iname '*ceres*' -o -iname 'Eigen' \) -not -path '/proc/*' 2>/dev/null | grep -v "$GRAFT_REPO_ROOT" | head -20
  echo "== end of probe =="
} > $o/r04_ceres_probe.txt 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $o/r04_base_driver.json 2> $o/r04_base_driver.err
python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > $o/r04_base_100.json 2> /dev/null
python3 bench.py --config 5 --steps 100 --warmup 10 --no-cpu-baseline > $o/r04_base_c5.json 2> /dev/null
python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --jacobian-fp32 > $o/r04_base_f32.json 2> /dev/null
python3 tools/bench_shards.py --config 4 --worlds 1,8 > $o/r04_base_shards4.json 2> /dev/null
python3 tools/bench_shards.py --config 5 --worlds 1,8 > $o/r04_base_shards5.json 2> /dev/null
for f in r04_base_driver r04_base_100 r04_base_c5 r04_base_f32; do grep '^{' $o/$f.json | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$f: steps %d %.0f it/s  %.1f us/step  eval %.2f us (%d timed) frac %.3f' % (d['steps'], d['value'], 1e3*d['ms_per_step'], 1e3*r['avg_launch_ms'], r['launches'], r['frac']))"; done
cat $o/r04_base_shards4.json $o/r04_base_shards5.json
cat $o/r04_ceres_probe.txt
