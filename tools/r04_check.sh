#!/bin/bash
# usage (GPU box): bash tools/r04_check.sh [tag] [pytest args]  -- GPU tests, then the bench lines the round is judged on
cd $GRAFT_REPO_ROOT
tag=${1:-r04}; shift
o=gpurun_out
timeout 1500 python3 -m pytest tests -m gpu -x -q "$@" > $o/${tag}_pytest.txt 2>&1; echo "pytest rc=$?" >> $o/${tag}_pytest.txt
tail -5 $o/${tag}_pytest.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $o/${tag}_driver.json 2> $o/${tag}_driver.err
python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > $o/${tag}_100.json 2> /dev/null
python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --exec-flags 32 > $o/${tag}_100_dense.json 2> /dev/null
python3 bench.py --config 5 --steps 100 --warmup 10 --no-cpu-baseline > $o/${tag}_c5.json 2> /dev/null
python3 bench.py --config 5 --steps 100 --warmup 10 --no-cpu-baseline --exec-flags 32 > $o/${tag}_c5_dense.json 2> /dev/null
python3 tools/bench_shards.py --config 4 --worlds 1,8 > $o/${tag}_shards4.json 2> /dev/null
python3 tools/bench_shards.py --config 5 --worlds 1,8 > $o/${tag}_shards5.json 2> /dev/null
for f in driver 100 100_dense c5 c5_dense; do grep '^{' $o/${tag}_$f.json | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('${tag}_$f: steps %d %.0f it/s  %.1f us/step  eval %.2f us (%d timed) frac %.3f' % (d['steps'], d['value'], 1e3*d['ms_per_step'], 1e3*r['avg_launch_ms'], r['launches'], r['frac']))"; done
cat $o/${tag}_shards4.json $o/${tag}_shards5.json
