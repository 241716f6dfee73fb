# usage (GPU box): bash tools/round_end_bench_lines.sh -- the default and the driver-command bench lines (with CPU baseline) into gpurun_out/
cd $GRAFT_REPO_ROOT
o=gpurun_out
python3 bench.py > $o/r03f_bench.json 2> $o/r03f_bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $o/r03f_bench_driver.json 2> /dev/null
for f in r03f_bench r03f_bench_driver; do grep '^{' $o/$f.json | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$f: steps %d %.0f it/s  %.1f us/step  eval %.2f us (%d timed) frac %.3f traffic %s cpu %s' % (d['steps'], d['value'], 1e3*d['ms_per_step'], 1e3*r['avg_launch_ms'], r['launches'], r['frac'], r.get('traffic'), d.get('cpu_baseline',{}).get('value')))"; done
