#!/bin/bash
# usage (GPU box, via gpurun): bash tools/r04_round_end.sh [part]  -- the round's evidence of the release build into gpurun_out/r04f_*
#   part 1: kernel stats + medians (configs 4, 5, fp32), PMC passes (configs 4 and 5), shards, timelines
#   part 2 (after `python tools/record_pmc.py ...` here): the bench lines that carry roofline.traffic
cd $GRAFT_REPO_ROOT
d=tscm_calib_amd/csrc
o=gpurun_out
line() { for f in "$@"; do grep '^{' $o/$f.json | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$f: steps %d %.0f it/s  %.1f us/step  %s %.2f us (%d timed) frac %.3f of-ceiling %s traffic %s cpu %s' % (d['steps'], d['value'], 1e3*d['ms_per_step'], r['kernel'], 1e3*r['avg_launch_ms'], r['launches'], r['frac'], r.get('frac_of_measured_ceiling'), r.get('traffic'), (d.get('cpu_baseline') or {}).get('value')))"; done; }
if [ "${1:-1}" = "1" ]; then
  bash tools/prof.sh r04f > $o/r04f_prof.txt 2>&1
  bash tools/prof.sh r04f_c5 --config 5 > $o/r04f_c5_prof.txt 2>&1
  bash tools/prof.sh r04f_f32 --jacobian-fp32 > $o/r04f_f32_prof.txt 2>&1
  bash tools/prof.sh r04f_c5_f32 --config 5 --jacobian-fp32 > $o/r04f_c5_f32_prof.txt 2>&1
  bash tools/pmc.sh r04f_fetch FETCH_SIZE > $o/pmc_r04f_fetch.txt 2>&1
  bash tools/pmc.sh r04f_write WRITE_SIZE > $o/pmc_r04f_write.txt 2>&1
  bash tools/pmc.sh r04f_sq1 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" > $o/pmc_r04f_sq1.txt 2>&1
  bash tools/pmc.sh r04f_sq2 "SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" > $o/pmc_r04f_sq2.txt 2>&1
  bash tools/pmc.sh r04f_sq3 "GRBM_GUI_ACTIVE" > $o/pmc_r04f_sq3.txt 2>&1
  bash tools/pmc.sh r04f_c5_fetch FETCH_SIZE --config 5 > $o/pmc_r04f_c5_fetch.txt 2>&1
  bash tools/pmc.sh r04f_c5_write WRITE_SIZE --config 5 > $o/pmc_r04f_c5_write.txt 2>&1
  bash tools/r04_prof_shards.sh r04f_shards8_config4 4 8 > /dev/null 2>&1
  bash tools/r04_prof_shards.sh r04f_shards8_config5 5 8 > /dev/null 2>&1
  python3 tools/bench_shards.py --config 4 --worlds 1,2,4,8 > $o/r04f_shards_config4_bench.json 2> /dev/null
  python3 tools/bench_shards.py --config 5 --worlds 1,8 > $o/r04f_shards_config5_bench.json 2> /dev/null
  if [ -f $d/variants/libT.so ]; then
    cp $d/libtscm_hip.so /tmp/rel.so
    cp $d/variants/libT.so $d/libtscm_hip.so
    python3 tools/kernel_timeline.py --config 4 > $o/r04f_kernel_timeline.txt 2>&1
    python3 tools/kernel_timeline.py --config 5 > $o/r04f_c5_kernel_timeline.txt 2>&1
    python3 tools/wave_timeline.py --config 4 > $o/r04f_wave_timeline.txt 2>&1
    cp /tmp/rel.so $d/libtscm_hip.so
  fi
  tail -14 $o/r04f_prof.txt; tail -12 $o/r04f_c5_prof.txt; cat $o/r04f_shards_config4_bench.json $o/r04f_shards_config5_bench.json
else
  python3 bench.py > $o/r04f_bench.json 2> $o/r04f_bench.err
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $o/r04f_bench_driver.json 2> /dev/null
  python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > $o/r04f_bench_100.json 2> /dev/null
  python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --jacobian-fp32 > $o/r04f_bench_100_f32.json 2> /dev/null
  python3 bench.py --config 5 --steps 100 --warmup 10 --no-cpu-baseline > $o/r04f_c5_bench.json 2> /dev/null
  python3 bench.py --config 5 --steps 100 --warmup 10 --no-cpu-baseline --jacobian-fp32 > $o/r04f_c5_bench_f32.json 2> /dev/null
  for c in 1 2 3; do python3 bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline > $o/r04f_c${c}_bench.json 2> /dev/null; done
  line r04f_bench r04f_bench_driver r04f_bench_100 r04f_bench_100_f32 r04f_c5_bench r04f_c5_bench_f32 r04f_c1_bench r04f_c2_bench r04f_c3_bench
fi
