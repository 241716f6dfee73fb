#!/bin/bash
# usage: bash tools/round_end.sh <tag> <part>
#   part 1 (GPU box, via gpurun): the evidence of the release build into gpurun_out/<tag>f_* -- kernel stats + medians (configs 4, 5,
#           fp32 tier), PMC passes (configs 4 and 5, separate passes per counter group), 8-shard profiles, shard benches, and with
#           csrc/variants/libT.so (make variant VARIANT=T EXTRA=-DTSCM_WAVE_TIMELINE) the kernel / wave / phase timelines
#   (here:  python tools/record_pmc.py gpurun_out/pmc_<tag>f_fetch gpurun_out/pmc_<tag>f_write 4 gpurun_out/pmc_<tag>f_sq{1,2,3}, and
#           ... gpurun_out/pmc_<tag>f_c5_fetch gpurun_out/pmc_<tag>f_c5_write 5  -> profiles/pmc_eval_gram.json)
#   part 2 (GPU box): the bench lines, which then carry roofline.traffic of the recorded sources
#   part 3 (here): copies the summaries from gpurun_out/ (scratch) to profiles/<tag>_* (tracked)
tag=${1:?tag, e.g. r05}; part=${2:-1}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
d=tscm_calib_amd/csrc; o=gpurun_out; p=profiles; t=${tag}f
line() { for f in "$@"; do grep '^{' $o/$f.json | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$f: steps %d %.0f it/s  %.1f us/step  %s %.2f us (%d timed) frac %.3f iteration %.3f of-ceiling %s traffic %s cpu %s' % (d['steps'], d['value'], 1e3*d['ms_per_step'], r['kernel'], 1e3*r['avg_launch_ms'], r['launches'], r['frac'], r.get('iteration_frac', 0), r.get('frac_of_measured_ceiling'), r.get('traffic'), (d.get('cpu_baseline') or {}).get('value')))"; done; }
if [ "$part" = "1" ]; then
  bash tools/prof.sh $t > $o/${t}_prof.txt 2>&1
  bash tools/prof.sh ${t}_c5 --config 5 > $o/${t}_c5_prof.txt 2>&1
  # the performance regression guard next to the bit guard: kernel medians of configs 4 and 5 against tools/regress_perf.expected (3 % band)
  python3 tools/regress_perf.py $o/${t}_kernel_medians.csv $o/${t}_c5_kernel_medians.csv > $o/${t}_regress_perf.txt 2>&1 || echo "ROUND-END PASS FAILED: regress_perf (see $o/${t}_regress_perf.txt)"
  cat $o/${t}_regress_perf.txt
  python3 tools/regress_bits.py --check > $o/${t}_regress_bits.txt 2>&1 || echo "ROUND-END PASS FAILED: regress_bits"
  tail -3 $o/${t}_regress_bits.txt
  bash tools/prof.sh ${t}_f32 --jacobian-fp32 > $o/${t}_f32_prof.txt 2>&1
  bash tools/prof.sh ${t}_c5_f32 --config 5 --jacobian-fp32 > $o/${t}_c5_f32_prof.txt 2>&1
  bash tools/pmc.sh ${t}_fetch FETCH_SIZE > $o/pmc_${t}_fetch.txt 2>&1
  bash tools/pmc.sh ${t}_write WRITE_SIZE > $o/pmc_${t}_write.txt 2>&1
  bash tools/pmc.sh ${t}_sq1 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" > $o/pmc_${t}_sq1.txt 2>&1
  bash tools/pmc.sh ${t}_sq2 "SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" > $o/pmc_${t}_sq2.txt 2>&1
  bash tools/pmc.sh ${t}_sq3 "GRBM_GUI_ACTIVE" > $o/pmc_${t}_sq3.txt 2>&1
  bash tools/pmc.sh ${t}_c5_fetch FETCH_SIZE --config 5 > $o/pmc_${t}_c5_fetch.txt 2>&1
  bash tools/pmc.sh ${t}_c5_write WRITE_SIZE --config 5 > $o/pmc_${t}_c5_write.txt 2>&1
  bash tools/prof_shards.sh ${t}_shards8_config4 4 8 > /dev/null 2>&1
  bash tools/prof_shards.sh ${t}_shards8_config5 5 8 > /dev/null 2>&1
  python3 tools/bench_shards.py --config 4 --worlds 1,2,4,8 > $o/${t}_shards_config4_bench.json 2> /dev/null
  python3 tools/bench_shards.py --config 5 --worlds 1,8 > $o/${t}_shards_config5_bench.json 2> /dev/null
  python3 tools/ipc_check.py --world 2 --config 3 > $o/${t}_ipc_check.txt 2>&1
  if [ -f $d/variants/libT.so ]; then
    cp $d/libtscm_hip.so /tmp/rel.so
    trap 'cp /tmp/rel.so $d/libtscm_hip.so' EXIT
    cp $d/variants/libT.so $d/libtscm_hip.so
    python3 tools/kernel_timeline.py --config 4 > $o/${t}_kernel_timeline.txt 2>&1
    python3 tools/kernel_timeline.py --config 5 > $o/${t}_c5_kernel_timeline.txt 2>&1
    python3 tools/wave_timeline.py --config 4 > $o/${t}_wave_timeline.txt 2>&1
    python3 tools/phase_timeline.py --config 4 > $o/${t}_phase_timeline.txt 2>&1
    cp /tmp/rel.so $d/libtscm_hip.so
  fi
  tail -14 $o/${t}_prof.txt; tail -12 $o/${t}_c5_prof.txt; cat $o/${t}_shards_config4_bench.json $o/${t}_shards_config5_bench.json
elif [ "$part" = "2" ]; then
  python3 bench.py > $o/${t}_bench.json 2> $o/${t}_bench.err
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $o/${t}_bench_driver.json 2> /dev/null
  python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > $o/${t}_bench_100.json 2> /dev/null
  python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --jacobian-fp32 > $o/${t}_bench_100_f32.json 2> /dev/null
  python3 bench.py --config 5 --steps 100 --warmup 10 --no-cpu-baseline > $o/${t}_c5_bench.json 2> /dev/null
  python3 bench.py --config 5 --steps 100 --warmup 10 --no-cpu-baseline --jacobian-fp32 > $o/${t}_c5_bench_f32.json 2> /dev/null
  for c in 1 2 3; do python3 bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline > $o/${t}_c${c}_bench.json 2> /dev/null; done
  line ${t}_bench ${t}_bench_driver ${t}_bench_100 ${t}_bench_100_f32 ${t}_c5_bench ${t}_c5_bench_f32 ${t}_c1_bench ${t}_c2_bench ${t}_c3_bench
else
  newest() { ls -t $o/prof_$1/*/*_kernel_stats.csv | head -1; }
  cp "$(newest $t)" $p/${tag}_final_kernel_stats.csv
  cp "$(newest ${t}_c5)" $p/${tag}_config5_kernel_stats.csv
  cp $o/${t}_kernel_medians.csv $p/${tag}_final_kernel_medians.csv
  cp $o/${t}_c5_kernel_medians.csv $p/${tag}_config5_kernel_medians.csv
  cp $o/${t}_f32_kernel_medians.csv $p/${tag}_fp32_config4_kernel_medians.csv
  cp $o/${t}_c5_f32_kernel_medians.csv $p/${tag}_fp32_config5_kernel_medians.csv
  cp $o/${t}_shards8_config4_kernel_medians.csv $p/${tag}_shards8_config4_kernel_medians.csv
  cp $o/${t}_shards8_config5_kernel_medians.csv $p/${tag}_shards8_config5_kernel_medians.csv
  cp $o/${t}_shards_config4_bench.json $p/${tag}_shards_config4_bench.json
  cp $o/${t}_shards_config5_bench.json $p/${tag}_shards_config5_bench.json
  for f in kernel_timeline c5_kernel_timeline:config5_kernel_timeline wave_timeline phase_timeline ipc_check regress_perf regress_bits; do [ -f $o/${t}_${f%%:*}.txt ] && cp $o/${t}_${f%%:*}.txt $p/${tag}_${f##*:}.txt; done
  for f in bench:final_bench bench_driver:final_bench_driver_command bench_100:final_bench_100 bench_100_f32:fp32_config4_bench c5_bench:config5_bench c5_bench_f32:fp32_config5_bench c1_bench:config1_bench c2_bench:config2_bench c3_bench:config3_bench; do
    [ -f $o/${t}_${f%%:*}.json ] && grep '^{' $o/${t}_${f%%:*}.json > $p/${tag}_${f##*:}.json
  done
  git status --short $p | head -40
fi
