#!/bin/bash
# usage (here, after `gpurun -- bash tools/r04_round_end.sh 1`, `python tools/record_pmc.py ...`, `gpurun -- bash tools/r04_round_end.sh 2`):
# bash tools/r04_collect.sh  -- copies the round's evidence from gpurun_out/ (scratch) to profiles/ (tracked)
cd "$(dirname "$0")/.."
o=gpurun_out; p=profiles
newest() { ls -t $o/prof_$1/*/*_kernel_stats.csv | head -1; }
cp "$(newest r04f)" $p/r04_final_kernel_stats.csv
cp "$(newest r04f_c5)" $p/r04_config5_kernel_stats.csv
cp $o/r04f_kernel_medians.csv $p/r04_final_kernel_medians.csv
cp $o/r04f_c5_kernel_medians.csv $p/r04_config5_kernel_medians.csv
cp $o/r04f_f32_kernel_medians.csv $p/r04_fp32_config4_kernel_medians.csv
cp $o/r04f_c5_f32_kernel_medians.csv $p/r04_fp32_config5_kernel_medians.csv
cp $o/r04f_shards8_config4_kernel_medians.csv $p/r04_shards8_config4_kernel_medians.csv
cp $o/r04f_shards8_config5_kernel_medians.csv $p/r04_shards8_config5_kernel_medians.csv
cp $o/r04f_shards_config4_bench.json $p/r04_shards_config4_bench.json
cp $o/r04f_shards_config5_bench.json $p/r04_shards_config5_bench.json
cp $o/r04f_kernel_timeline.txt $p/r04_kernel_timeline.txt
cp $o/r04f_c5_kernel_timeline.txt $p/r04_config5_kernel_timeline.txt
cp $o/r04f_wave_timeline.txt $p/r04_wave_timeline.txt
for f in bench:final_bench bench_driver:final_bench_driver_command bench_100:final_bench_100 bench_100_f32:fp32_config4_bench c5_bench:config5_bench c5_bench_f32:fp32_config5_bench c1_bench:config1_bench c2_bench:config2_bench c3_bench:config3_bench; do
  grep '^{' $o/r04f_${f%%:*}.json > $p/r04_${f##*:}.json
done
git status --short $p | head -40
