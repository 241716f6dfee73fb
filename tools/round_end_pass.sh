# usage (GPU box, via gpurun): bash tools/round_end_pass.sh  -- kernel stats + medians, bench lines, PMC passes, config 5, kernel / wave timelines
# of the release build into gpurun_out/r03f_*; then, here: python tools/record_pmc.py gpurun_out/pmc_r03f_fetch gpurun_out/pmc_r03f_write 4 gpurun_out/pmc_r03f_sq{1,2,3}
# and bash tools/round_end_bench_lines.sh on the box (bench lines that carry roofline.traffic of the recorded sources)
cd $GRAFT_REPO_ROOT
d=tscm_calib_amd/csrc
o=gpurun_out
bash tools/prof.sh r03f > $o/r03f_prof.txt 2>&1
python3 bench.py > $o/r03f_bench.json 2> $o/r03f_bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $o/r03f_bench_driver.json 2> /dev/null
python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > $o/r03f_bench_100.json 2> /dev/null
python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --exec-flags 4 > $o/r03f_bench_100_gram16.json 2> /dev/null
python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --jacobian-fp32 > $o/r03f_bench_100_f32.json 2> /dev/null
bash tools/pmc.sh r03f_fetch FETCH_SIZE > $o/pmc_r03f_fetch.txt 2>&1
bash tools/pmc.sh r03f_write WRITE_SIZE > $o/pmc_r03f_write.txt 2>&1
bash tools/pmc.sh r03f_sq1 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" > $o/pmc_r03f_sq1.txt 2>&1
bash tools/pmc.sh r03f_sq2 "SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" > $o/pmc_r03f_sq2.txt 2>&1
bash tools/pmc.sh r03f_sq3 "GRBM_GUI_ACTIVE" > $o/pmc_r03f_sq3.txt 2>&1
bash tools/prof.sh r03f_c5 --config 5 > $o/r03f_c5_prof.txt 2>&1
python3 bench.py --config 5 --steps 100 --warmup 10 --no-cpu-baseline > $o/r03f_c5_bench.json 2> /dev/null
cp $d/libtscm_hip.so /tmp/rel.so
cp $d/variants/libT.so $d/libtscm_hip.so
python3 tools/kernel_timeline.py --config 4 > $o/r03f_kernel_timeline.txt 2>&1
python3 tools/wave_timeline.py --config 4 > $o/r03f_wave_timeline.txt 2>&1
cp /tmp/rel.so $d/libtscm_hip.so
for f in r03f_bench r03f_bench_driver r03f_bench_100 r03f_bench_100_gram16 r03f_bench_100_f32 r03f_c5_bench; do grep '^{' $o/$f.json | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$f: steps %d %.0f it/s  %.1f us/step  eval %.2f us (%d timed) frac %.3f traffic %s' % (d['steps'], d['value'], 1e3*d['ms_per_step'], 1e3*r['avg_launch_ms'], r['launches'], r['frac'], r.get('traffic')))"; done
cat $o/r03f_prof.txt | tail -14
tail -12 $o/r03f_kernel_timeline.txt
