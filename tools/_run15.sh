cd $GRAFT_REPO_ROOT
echo "=== bits"; timeout 600 python3 tools/regress_bits.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" > gpurun_out/bits_final.txt
diff tools/regress_bits.expected gpurun_out/bits_final.txt && echo "BITS IDENTICAL" || echo "BITS DIFFER"
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6
bash tools/prof.sh r03_g4 > gpurun_out/r03_g4.log 2>&1; cat gpurun_out/r03_g4.log | cut -c1-100; cut -c1-100 gpurun_out/r03_g4_kernel_medians.csv
bash tools/pmc.sh r03b_fetch FETCH_SIZE > gpurun_out/pmc_r03b_fetch.txt 2>&1
bash tools/pmc.sh r03b_write WRITE_SIZE > gpurun_out/pmc_r03b_write.txt 2>&1
bash tools/pmc.sh r03b_sq1 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" > gpurun_out/pmc_r03b_sq1.txt 2>&1
bash tools/pmc.sh r03b_sq2 "SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" > gpurun_out/pmc_r03b_sq2.txt 2>&1
bash tools/pmc.sh r03b_sq3 "GRBM_GUI_ACTIVE" > gpurun_out/pmc_r03b_sq3.txt 2>&1
for i in 1 2; do for f in 0 4; do python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --exec-flags $f 2>/dev/null | grep '^{' | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('exec_flags $f: %.0f it/s  %.1f us/step  eval %.2f us  frac %.3f  of measured ceiling %.3f  (mfma16 %.1f mfma4 %.1f valu %.1f TF/s)' % (d['value'], 1e3*d['ms_per_step'], 1e3*r['avg_launch_ms'], r['frac'], r['frac_of_measured_ceiling'], r['peak_measured_mfma_f64_16x16x4'], r['peak_measured_mfma_f64_4x4x4'], r['peak_measured_valu_f64']))"; done; done
python3 bench.py --config 5 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r03_g4_c5_bench.json; python3 -c "
import json; d=json.load(open('gpurun_out/r03_g4_c5_bench.json')); print('config 5: %.0f it/s eval %.1f us frac %.3f' % (d['value'], 1e3*d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
for c in 1 2 3; do python3 bench.py --config $c --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('config $c: %.0f it/s' % d['value'])"; done
python3 tools/bench_shards.py --config 4 --worlds 1,2,4,8 2>/dev/null | grep '^{' > gpurun_out/r03_g4_shards_c4.json; cat gpurun_out/r03_g4_shards_c4.json
python3 tools/bench_shards.py --config 5 --worlds 1,2,4,8 2>/dev/null | grep '^{' > gpurun_out/r03_g4_shards_c5.json; cat gpurun_out/r03_g4_shards_c5.json
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r03_g4_bench_driver.json 2>/dev/null; cut -c1-200 gpurun_out/r03_g4_bench_driver.json
python3 bench.py > gpurun_out/r03_g4_bench.json 2>/dev/null; cut -c1-200 gpurun_out/r03_g4_bench.json
