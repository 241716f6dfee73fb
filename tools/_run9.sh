cd $GRAFT_REPO_ROOT
echo "=== bits"; timeout 600 python3 tools/regress_bits.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" > gpurun_out/bits_new3.txt
diff tools/regress_bits.expected gpurun_out/bits_new3.txt && echo "BITS IDENTICAL" || echo "BITS DIFFER"
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6
bash tools/prof.sh r03_final > gpurun_out/r03_final.log 2>&1; cat gpurun_out/r03_final.log | cut -c1-100; cut -c1-100 gpurun_out/r03_final_kernel_medians.csv
bash tools/pmc.sh r03_fetch FETCH_SIZE > gpurun_out/pmc_r03_fetch.txt 2>&1
bash tools/pmc.sh r03_write WRITE_SIZE > gpurun_out/pmc_r03_write.txt 2>&1
bash tools/pmc.sh r03_sq1 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" > gpurun_out/pmc_r03_sq1.txt 2>&1; grep k_eval gpurun_out/pmc_r03_sq1.txt
bash tools/pmc.sh r03_sq2 "SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" > gpurun_out/pmc_r03_sq2.txt 2>&1; grep k_eval gpurun_out/pmc_r03_sq2.txt
bash tools/pmc.sh r03_sq3 "GRBM_GUI_ACTIVE" > gpurun_out/pmc_r03_sq3.txt 2>&1; grep k_eval gpurun_out/pmc_r03_sq3.txt
python3 bench.py --jacobian-fp32 --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | grep '^{' | cut -c1-200
for c in 1 2 3; do python3 bench.py --config $c --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('config $c: %.0f it/s' % d['value'])"; done
python3 bench.py --config 2 --poses-fixed --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('config 2 poses fixed: %.0f it/s' % d['value'])"
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r03_final_bench_driver.json 2>/dev/null; cut -c1-300 gpurun_out/r03_final_bench_driver.json
python3 bench.py > gpurun_out/r03_final_bench.json 2>/dev/null; cut -c1-300 gpurun_out/r03_final_bench.json
