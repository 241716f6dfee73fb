cd $GRAFT_REPO_ROOT
d=tscm_calib_amd/csrc
cp $d/libtscm_hip.so /tmp/rel.so
echo "=== bits OLD (round-2 code path)"; cp $d/variants/libP0.so $d/libtscm_hip.so; timeout 600 python3 tools/regress_bits.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" > gpurun_out/bits_old.txt; cat gpurun_out/bits_old.txt
echo "=== bits NEW"; cp /tmp/rel.so $d/libtscm_hip.so; timeout 600 python3 tools/regress_bits.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" > gpurun_out/bits_new.txt; cat gpurun_out/bits_new.txt
diff gpurun_out/bits_old.txt gpurun_out/bits_new.txt > /dev/null && echo "BITS IDENTICAL" || echo "BITS DIFFER"
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8
bash tools/variants.sh "P0 N NP0 NX0P0 NP5 NP4" 2 > gpurun_out/r03_var3.log 2>&1
cat gpurun_out/r03_var3.log
cp $d/variants/libTN.so $d/libtscm_hip.so
echo "== TN"; timeout 120 python3 tools/wave_timeline.py --config 4 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
cp /tmp/rel.so $d/libtscm_hip.so
bash tools/prof.sh r03_epi2 > gpurun_out/r03_epi2.log 2>&1; cat gpurun_out/r03_epi2.log
