cd $GRAFT_REPO_ROOT
bash tools/variants.sh "G GV GK GP11 GP12 GKP11" 2 2>&1
d=tscm_calib_amd/csrc
cp $d/libtscm_hip.so /tmp/rel.so
cp $d/variants/libTGK.so $d/libtscm_hip.so
echo "== TGK"; timeout 120 python3 tools/wave_timeline.py --config 4 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | head -3
cp $d/variants/libGK.so $d/libtscm_hip.so
timeout 600 python3 tools/regress_bits.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" > gpurun_out/bits_gk.txt; diff tools/regress_bits.expected gpurun_out/bits_gk.txt > /dev/null && echo "GK BITS IDENTICAL" || echo "GK BITS DIFFER"
cp /tmp/rel.so $d/libtscm_hip.so
