cd $GRAFT_REPO_ROOT
d=tscm_calib_amd/csrc
cp $d/libtscm_hip.so /tmp/rel.so
cp $d/variants/libTG.so $d/libtscm_hip.so
echo "== TG"; timeout 120 python3 tools/wave_timeline.py --config 4 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
cp /tmp/rel.so $d/libtscm_hip.so
