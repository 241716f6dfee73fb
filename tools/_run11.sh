cd $GRAFT_REPO_ROOT
bash tools/variants.sh "G GA1 GA2 GA4 GA5 GA7 GP0" 1 2>&1
d=tscm_calib_amd/csrc
cp $d/libtscm_hip.so /tmp/rel.so
cp $d/variants/libTG.so $d/libtscm_hip.so
echo "== TG"; timeout 120 python3 tools/wave_timeline.py --config 4 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
cp /tmp/rel.so $d/libtscm_hip.so
bash tools/pmc.sh g4_sq1 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" 2>&1 | grep k_eval
bash tools/pmc.sh g4_sq2 "SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" 2>&1 | grep k_eval
bash tools/pmc.sh g4_sq3 "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" 2>&1 | grep k_eval
