cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6
python3 tools/regress_bits.py > gpurun_out/bits_now.txt 2>&1; diff gpurun_out/bits_now.txt tools/regress_bits.expected && echo BITS_SAME
