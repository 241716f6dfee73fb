cd $GRAFT_REPO_ROOT
bash tools/variants.sh "PREV NEW" 3 2>&1
bash tools/prof.sh geo_NEW3 > /dev/null 2>&1; awk -F'",' '{print $1"\" "$2}' gpurun_out/geo_NEW3_kernel_medians.csv | grep -v "peak\|rocclr\|Name" | sed 's/tscm::DevProblem, tscm::DevState//' | cut -c1-100
python3 tools/regress_bits.py > gpurun_out/bits_new.txt 2>&1; tail -3 gpurun_out/bits_new.txt
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6
