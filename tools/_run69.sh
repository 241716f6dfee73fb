cd $GRAFT_REPO_ROOT
bash tools/prof_tool.sh sh8 tools/bench_shards.py --config 4 --worlds 8 > /dev/null 2>&1
python3 tools/kernel_medians.py gpurun_out/prof_sh8 gpurun_out/sh8_kernel_medians.csv > /dev/null 2>&1
awk -F'",' '{print $1"\" "$2}' gpurun_out/sh8_kernel_medians.csv | sed 's/tscm::DevProblem, tscm::DevState//' | cut -c1-110 | head -16
