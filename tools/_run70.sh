cd $GRAFT_REPO_ROOT
python3 tools/bench_shards.py --config 4 --worlds 1,2,4,8 2>&1 | tail -1 | cut -c1-900
python3 tools/bench_shards.py --config 5 --worlds 1,8 2>&1 | tail -1 | cut -c1-500
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6
