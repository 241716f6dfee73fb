cd $GRAFT_REPO_ROOT
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-400
