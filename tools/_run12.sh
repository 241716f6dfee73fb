cd $GRAFT_REPO_ROOT
d=tscm_calib_amd/csrc
cp $d/libtscm_hip.so /tmp/rel.so
for v in G GA1 GA2 GA4 GA5 GA7; do
  cp $d/variants/lib$v.so $d/libtscm_hip.so; rm -rf gpurun_out/prof_abl_$v
  cd /tmp; export TMPDIR=/tmp
  rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_abl_$v -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  echo "== $v"; python3 tools/kernel_medians.py gpurun_out/prof_abl_$v | grep "k_eval_gram4" | cut -c1-120
done
cp /tmp/rel.so $d/libtscm_hip.so
