cd $GRAFT_REPO_ROOT
echo "=== bits"; timeout 600 python3 tools/regress_bits.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" > gpurun_out/bits_g4.txt
diff tools/regress_bits.expected gpurun_out/bits_g4.txt && echo "BITS IDENTICAL" || echo "BITS DIFFER"
for i in 1 2; do
for f in 0 4; do python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --exec-flags $f 2>/dev/null | grep '^{' | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('exec_flags $f: %.0f it/s  %.1f us/step  eval %.2f us' % (d['value'], 1e3*d['ms_per_step'], 1e3*r['avg_launch_ms']))"; done; done
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6
