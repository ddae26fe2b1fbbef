cd $GRAFT_REPO_ROOT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.3f ms per launch  frac %.3f' % (r['kernel_ms'], r['frac']))"; }
echo "## layout 2: $(MLGPU_DELAY_WINDOWS=2 timeout 300 python bench.py --no-cpu-baseline --workload strings 2>/dev/null | tail -1 | line)"
export MLGPU_CACHE_DIR=off
for x in FULLPIECE NOFLUSH; do
echo "## layout 2 $x: $(MLGPU_JIT_EXTRA_OPTS=-DMLGPU_RING_X_$x MLGPU_DELAY_WINDOWS=2 timeout 300 python bench.py --no-cpu-baseline --workload strings 2>/dev/null | tail -1 | line)"
done
