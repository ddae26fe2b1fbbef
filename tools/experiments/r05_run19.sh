cd $GRAFT_REPO_ROOT
export MLGPU_CACHE_DIR=off
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.3f ms per launch  frac %.3f' % (r['kernel_ms'], r['frac']))"; }
echo "## layout 1: $(MLGPU_DELAY_WINDOWS=1 timeout 300 python bench.py --no-cpu-baseline --workload strings 2>/dev/null | tail -1 | line)"
echo "## layout 2: $(MLGPU_DELAY_WINDOWS=2 timeout 300 python bench.py --no-cpu-baseline --workload strings 2>/dev/null | tail -1 | line)"
for x in NOFLUSH NOBOUNDARY NOLOAD NOLDSREAD NOMISS; do
echo "## layout 2 $x: $(MLGPU_JIT_EXTRA_OPTS=-DMLGPU_RING_X_$x MLGPU_DELAY_WINDOWS=2 timeout 300 python bench.py --no-cpu-baseline --workload strings 2>/dev/null | tail -1 | line)"
done
echo "## layout 2 NOFLUSH+NOBOUNDARY: $(MLGPU_JIT_EXTRA_OPTS='-DMLGPU_RING_X_NOFLUSH -DMLGPU_RING_X_NOBOUNDARY' MLGPU_DELAY_WINDOWS=2 timeout 300 python bench.py --no-cpu-baseline --workload strings 2>/dev/null | tail -1 | line)"
echo "## layout 2 all off: $(MLGPU_JIT_EXTRA_OPTS='-DMLGPU_RING_X_NOFLUSH -DMLGPU_RING_X_NOBOUNDARY -DMLGPU_RING_X_NOLDSREAD -DMLGPU_RING_X_NOMISS' MLGPU_DELAY_WINDOWS=2 timeout 300 python bench.py --no-cpu-baseline --workload strings 2>/dev/null | tail -1 | line)"
echo "## layout 2 FB_AHEAD=0: $(MLGPU_GRAPH_FB_AHEAD=0 MLGPU_DELAY_WINDOWS=2 timeout 300 python bench.py --no-cpu-baseline --workload strings 2>/dev/null | tail -1 | line)"
