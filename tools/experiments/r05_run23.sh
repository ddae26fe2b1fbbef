cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_delays.py -x -q -m gpu 2>&1 | tail -5
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.3f ms per launch  frac %.3f' % (r['kernel_ms'], r['frac']))"; }
for i in 1 2; do
echo "## layout 1: $(MLGPU_DELAY_WINDOWS=1 timeout 300 python bench.py --no-cpu-baseline --workload strings 2>/dev/null | tail -1 | line)"
echo "## layout 2: $(MLGPU_DELAY_WINDOWS=2 timeout 300 python bench.py --no-cpu-baseline --workload strings 2>/dev/null | tail -1 | line)"
done
export MLGPU_CACHE_DIR=off
echo "poisoned misses:"; MLGPU_JIT_EXTRA_OPTS=-DMLGPU_RING_X_MISSPOISON timeout 300 python tools/experiments/r05_miss_probe.py 2>&1 | tail -8
for x in NOFLUSH NOMISS; do
echo "## layout 2 $x: $(MLGPU_JIT_EXTRA_OPTS=-DMLGPU_RING_X_$x MLGPU_DELAY_WINDOWS=2 timeout 300 python bench.py --no-cpu-baseline --workload strings 2>/dev/null | tail -1 | line)"
done
