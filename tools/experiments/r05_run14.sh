cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_delays.py -x -q -m gpu 2>&1 | tail -4
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.4g voice-samples/s  %.3f ms per launch  frac %.3f' % (d['value'], r['kernel_ms'], r['frac']))"; }
for l in 1 2; do echo "## strings layout $l: $(MLGPU_DELAY_WINDOWS=$l timeout 300 python bench.py --no-cpu-baseline --workload strings 2>/dev/null | tail -1 | line)"; done
echo "## strings layout 2 no phases: $(MLGPU_JIT_EXTRA_OPTS=-DMLGPU_RING_PHASES=0 MLGPU_DELAY_WINDOWS=2 timeout 300 python bench.py --no-cpu-baseline --workload strings 2>/dev/null | tail -1 | line)"
