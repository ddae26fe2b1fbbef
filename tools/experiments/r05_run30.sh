cd $GRAFT_REPO_ROOT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.3f ms per launch  frac %.3f' % (r['kernel_ms'], r['frac']))"; }
for l in 0 1 2; do
echo "## uniform delay times, layout $l: $(MLGPU_UNIFORM_DELAY=1 MLGPU_DELAY_WINDOWS=$l timeout 300 python bench.py --no-cpu-baseline --workload strings 2>/dev/null | tail -1 | line)"
done
for l in 0 1 2; do
echo "## per-voice delay times, layout $l: $(MLGPU_DELAY_WINDOWS=$l timeout 300 python bench.py --no-cpu-baseline --workload strings 2>/dev/null | tail -1 | line)"
done
