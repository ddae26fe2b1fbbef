cd $GRAFT_REPO_ROOT
root=$PWD
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.4g voice-samples/s  %.3f ms per block (wall %.3f ms per step of %d)' % (d['value'], r['kernel_ms'], d['ms_per_step'], d['config']['launches_per_step']))"; }
echo "## synthfused sum=graph: $(MLGPU_BENCH_MIXDOWN=graph timeout 300 python bench.py --no-cpu-baseline --workload synthfused 2>/dev/null | tail -1 | line)"
echo "## synthfused sum=graph: $(MLGPU_BENCH_MIXDOWN=graph timeout 300 python bench.py --no-cpu-baseline --workload synthfused 2>/dev/null | tail -1 | line)"
echo "## events: $(timeout 300 python bench.py --no-cpu-baseline --workload events 2>/dev/null | tail -1 | line)"
