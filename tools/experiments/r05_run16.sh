cd $GRAFT_REPO_ROOT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.4g voice-samples/s  %.3f ms per block (wall %.3f ms per step of %d)' % (d['value'], r['kernel_ms'], d['ms_per_step'], d['config']['launches_per_step']))"; }
for i in 1 2 3; do
echo "## synth two streams: $(timeout 300 python bench.py --no-cpu-baseline --workload synth 2>/dev/null | tail -1 | line)"
echo "## synth one stream:  $(MLGPU_EVENTS_ONE_STREAM=1 timeout 300 python bench.py --no-cpu-baseline --workload synth 2>/dev/null | tail -1 | line)"
done
