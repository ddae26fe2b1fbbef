# The instrument bank (bench.py --workload synth / synthfused) in its forms: the per-instrument voice sum as a kernel of its own
# (MLGPU_BENCH_MIXDOWN=kernel, the default) or inside the voice kernel (=graph), the events kernel on the same or on a second stream.
cd $GRAFT_REPO_ROOT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.4g voice-samples/s  %.3f ms per block' % (d['value'], r['kernel_ms']))"; }
for rep in 1 2; do
for w in synth synthfused; do for m in kernel graph; do echo "## $w  sum=$m: $(MLGPU_BENCH_MIXDOWN=$m python bench.py --no-cpu-baseline --workload $w 2>/dev/null | tail -1 | line)"; done; done
for m in kernel graph; do echo "## synth --two-streams  sum=$m: $(MLGPU_BENCH_MIXDOWN=$m python bench.py --no-cpu-baseline --workload synth --two-streams 2>/dev/null | tail -1 | line)"; done
done
