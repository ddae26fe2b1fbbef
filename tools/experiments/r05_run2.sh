cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests/test_gpu_events.py tests/test_gpu_graph.py -x -q -m gpu -k "event_rows or instrument_bank or voice_sum or dropin" 2>&1 | tail -15
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.4g voice-samples/s  %.3f ms per block (wall %.3f ms per step of %d)' % (d['value'], r['kernel_ms'], d['ms_per_step'], d['config']['launches_per_step']))"; }
for m in kernel graph; do echo "## synthfused  sum=$m: $(MLGPU_BENCH_MIXDOWN=$m timeout 300 python bench.py --no-cpu-baseline --workload synthfused 2>gpurun_out/r05/sf_$m.err | tail -1 | line)"; done
echo "## synthfused sum=graph dpp: $(MLGPU_GRAPH_GROUP_SUM=dpp MLGPU_BENCH_MIXDOWN=graph timeout 300 python bench.py --no-cpu-baseline --workload synthfused 2>/dev/null | tail -1 | line)"
echo "## synth sum=graph: $(MLGPU_BENCH_MIXDOWN=graph timeout 300 python bench.py --no-cpu-baseline --workload synth 2>/dev/null | tail -1 | line)"
cd /tmp && export TMPDIR=/tmp
MLGPU_BENCH_MIXDOWN=graph timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r05/prof_sf -o sf -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --workload synthfused --steps 5 --warmup 2 > /dev/null 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/r05/prof_sf -name "*kernel_stats.csv" | head -1 | xargs head -8
