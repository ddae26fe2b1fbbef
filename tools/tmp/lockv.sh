python -m pytest tests/test_gpu_events.py tests/test_gpu_graph.py -m gpu -x -q 2>&1 | tail -2
run() { echo "== $*"; env "$@" python bench.py --workload $W --no-cpu-baseline --no-extras --no-live-counters --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"; }
for W in synth synthrows; do run A=1; run MLGPU_GRAPH_LOCK_VERSIONS=0; run A=2; run MLGPU_GRAPH_LOCK_VERSIONS=0; done
