python -m pytest tests/test_gpu_delays.py tests/test_gpu_examples.py tests/test_gpu_widened_parity.py tests/test_gpu_graph.py tests/test_gpu_dropin.py tests/test_gpu_regions.py -m gpu -x -q 2>&1 | tail -4
run() { echo "== $*"; env "$@" python bench.py --workload $W --no-cpu-baseline --no-extras --no-live-counters --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])"; }
W=reverb
run A=1
run MLGPU_GRAPH_ROW_ADDR32=0
run MLGPU_GRAPH_UNROLL=4


W=strings
run MLGPU_DELAY_WINDOWS=0
run MLGPU_DELAY_WINDOWS=0 MLGPU_GRAPH_ROW_ADDR32=0
python tools/tmp/rv_src.py
W=cfg5
run A=1
W=synth
run A=1
