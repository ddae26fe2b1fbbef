#!/bin/bash
# round 6, run 6: layout 4 with whole-sector stores and the PitchbendableDelay's one-ring form; the sharded mixdown
export TMPDIR=/tmp
mkdir -p gpurun_out/r06f
timeout 900 python -m pytest tests/test_gpu_delays.py tests/test_gpu_processbuffer.py -x -q -m gpu > gpurun_out/r06f/tests.txt 2>&1; tail -6 gpurun_out/r06f/tests.txt
MLGPU_SOAK_LAYOUT=4 timeout 900 python tools/ring_layout_soak.py 200 5 > gpurun_out/r06f/soak_layout4.txt 2>&1; tail -4 gpurun_out/r06f/soak_layout4.txt
timeout 300 tests/cpp/multi_engine_test > gpurun_out/r06f/multi_engine.txt 2>&1; tail -5 gpurun_out/r06f/multi_engine.txt
for v in 65536 131072; do
  MLGPU_DELAY_WINDOWS=4 timeout 300 python bench.py --workload allpass4 --voices $v --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('allpass4 V=$v layout=4', 'ms', round(r['kernel_ms'],3), 'frac', round(r['frac'],3))"
done 2>&1 | tee gpurun_out/r06f/allpass4.txt
MLGPU_DELAY_WINDOWS=4 EXTRA="--voices 131072" timeout 1200 bash tools/gpu_profile_all.sh r06f allpass4 > gpurun_out/r06f/profile.log 2>&1
grep -A6 "== mlgpu_graph_kernel" gpurun_out/profiles_r06f/r06f_allpass4_pmc.txt | grep -E "FETCH_SIZE|WRITE_SIZE|SQ_WAIT_ANY |SQ_WAVE_CYCLES|SQ_INSTS_VALU "
cp gpurun_out/profiles_r06f/*allpass4* gpurun_out/r06f/
