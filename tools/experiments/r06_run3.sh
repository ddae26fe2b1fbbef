#!/bin/bash
# round 6, run 3: ring layout 4 (sector trips): parity, then allpass4 / strings against layouts 1 and 2
export TMPDIR=/tmp
mkdir -p gpurun_out/r06c
timeout 900 python -m pytest tests/test_gpu_delays.py -x -q -m gpu > gpurun_out/r06c/tests_delays.txt 2>&1; tail -8 gpurun_out/r06c/tests_delays.txt
MLGPU_SOAK_LAYOUT=4 timeout 900 python tools/ring_layout_soak.py 200 3 > gpurun_out/r06c/soak_layout4.txt 2>&1; tail -6 gpurun_out/r06c/soak_layout4.txt
for v in 16384 131072 262144; do for l in 1 4; do
  MLGPU_DELAY_WINDOWS=$l timeout 300 python bench.py --workload allpass4 --voices $v --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('allpass4 V=$v layout=$l', 'ms', round(r['kernel_ms'],3), 'frac', round(r['frac'],3))"
done; done 2>&1 | tee gpurun_out/r06c/allpass4.txt
for l in 1 2 4; do
  MLGPU_DELAY_WINDOWS=$l timeout 300 python bench.py --workload strings --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('strings layout=$l', 'ms', round(r['kernel_ms'],3), 'frac', round(r['frac'],3))"
done 2>&1 | tee gpurun_out/r06c/strings.txt
