#!/bin/bash
# round 6, run 7: layout 4 with the next trip's sector asked for a trip ahead; 3-ring and 5-ring graphs in layouts 1 / 2 / 4
export TMPDIR=/tmp
mkdir -p gpurun_out/r06g
timeout 900 python -m pytest tests/test_gpu_delays.py -x -q -m gpu > gpurun_out/r06g/tests.txt 2>&1; tail -3 gpurun_out/r06g/tests.txt
MLGPU_SOAK_LAYOUT=4 timeout 900 python tools/ring_layout_soak.py 300 7 > gpurun_out/r06g/soak_layout4.txt 2>&1; tail -2 gpurun_out/r06g/soak_layout4.txt
for v in 65536 131072 262144; do
  MLGPU_DELAY_WINDOWS=4 timeout 300 python bench.py --workload allpass4 --voices $v --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('allpass4 V=$v layout=4', 'ms', round(r['kernel_ms'],3), 'frac', round(r['frac'],3))"
done 2>&1 | tee gpurun_out/r06g/allpass4.txt
timeout 600 python tools/experiments/r05_rings_multi.py > gpurun_out/r06g/rings_multi.txt 2>&1; tail -6 gpurun_out/r06g/rings_multi.txt
for l in 2 4; do MLGPU_DELAY_WINDOWS=$l timeout 300 python bench.py --workload strings --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('strings layout=$l', round(r['kernel_ms'],3), round(r['frac'],3))"; done
