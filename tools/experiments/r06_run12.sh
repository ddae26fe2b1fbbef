#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r06l
timeout 900 python -m pytest tests/test_gpu_delays.py tests/test_gpu_examples.py tests/test_gpu_widened_parity.py tests/test_gpu_dropin.py tests/test_gpu_regions.py tests/test_gpu_immediate.py -x -q -m gpu 2>&1 | tail -3
for lay in 2 4; do MLGPU_SOAK_LAYOUT=$lay timeout 900 python tools/ring_layout_soak.py 200 17 2>&1 | tail -1; done
for v in 65536 131072; do
  timeout 600 python bench.py --workload reverb --voices $v --no-cpu-baseline --steps 5 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('reverb V=$v', 'ms', round(r['kernel_ms'],3), 'frac', round(r['frac'],3), 'value', d['value'], 'stereo reverbs in real time', int(d['value']/48000))"
done
MLGPU_UNIFORM_DELAY=1 MLGPU_DELAY_WINDOWS=0 timeout 300 python bench.py --workload allpass4 --voices 65536 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('allpass4 uniform delay layout 0', round(r['kernel_ms'],3), round(r['frac'],3))"
MLGPU_UNIFORM_DELAY=1 MLGPU_DELAY_WINDOWS=0 timeout 300 python bench.py --workload strings --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('strings uniform delay layout 0', round(r['kernel_ms'],3), round(r['frac'],3))"
python tools/aaltoverb_bench.py 16384 16 0
