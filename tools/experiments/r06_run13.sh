#!/bin/bash
export TMPDIR=/tmp
for v in 65536 131072 262144; do
  timeout 600 python bench.py --workload reverb --voices $v --no-cpu-baseline --steps 5 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('reverb V=$v', 'ms', round(r['kernel_ms'],3), 'frac', round(r['frac'],3), 'value', d['value'], 'stereo reverbs in real time', int(d['value']/48000))"
done
MLGPU_UNIFORM_DELAY=1 MLGPU_DELAY_WINDOWS=0 timeout 300 python bench.py --workload strings --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('strings uniform delay layout 0', round(r['kernel_ms'],3), round(r['frac'],3))"
MLGPU_UNIFORM_DELAY=1 MLGPU_DELAY_WINDOWS=0 timeout 300 python bench.py --workload allpass4 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('allpass4 uniform delay layout 0', round(r['kernel_ms'],3), round(r['frac'],3))"
MLGPU_UNIFORM_DELAY=1 MLGPU_DELAY_WINDOWS=4 timeout 300 python bench.py --workload allpass4 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('allpass4 uniform delay layout 4', round(r['kernel_ms'],3), round(r['frac'],3))"
python tools/aaltoverb_bench.py 65536 16 0
