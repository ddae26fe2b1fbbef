#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r06k
for v in 65536 131072; do
  timeout 600 python bench.py --workload reverb --voices $v --no-cpu-baseline --steps 5 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('reverb V=$v', 'ms', round(r['kernel_ms'],3), 'frac', round(r['frac'],3), 'value', d['value'], 'stereo reverbs in real time', int(d['value']/48000))"
done
MLGPU_UNIFORM_DELAY=1 MLGPU_DELAY_WINDOWS=0 timeout 300 python bench.py --workload allpass4 --voices 65536 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('allpass4 uniform delay layout 0', round(r['kernel_ms'],3), round(r['frac'],3))"
timeout 900 python -m pytest tests/test_gpu_delays.py tests/test_gpu_examples.py -x -q -m gpu 2>&1 | tail -2
timeout 1200 bash tools/gpu_profile_all.sh r06k reverb > gpurun_out/r06k/profile.log 2>&1
grep -A8 "== mlgpu_graph_kernel" gpurun_out/profiles_r06k/r06k_reverb_pmc.txt | grep -E "FETCH_SIZE|WRITE_SIZE|SQ_WAIT_ANY |SQ_WAVE_CYCLES|SQ_INSTS_VALU |SQ_INSTS_VMEM"
cp gpurun_out/profiles_r06k/*reverb* gpurun_out/r06k/
