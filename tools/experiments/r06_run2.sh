#!/bin/bash
# round 6, run 2: baselines of the many-ring graphs at today's layouts
export TMPDIR=/tmp
mkdir -p gpurun_out/r06b
for v in 16384 131072; do for l in 0 1; do
  MLGPU_DELAY_WINDOWS=$l timeout 300 python bench.py --workload allpass4 --voices $v --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('allpass4 V=$v layout=$l', 'ms', round(r['kernel_ms'],3), 'frac', round(r['frac'],3), 'p10/50/90', r.get('kernel_ms_p10'), r.get('kernel_ms_p50'), r.get('kernel_ms_p90'))"
done; done 2>&1 | tee gpurun_out/r06b/allpass4.txt
for v in 65536 131072 262144; do
  timeout 600 python bench.py --workload reverb --voices $v --no-cpu-baseline --steps 5 --warmup 1 2>gpurun_out/r06b/reverb_$v.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('reverb V=$v', 'ms', round(r['kernel_ms'],3), 'frac', round(r['frac'],3), 'value', d['value'])"
done 2>&1 | tee gpurun_out/r06b/reverb.txt
timeout 600 python bench.py > gpurun_out/r06b/default_bench.json 2> gpurun_out/r06b/default_bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06b/default_bench.json').read().strip().splitlines()[-1])
r=d['roofline']
print({k:v for k,v in r.items() if not isinstance(v,dict)})
print({k:(v.get('error') or 'ok') for k,v in d['other_configs'].items() if isinstance(v,dict)})
print(d.get('parity_512'))
PY
