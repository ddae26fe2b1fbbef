#!/bin/bash
# round 6, run 9: the default line with live counters and the allpass4 leg; rt across two ranks on one GPU
export TMPDIR=/tmp
mkdir -p gpurun_out/r06i
( time timeout 900 python bench.py > gpurun_out/r06i/default_bench.json 2> gpurun_out/r06i/default_bench.err ) 2> gpurun_out/r06i/default_bench.time
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06i/default_bench.json').read().strip().splitlines()[-1])
r=d['roofline']
print({k:v for k,v in list(r.items())[:26]})
print({k:(v.get('error') or 'ok') for k,v in d['other_configs'].items() if isinstance(v,dict)})
PY
cat gpurun_out/r06i/default_bench.time | tail -3; tail -3 gpurun_out/r06i/default_bench.err
timeout 600 python bench.py --workload rt --gpus 2 --oversubscribe --steps 4 --warmup 1 > gpurun_out/r06i/rt_group.json 2> gpurun_out/r06i/rt_group.err; tail -3 gpurun_out/r06i/rt_group.err; python -c "
import json; d=json.loads(open('gpurun_out/r06i/rt_group.json').read().strip().splitlines()[-1]); print(d.get('rt_group'))"
timeout 600 python bench.py --workload rt --gpus 2 --oversubscribe --launcher threads --steps 4 --warmup 1 > gpurun_out/r06i/rt_group_threads.json 2> gpurun_out/r06i/rt_group_threads.err; tail -3 gpurun_out/r06i/rt_group_threads.err; python -c "
import json; d=json.loads(open('gpurun_out/r06i/rt_group_threads.json').read().strip().splitlines()[-1]); print(d.get('rt_group'))"
