cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
( time python bench.py ) > gpurun_out/r05/default_bench.json 2> gpurun_out/r05/default_bench.err
tail -c 600 gpurun_out/r05/default_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/default_bench.json').read().strip().splitlines()[-1])
r=d['roofline']
print({k:v for k,v in r.items() if not isinstance(v,(dict,list))})
print(d.get('parity_512'))
print(d['other_configs'].get('seconds'), d['other_configs'].get('rt'))
PY
python bench.py --workload rt --steps 5 --warmup 1 > gpurun_out/r05/rt_bench.json 2> gpurun_out/r05/rt_bench.err; tail -c 300 gpurun_out/r05/rt_bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r05/rt_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['rt'])"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.4g voice-samples/s  %.3f ms per block' % (d['value'], r['kernel_ms']))"; }
for w in synth synthfused; do for m in kernel graph; do echo "## $w  sum=$m: $(MLGPU_BENCH_MIXDOWN=$m python bench.py --no-cpu-baseline --workload $w 2>/dev/null | tail -1 | line)"; done; done
echo "## synth --two-streams sum=graph: $(MLGPU_BENCH_MIXDOWN=graph python bench.py --no-cpu-baseline --workload synth --two-streams 2>/dev/null | tail -1 | line)"
