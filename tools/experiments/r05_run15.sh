cd $GRAFT_REPO_ROOT
root=$PWD
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_events.py tests/test_gpu_dropin.py tests/test_gpu_sequence.py -x -q -m gpu 2>&1 | tail -4
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.4g voice-samples/s  %.3f ms per block (wall %.3f ms per step of %d)' % (d['value'], r['kernel_ms'], d['ms_per_step'], d['config']['launches_per_step']))"; }
for i in 1 2; do echo "## synth: $(timeout 300 python bench.py --no-cpu-baseline --workload synth 2>/dev/null | tail -1 | line)"; done
out=$root/gpurun_out/r05/prof_synth2; rm -rf $out; mkdir -p $out
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $root/bench.py --no-cpu-baseline --workload synth --steps 5 --warmup 2 > $out/bench.json 2> $out/stderr.log )
f=$(find $out -name '*_kernel_stats.csv' | head -1); head -4 $f
python - $(find $out -name '*_kernel_trace.csv' | head -1) <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'graph_kernel' in r['Kernel_Name'] or 'e2s_ctl' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[20]['Start_Timestamp'])
for r in rows[20:30]:
    print(r['Kernel_Name'][:30].ljust(30), (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-t0)/1e3)
PY
find $out -name '*.csv' -size +2M -delete
