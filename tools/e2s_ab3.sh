cd $GRAFT_REPO_ROOT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r['kernel'], r['kernel_ms'], r['frac'])"; }
timeout 600 python -m pytest tests/test_gpu_events.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2; do for nb in "" 1; do
  if [ -n "$nb" ]; then export MLGPU_E2S_NO_BLOCKS=1; else unset MLGPU_E2S_NO_BLOCKS; fi
  for rows in 0,1 0,1,2,3,4,5,6,7; do
    echo "## no_blocks=$nb rows $rows: sparse events / none"; MLGPU_EVENT_ROWS=$rows python bench.py --no-cpu-baseline --workload events --warmup 5 2>/dev/null | tail -1 | line
    MLGPU_BENCH_EVENTS_UNTIL=24 MLGPU_EVENT_ROWS=$rows python bench.py --no-cpu-baseline --workload events --warmup 5 2>/dev/null | tail -1 | line
  done
  echo "## no_blocks=$nb synth, synth --two-streams"
  python bench.py --no-cpu-baseline --workload synth 2>/dev/null | tail -1 | line
  python bench.py --no-cpu-baseline --workload synth --two-streams 2>/dev/null | tail -1 | line
done; done
