cd $GRAFT_REPO_ROOT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r['kernel'], r['kernel_ms'], r['frac'])"; }
for v in b4 b2 b1; do
  export MLGPU_LIB=$GRAFT_REPO_ROOT/tools/bin/libmlgpu_$v.so
  for rows in 0,1 0,1,2,3,4,5,6,7; do
    echo "## $v rows $rows: sparse events / none"; MLGPU_EVENT_ROWS=$rows python bench.py --no-cpu-baseline --workload events --warmup 5 2>/dev/null | tail -1 | line
    MLGPU_BENCH_EVENTS_UNTIL=24 MLGPU_EVENT_ROWS=$rows python bench.py --no-cpu-baseline --workload events --warmup 5 2>/dev/null | tail -1 | line
  done
  echo "## $v synth, synth --two-streams"
  python bench.py --no-cpu-baseline --workload synth 2>/dev/null | tail -1 | line
  python bench.py --no-cpu-baseline --workload synth --two-streams 2>/dev/null | tail -1 | line
done
