cd $GRAFT_REPO_ROOT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r['kernel'], r['kernel_ms'], r['frac'])"; }
for rep in 1 2; do for lib in base new; do
  if [ $lib = base ]; then export MLGPU_LIB=$PWD/tools/bin/libmlgpu_base.so; else unset MLGPU_LIB; fi
  for rows in 0,1 0,1,2,3,4,5,6,7; do
    echo "## $lib rows $rows: sparse events"; MLGPU_EVENT_ROWS=$rows python bench.py --no-cpu-baseline --workload events --warmup 5 2>/dev/null | tail -1 | line
  done
  echo "## $lib synth"; python bench.py --no-cpu-baseline --workload synth 2>/dev/null | tail -1 | line
done; done
unset MLGPU_LIB
python -m pytest tests/test_gpu_events.py -x -q -m gpu 2>&1 | tail -2
