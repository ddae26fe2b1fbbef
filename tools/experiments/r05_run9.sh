cd $GRAFT_REPO_ROOT
root=$PWD
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_parity.py tests/test_gpu_events.py -x -q -m gpu 2>&1 | tail -4
pmc() { tag=$1; shift; ctrs=$1; shift
  out=$root/gpurun_out/r05/pmc_$tag; rm -rf $out; mkdir -p $out
  ( cd /tmp && env "$@" rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -- python $root/bench.py --no-cpu-baseline --workload synthfused --steps 2 --warmup 1 > /dev/null 2> $out/stderr.log )
  f=$(find $out -name '*counter_collection.csv' | head -1)
  python - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    if 'graph_kernel' in k or 'e2s' in k:
        print(k, {c:(sum(x)/len(x)) for c,x in v.items()})
PY
}
pmc a "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_LDS" MLGPU_BENCH_MIXDOWN=graph
pmc b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" MLGPU_BENCH_MIXDOWN=graph
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.4g voice-samples/s  %.3f ms per block (wall %.3f ms per step of %d)' % (d['value'], r['kernel_ms'], d['ms_per_step'], d['config']['launches_per_step']))"; }
echo "## synthfused sum=graph: $(MLGPU_BENCH_MIXDOWN=graph timeout 300 python bench.py --no-cpu-baseline --workload synthfused 2>/dev/null | tail -1 | line)"
echo "## cfg3: $(timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | line)"
echo "## cfg5: $(timeout 300 python bench.py --no-cpu-baseline --workload cfg5 2>/dev/null | tail -1 | line)"
