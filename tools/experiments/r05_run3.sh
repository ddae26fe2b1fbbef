cd $GRAFT_REPO_ROOT
root=$PWD
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
prof() { # tag, env..., 
  tag=$1; shift
  out=$root/gpurun_out/r05/prof_$tag; rm -rf $out; mkdir -p $out
  ( cd /tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $root/bench.py --no-cpu-baseline --workload synthfused --steps 5 --warmup 2 > /dev/null 2> $out/stderr.log )
  f=$(find $out -name '*_kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $root/gpurun_out/r05/${tag}_kernel_stats.csv && head -6 $f
  find $out -name '*.csv' -size +2M -delete
}
prof sf_graph MLGPU_BENCH_MIXDOWN=graph
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
pmc b "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" MLGPU_BENCH_MIXDOWN=graph
