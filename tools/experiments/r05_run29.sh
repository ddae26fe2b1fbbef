cd $GRAFT_REPO_ROOT
root=$PWD
export TMPDIR=/tmp
pmc() { tag=$1; shift; ctrs=$1; shift
  out=$root/gpurun_out/r05/pmc_$tag; rm -rf $out; mkdir -p $out
  ( cd /tmp && env "$@" rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -- python $root/bench.py --no-cpu-baseline --workload strings --steps 2 --warmup 1 > /dev/null 2> $out/stderr.log )
  f=$(find $out -name '*counter_collection.csv' | head -1)
  python - "$f" "$tag" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    if 'graph_kernel' in k:
        print(sys.argv[2], {c:round(sum(x)/len(x)) for c,x in v.items()})
PY
}
for l in 1 2; do
pmc s${l}a "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY" MLGPU_DELAY_WINDOWS=$l
pmc s${l}b "FETCH_SIZE" MLGPU_DELAY_WINDOWS=$l
pmc s${l}c "WRITE_SIZE" MLGPU_DELAY_WINDOWS=$l
pmc s${l}d "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" MLGPU_DELAY_WINDOWS=$l
pmc s${l}e "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TA_DATA_STALL_CYCLES_sum" MLGPU_DELAY_WINDOWS=$l
done
