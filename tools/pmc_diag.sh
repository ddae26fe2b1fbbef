#!/bin/bash
# Run on the GPU box (via gpurun): issue-side PMC counters of a bench workload's dominant kernel, two passes of eight.
#   tools/pmc_diag.sh <tag> <workload> [bench.py options]      -> gpurun_out/<tag>_<workload>_pmc_diag.txt
set -u
tag=$1; w=$2; shift 2
export TMPDIR=/tmp
root=$PWD
mkdir -p $root/gpurun_out
s=/tmp/pmcdiag_${tag}_$w; rm -rf $s; mkdir -p $s
P1="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH"
P3="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM"
i=0
for pass in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  ( cd /tmp && rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $s/p$i -- python $root/bench.py --workload $w "$@" --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $s/p$i.log )
done
{ echo "# rocprofv3 --pmc (three passes) --kernel-trace -- python bench.py --workload $w $* --steps 2 --warmup 1 --no-cpu-baseline ($tag, MI355X)";
  python $root/tools/pmc_summary.py pmc $s/p1 $s/p2 $s/p3; } > $root/gpurun_out/${tag}_${w}_pmc_diag.txt 2>&1
grep -A30 "mlgpu_graph_kernel\|chain_kernel\|cascade" $root/gpurun_out/${tag}_${w}_pmc_diag.txt | head -80
