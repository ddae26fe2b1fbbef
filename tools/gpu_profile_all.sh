#!/bin/bash
# Run on the GPU box (via gpurun): per workload, rocprofv3 kernel stats + six PMC passes (SQ set, FETCH_SIZE, WRITE_SIZE, two of instruction classes,
# one of issue occupancy: SQ_ACTIVE_INST_VALU2, SQ_BUSY_CU_CYCLES, scalar and LDS activity;
# counters are collected with --kernel-trace only, in passes of their own) + a plain bench line.
#   tools/gpu_profile_all.sh <tag> cfg3 cfg4 cfg5
# Output: gpurun_out/profiles_<tag>/<tag>_<w>_{bench.json,bench_under_rocprof.json,kernel_stats.csv,pmc.txt} and pmc_traffic.json
# EXTRA="--voices N ..." in the environment is passed to every bench.py call (e.g. the 1 GiB config-2 case). PMC=0 skips the counter passes
# (kernel stats and bench lines only).
set -u
export MLGPU_BENCH_CHILD=1   # (bench.py does not start counter passes of its own inside these runs)
EXTRA=${EXTRA:-}
tag=$1; shift
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/profiles_$tag
rm -rf $out; mkdir -p $out
fetchdirs=""; writedirs=""
EXTRA0=$EXTRA
for w in "$@"; do
  EXTRA=$EXTRA0; [ $w = cfg3 ] && EXTRA="$EXTRA0 --no-extras"   # (the default workload's extra legs - the other configs, the paced target - are not this profile's subject)
  scratch=/tmp/prof_${tag}_$w; rm -rf $scratch; mkdir -p $scratch
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $scratch/stats -- python $root/bench.py --workload $w $EXTRA --no-cpu-baseline > $out/${tag}_${w}_bench_under_rocprof.json 2> $scratch/stats.log )
  f=$(find $scratch/stats -name '*_kernel_stats.csv' | head -1)
  { echo "# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --workload $w --no-cpu-baseline   ($tag, MI355X)"; [ -n "$f" ] && head -6 $f; } > $out/${tag}_${w}_kernel_stats.csv
  [ "${PMC:-1}" = "0" ] || for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" \
              "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU" \
              "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_INSTS_SMEM" \
              "SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS"; do
    name=$(echo $pass | cut -d' ' -f1)
    ( cd /tmp && rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $scratch/pmc_$name -- python $root/bench.py --workload $w $EXTRA --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $scratch/pmc_$name.log )
  done
  { echo "# rocprofv3 --pmc <counters> --kernel-trace (separate passes: SQ set, FETCH_SIZE, WRITE_SIZE) -- python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline ($tag, MI355X)";
    python $root/tools/pmc_summary.py pmc $scratch/pmc_SQ_WAVE_CYCLES $scratch/pmc_FETCH_SIZE $scratch/pmc_WRITE_SIZE $scratch/pmc_SQ_INSTS_VALU_ADD_F32 $scratch/pmc_SQ_INSTS_VALU_ADD_F64 $scratch/pmc_SQ_ACTIVE_INST_VALU2; } > $out/${tag}_${w}_pmc.txt 2>&1
  python $root/tools/pmc_summary.py traffic $scratch/pmc_FETCH_SIZE $scratch/pmc_WRITE_SIZE $out/${tag}_${w}_traffic.json > /dev/null 2>&1
  python $root/tools/pmc_workloads.py $out/pmc_workloads.json "$(python $root/bench.py --workload $w $EXTRA --print-case)@$out/${tag}_${w}" > /dev/null 2>&1
  python $root/tools/pmc_workloads.py $root/profiles/pmc_workloads.json "$(python $root/bench.py --workload $w $EXTRA --print-case)@$out/${tag}_${w}" > /dev/null 2>&1   # the bench line below reads the fresh record
  MLGPU_BENCH_CHILD= python $root/bench.py --workload $w $EXTRA $( [ $w = cfg3 ] || echo --no-cpu-baseline ) 2>/dev/null | tail -1 > $out/${tag}_${w}_bench.json
  echo "== $w"; sed -n 2,3p $out/${tag}_${w}_kernel_stats.csv | cut -c1-200; python -c "import sys,json; d=json.loads(open('$out/${tag}_${w}_bench.json').read()); print(d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'])"
done
python - <<PY
import json, glob
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py --workload <w> --steps 2 --warmup 1",
       "correction": "bytes = KiB*1024; gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported", "kernels": {}}
for f in sorted(glob.glob("$out/*_traffic.json")):
    d = json.load(open(f))
    for k, v in d.get("kernels", {}).items():
        if v["launches"] >= 8:
            res["kernels"][k] = v
json.dump(res, open("$out/pmc_traffic.json", "w"), indent=1)
print(list(res["kernels"]))
PY
