#!/bin/bash
# Run on the GPU box (via gpurun): the round-6 investigation of the headline kernel's launch-to-launch spread.
#   1. bench.py --workload cfg3 under rocprofv3 --kernel-trace, the per-dispatch CSV kept (start, end, duration in launch order)
#   2. the same under --pmc GRBM_GUI_ACTIVE (a pass of its own), per dispatch
#   3. tools/spread_probe.hip: the product kernel body with per-wavefront clock stamps, over the variants that separate the causes
# Everything lands in gpurun_out/spread/.
set -u
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/spread
mkdir -p $out
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp -Wno-unused-value -Wno-unused-result"
mkdir -p tools/bin
[ -x tools/bin/spread_probe ] || /opt/rocm/bin/hipcc $F tools/spread_probe.hip -o tools/bin/spread_probe -lpthread
[ -x tools/bin/spread_probe_noturns ] || /opt/rocm/bin/hipcc $F -DMLGPU_CHAIN_TURNS=0 tools/spread_probe.hip -o tools/bin/spread_probe_noturns -lpthread
[ -x tools/bin/spread_probe_strict ] || /opt/rocm/bin/hipcc $F -DMLGPU_SVF_STRICT=1 tools/spread_probe.hip -o tools/bin/spread_probe_strict -lpthread

if [ "${SKIP_ROCPROF:-0}" != 1 ]; then
  rm -rf $out/trace; mkdir -p $out/trace
  ( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python $root/bench.py --workload cfg3 --no-cpu-baseline --no-extras > $out/bench_under_trace.json 2> $out/trace/stderr.log )
  f=$(find $out/trace -name '*kernel_trace.csv' | head -1)
  if [ -n "$f" ]; then
    python - "$f" $out/cfg3_dispatch_trace.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if 'chain_kernel' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
with open(sys.argv[2], 'w') as f:
    f.write('dispatch,start_us,end_us,duration_us,gap_before_us\n')
    prev = None
    for i, r in enumerate(rows):
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        f.write(f"{i},{(s - t0) / 1e3:.2f},{(e - t0) / 1e3:.2f},{(e - s) / 1e3:.2f},{((s - prev) / 1e3) if prev else 0:.2f}\n")
        prev = e
print(len(rows), 'dispatches of chain_kernel')
PY
  fi
  rm -rf $out/pmc; mkdir -p $out/pmc
  ( cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/pmc -- python $root/bench.py --workload cfg3 --no-cpu-baseline --no-extras > $out/bench_under_pmc.json 2> $out/pmc/stderr.log )
  f=$(find $out/pmc -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then
    python - "$f" $out/cfg3_dispatch_cycles.csv <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'chain_kernel' in r['Kernel_Name'] and r['Counter_Name'] == 'GRBM_GUI_ACTIVE']
with open(sys.argv[2], 'w') as f:
    f.write('dispatch,GRBM_GUI_ACTIVE\n')
    for i, r in enumerate(sorted(rows, key=lambda r: int(r['Dispatch_Id']))):
        f.write(f"{i},{float(r['Counter_Value']):.0f}\n")
print(len(rows), 'counter rows')
PY
  fi
  find $out/trace $out/pmc -name '*.csv' -size +1M -delete
fi

P=tools/bin/spread_probe
run() { echo "== $*"; "$@" --out $out | tee -a $out/summary.txt; }
: > $out/summary.txt
run $P --tag base --per_step 600
run $P --tag plain --plain --per_step 600
run $P --tag step25 --per_step 25
run $P --tag nbuf1 --nbuf 1 --per_step 600
run $P --tag nbuf4 --nbuf 4 --per_step 600
run $P --tag nbuf8 --nbuf 8 --per_step 600
run $P --tag T15 --T 15 --launches 1200 --per_step 1200
run $P --tag T60 --T 60 --launches 300 --per_step 300
run $P --tag T8 --T 8 --launches 2000 --per_step 2000
run $P --tag gap50 --gap_us 50
run $P --tag gap300 --gap_us 300
run $P --tag gap2000 --gap_us 2000 --launches 300
run $P --tag synceach --sync_each
run $P --tag V245760 --V 245760 --per_step 600
run $P --tag V196608 --V 196608 --per_step 600
run $P --tag V131072 --V 131072 --per_step 600
run $P --tag V524288_T15 --V 524288 --T 15 --per_step 600
run tools/bin/spread_probe_noturns --tag noturns --per_step 600
run tools/bin/spread_probe_strict --tag strict --per_step 600
run $P --tag base_again --per_step 600
run $P --tag long --launches 3000 --per_step 3000 --dump 0
ls -la $out | head -80
