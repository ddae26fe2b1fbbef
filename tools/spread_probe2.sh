#!/bin/bash
# second pass of the spread investigation (tools/spread_probe.sh was the first): the store stream alone and the arithmetic alone with
# the same stamps, the turn period, and the same base case on (probably) another box
set -u
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/spread2
mkdir -p $out
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp -Wno-unused-value -Wno-unused-result"
for s in 9 10 11 12 14 15; do
  [ -x tools/bin/spread_probe_shift$s ] || /opt/rocm/bin/hipcc $F -DMLGPU_CHAIN_TURN_SHIFT=$s tools/spread_probe.hip -o tools/bin/spread_probe_shift$s -lpthread
done
ls /sys/class/drm/ > $out/sys_class_drm.txt 2>&1
rocm-smi --showpower --showclocks --showbus > $out/rocm_smi.txt 2>&1
P=tools/bin/spread_probe
run() { echo "== $*"; "$@" --out $out | tee -a $out/summary.txt; }
: > $out/summary.txt
run $P --tag base --per_step 600
run $P --tag gain --gain --per_step 600
run $P --tag l2only --l2only --per_step 600
run $P --tag gain_l2only --gain --l2only --per_step 600
for s in 9 10 11 12 14 15; do run tools/bin/spread_probe_shift$s --tag shift$s --per_step 600 --dump 1; done
run tools/bin/spread_probe_noturns --tag noturns --per_step 600 --dump 1
run $P --tag base_again --per_step 600
run $P --tag gap300 --gap_us 300
run $P --tag long --launches 3000 --per_step 3000 --dump 0
head -3 $out/base_sysfs.csv
cat $out/rocm_smi.txt | head -40
