#!/bin/bash
# Copy what tools/profile_round3.sh left under gpurun_out/r03f/ into profiles/ under the names profiles/README.md lists.
set -eu
cd "$(dirname "$0")/.."
S=gpurun_out/r03f
D=profiles
cpset() {  # <dir> <workload in file names> <name in profiles>
  for f in $S/$1/r03_$2_*; do b=$(basename $f); cp $f $D/${b/r03_$2/$3}; done
}
for w in cfg3 cfg4 cfg5 cfg5full synth synthfused events resample; do cpset profiles_main $w r03_$w; done
cpset profiles_cfg2_1GiB cfg2 r03_cfg2_1GiB
cpset profiles_cfg2_32MiB cfg2 r03_cfg2_32MiB
cpset profiles_cfg4_262144 cfg4 r03_cfg4_V262144
cpset profiles_strings_windows strings r03w_strings_windows
cp $S/pmc_workloads.json $D/pmc_workloads.json
for f in bankbench instbench cfg4_forms strict_svf jit_maxilp synth_mixdown multi_gpu_launch_paths node_costs two_streams_lines osc_trips; do cp $S/$f.txt $D/r03_$f.txt; done
cp $S/cascade_lab.txt $D/r03_cascade_lanes.txt
cp $S/cascade_lab_pmc_mode0.txt $D/r03_cascade_lab_pmc_streams.txt
cp $S/cascade_lab_pmc_mode1.txt $D/r03_cascade_lab_pmc_no_hbm.txt
cp $S/cfg3_sustained.json $D/r03_cfg3_sustained.json
cp $S/cfg3_steps3000_bench.json $D/r03_cfg3_steps3000_bench.json
python tools/summarize_profiles.py $D r03 > $D/r03_summary.md
ls $D | grep -c r03
