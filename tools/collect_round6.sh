#!/bin/bash
# Copy what tools/profile_round6.sh left under gpurun_out/r06f/ into profiles/ under the names profiles/README.md lists.
set -eu
cd "$(dirname "$0")/.."
S=gpurun_out/r06f
D=profiles
cpset() {  # <dir> <workload in file names> <name in profiles>
  for f in $S/$1/r06_$2_*; do b=$(basename $f); cp $f $D/${b/r06_$2/$3}; done
}
for w in cfg3 cfg4; do cpset profiles_main $w r06_$w; done
for w in cfg5 cfg5full; do cpset profiles_graphs $w r06_$w; done
for w in synth synthrows events resample; do cpset profiles_wide $w r06_$w; done
cpset profiles_cfg2_1GiB cfg2 r06_cfg2_1GiB
cpset profiles_cfg2_32MiB cfg2 r06_cfg2_32MiB
cpset profiles_strings_best strings r06_strings
cpset profiles_allpass4_best allpass4 r06_allpass4
cpset profiles_reverb reverb r06_reverb
cp $S/pmc_workloads.json $D/pmc_workloads.json
cp $S/lines.txt $D/r06_lines.txt
for w in cfg3 cfg4 cfg5 cfg5full cfg2 synth synthrows strings allpass4 events resample reverb; do [ -f $S/${w}_line.json ] && cp $S/${w}_line.json $D/r06_${w}_line_with_pmc.json; done
cp $S/default_bench.json $D/r06_default_bench.json
cp $S/rt_kernel_stats.csv $D/r06_rt_kernel_stats.csv
cp $S/rt_bench.json $D/r06_rt_bench.json
cp $S/rt_group_2ranks_one_gpu.json $D/r06_rt_group_2ranks_one_gpu.json
cp $S/multi_engine_test.txt $D/r06_multi_engine_test.txt
cp $S/cfg3_dispatch_trace.csv $D/r06_cfg3_dispatch_trace_closing.csv
cp $S/valu_calibration.txt $D/r06_valu_calibration.txt
cp $S/gpu_tests.txt $D/r06_gpu_tests.txt
{ cat $S/ring_layout_soak.txt; grep '^#' $D/r06_ring_layout_soak.txt || true; } > $D/r06_ring_layout_soak.txt.new && mv $D/r06_ring_layout_soak.txt.new $D/r06_ring_layout_soak.txt   # (the notes below the three lines stay)
python tools/summarize_profiles.py $D r06 > $D/r06_summary.md
python tools/check_pmc_fresh.py
ls $D | grep -c r06
python - <<'PY'
import json
p = 'profiles/pmc_workloads.json'; d = json.load(open(p))
for k, n in {'cfg2:65536x1': 'r06_cfg2_32MiB', 'cfg2:4194304x1': 'r06_cfg2_1GiB'}.items():
    if d['workloads'].get(k, {}).get('files', '').startswith('r06'):
        d['workloads'][k]['files'] = n + '_{traffic.json,pmc.txt}'
json.dump(d, open(p, 'w'), indent=1)
PY
