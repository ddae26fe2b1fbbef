#!/bin/bash
# Copy what tools/profile_round5.sh left under gpurun_out/r05f/ into profiles/ under the names profiles/README.md lists.
set -eu
cd "$(dirname "$0")/.."
S=gpurun_out/r05f
D=profiles
cpset() {  # <dir> <workload in file names> <name in profiles>
  for f in $S/$1/r05_$2_*; do b=$(basename $f); cp $f $D/${b/r05_$2/$3}; done
}
for w in cfg3 cfg4; do cpset profiles_main $w r05_$w; done
for w in cfg5 cfg5full; do cpset profiles_graphs $w r05_$w; done
for w in synth synthrows events resample; do cpset profiles_wide $w r05_$w; done
cpset profiles_cfg2_1GiB cfg2 r05_cfg2_1GiB
cpset profiles_cfg2_32MiB cfg2 r05_cfg2_32MiB
cpset profiles_strings_windows strings r05w_strings_windows
cpset profiles_strings_transposed strings r05t_strings_transposed
cp $S/pmc_workloads.json $D/pmc_workloads.json
cp $S/lines.txt $D/r05_lines.txt
for w in cfg3 cfg4 cfg5 cfg5full cfg2 synth synthrows; do cp $S/${w}_line.json $D/r05_${w}_line_with_pmc.json; done
for l in 1 2; do cp $S/strings_layout${l}_line.json $D/r05_strings_layout${l}_line_with_pmc.json; done
cp $S/default_bench.json $D/r05_default_bench.json
cp $S/rt_kernel_stats.csv $D/r05_rt_kernel_stats.csv
cp $S/rt_bench.json $D/r05_rt_bench.json
cp $S/valu_calibration.txt $D/r05_valu_calibration.txt
cp $S/gpu_tests.txt $D/r05_gpu_tests.txt
python tools/summarize_profiles.py $D r05 > $D/r05_summary.md
python tools/check_pmc_fresh.py
ls $D | grep -c r05
# (the records name the files as they are called in profiles/)
python - <<'PY'
import json
p = 'profiles/pmc_workloads.json'; d = json.load(open(p))
for k, n in {'cfg2:65536x1': 'r05_cfg2_32MiB', 'cfg2:4194304x1': 'r05_cfg2_1GiB', 'strings:delay_windows=1:262144x16': 'r05w_strings_windows',
             'strings:delay_windows=2:262144x16': 'r05t_strings_transposed'}.items():
    if d['workloads'].get(k, {}).get('files', '').startswith('r05'):
        d['workloads'][k]['files'] = n + '_{traffic.json,pmc.txt}'
json.dump(d, open(p, 'w'), indent=1)
PY
