#!/bin/bash
# Copy what tools/profile_round4.sh left under gpurun_out/r04f/ into profiles/ under the names profiles/README.md lists.
set -eu
cd "$(dirname "$0")/.."
S=gpurun_out/r04f
D=profiles
cpset() {  # <dir> <workload in file names> <name in profiles>
  for f in $S/$1/r04_$2_*; do b=$(basename $f); cp $f $D/${b/r04_$2/$3}; done
}
for w in cfg3 cfg4; do cpset profiles_main $w r04_$w; done
for w in cfg5 cfg5full; do cpset profiles_graphs $w r04_$w; done   # (re-profiled with the default kernel form: no online tuning trials among the calls)
for w in synth synthfused events resample; do cpset profiles_wide $w r04_$w; done
cpset profiles_cfg2_1GiB cfg2 r04_cfg2_1GiB
cpset profiles_cfg2_32MiB cfg2 r04_cfg2_32MiB
cpset profiles_strings_windows strings r04w_strings_windows
cp $S/pmc_workloads.json $D/pmc_workloads.json   # (the box starts from the committed file: commit it before a partial re-profile)
for f in lines graph_steps wave_clock node_costs; do cp $S/$f.txt $D/r04_$f.txt; done
for w in cfg3 cfg4 cfg5 cfg5full cfg2; do cp $S/${w}_line.json $D/r04_${w}_line_with_pmc.json; done
cp $S/default_bench.json $D/r04_default_bench.json
cp $S/cfg3_sustained.json $D/r04_cfg3_sustained.json
python tools/summarize_profiles.py $D r04 > $D/r04_summary.md
python tools/check_pmc_fresh.py
ls $D | grep -c r04
# (the records name the files as they are called in profiles/)
python - <<'PY'
import json
p = 'profiles/pmc_workloads.json'; d = json.load(open(p))
for k, n in {'cfg2:65536x1': 'r04_cfg2_32MiB', 'cfg2:4194304x1': 'r04_cfg2_1GiB', 'strings:delay_windows=1:262144x16': 'r04w_strings_windows'}.items():
    if d['workloads'].get(k, {}).get('files', '').startswith('r04'):
        d['workloads'][k]['files'] = n + '_{traffic.json,pmc.txt}'
json.dump(d, open(p, 'w'), indent=1)
PY
