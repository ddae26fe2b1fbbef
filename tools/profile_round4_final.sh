# The round's closing call: every workload re-profiled at the frozen device code WITH the counter passes (tools/profile_round4.sh ran
# the widened rows without them), the bench lines once more with the fresh records in place, the default bench line, the GPU test suite.
# The A/B lines, per-wavefront clocks and node costs of profile_round4.sh are not repeated (same device code: tools/check_pmc_fresh.py).
# Output under gpurun_out/r04f/ (merged over what is there); tools/collect_round4.sh copies it into profiles/.
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f; mkdir -p $O
B="python bench.py --no-cpu-baseline"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r['kernel'], r['kernel_ms'], r['frac'], r['bound'], r.get('frac_of_ceiling'), (r.get('valu') or {}).get('busy_frac'))"; }
fin() { rm -rf $O/$1; mv gpurun_out/profiles_r04 $O/$1; }
tools/gpu_profile_all.sh r04 cfg3 cfg4 > $O/prof_main.log 2>&1; fin profiles_main
tools/gpu_profile_all.sh r04 cfg5 cfg5full > $O/prof_graphs.log 2>&1; fin profiles_graphs
EXTRA="--voices 4194304" tools/gpu_profile_all.sh r04 cfg2 > $O/prof_cfg2.log 2>&1; fin profiles_cfg2_1GiB
tools/gpu_profile_all.sh r04 cfg2 > $O/prof_cfg2s.log 2>&1; fin profiles_cfg2_32MiB
tools/gpu_profile_all.sh r04 synth synthfused events resample > $O/prof_wide.log 2>&1; fin profiles_wide
MLGPU_DELAY_WINDOWS=1 tools/gpu_profile_all.sh r04 strings > $O/prof_strings.log 2>&1; fin profiles_strings_windows
cp profiles/pmc_workloads.json $O/pmc_workloads.json
python tools/check_pmc_fresh.py > $O/pmc_fresh.txt 2>&1
for w in cfg3 cfg4 cfg5 cfg5full cfg2; do $B --workload $w 2>/dev/null | tail -1 > $O/${w}_line.json; echo "## $w"; cat $O/${w}_line.json | line; done > $O/lines.txt 2>&1
python bench.py 2>/dev/null | tail -1 > $O/default_bench.json
python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > $O/gpu_tests.txt
for dd in $O/profiles_*; do python tools/summarize_profiles.py $dd r04 > $dd/summary.md 2>/dev/null; done
cat $O/profiles_*/summary.md | grep -v "^|---\|^| bench file"
cat $O/lines.txt $O/pmc_fresh.txt $O/gpu_tests.txt
