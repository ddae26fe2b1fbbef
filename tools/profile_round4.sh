# Round 4's evidence, in one gpurun call: every workload re-profiled at the final code (kernel stats, five PMC passes for BASELINE's
# configs, bench lines), the round's A/B lines (knobs of the graph generator, same box), per-wavefront clocks, node costs.
# Output under gpurun_out/r04f/; tools/collect_round4.sh copies what is to be judged into profiles/.
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f; rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r['kernel'], r['kernel_ms'], r['frac'], r['bound'], r.get('frac_of_ceiling'), (r.get('valu') or {}).get('busy_frac'))"; }

tools/gpu_profile_all.sh r04 cfg3 cfg4 cfg5 cfg5full > $O/prof_main.log 2>&1
mv gpurun_out/profiles_r04 $O/profiles_main
EXTRA="--voices 4194304" tools/gpu_profile_all.sh r04 cfg2 > $O/prof_cfg2.log 2>&1
mv gpurun_out/profiles_r04 $O/profiles_cfg2_1GiB
tools/gpu_profile_all.sh r04 cfg2 > $O/prof_cfg2s.log 2>&1
mv gpurun_out/profiles_r04 $O/profiles_cfg2_32MiB
PMC=0 tools/gpu_profile_all.sh r04 synth synthfused events resample > $O/prof_wide.log 2>&1
mv gpurun_out/profiles_r04 $O/profiles_wide
PMC=0 MLGPU_DELAY_WINDOWS=1 tools/gpu_profile_all.sh r04 strings > $O/prof_strings.log 2>&1
mv gpurun_out/profiles_r04 $O/profiles_strings_windows
cp profiles/pmc_workloads.json $O/pmc_workloads.json

# the bench lines once more with the fresh PMC records in place (bound, busy_frac, traffic)
for w in cfg3 cfg4 cfg5 cfg5full cfg2; do $B --workload $w 2>/dev/null | tail -1 > $O/${w}_line.json; echo "## $w"; cat $O/${w}_line.json | line; done > $O/lines.txt 2>&1
python bench.py 2>/dev/null | tail -1 > $O/default_bench.json

# the graph generator's round-4 steps, one box: each knob off against everything on
{ echo "# tools/graph_ab.py, 262144 voices x 16 DSPVectors, one box; '-' = the round-4 defaults";
  AB_REPS=60 python tools/graph_ab.py cfg5,cfg5full,synthpitch - MLGPU_GRAPH_TURNS=0 MLGPU_GRAPH_PREFETCH=0 MLGPU_GRAPH_LOCK_OSC=0 MLGPU_GRAPH_TURNS=0,MLGPU_GRAPH_PREFETCH=0,MLGPU_GRAPH_LOCK_OSC=0 2>&1 | grep "round 1";
  echo "# the round-3 library on the same box";
  MLGPU_LIB=$PWD/tools/bin/libmlgpu_r03.so AB_REPS=60 python tools/graph_ab.py cfg5,cfg5full,synthpitch - 2>&1 | grep "round 1"; } > $O/graph_steps.txt 2>&1

for w in cfg5 cfg5full; do python tools/wave_clock.py $w; MLGPU_GRAPH_TURNS=0 python tools/wave_clock.py $w; done > $O/wave_clock.txt 2>&1
python tools/node_costs.py 2 10 > $O/node_costs.txt 2>&1
$B --workload cfg3 --sustained $O/cfg3_sustained.json --sustained-seconds 20 2>/dev/null | tail -1 > $O/cfg3_sustained_line.json
for dd in $O/profiles_*; do python tools/summarize_profiles.py $dd r04 > $dd/summary.md 2>/dev/null; done
cat $O/profiles_*/summary.md | grep -v "^|---\|^| bench file"
cat $O/lines.txt; cat $O/graph_steps.txt
