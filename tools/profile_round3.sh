# Round 3's evidence, in one gpurun call: every workload re-profiled at the final code (kernel stats, PMC, bench lines), the
# config-4 account (the cascade's forms, the lab with and without the HBM streams, their counters), the sustained run of the
# headline, strict-SVF cost, the 8-rank launch paths, instruction micro-benchmarks. Output under gpurun_out/; copy what is to
# be judged into profiles/ (tools/collect_round3.sh).
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f; rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r['kernel'], r['kernel_ms'], r['frac'], (r.get('clock') or {}).get('ghz_live'))"; }

tools/gpu_profile_all.sh r03 cfg3 cfg4 cfg5 cfg5full synth synthfused events resample > $O/prof_main.log 2>&1
mv gpurun_out/profiles_r03 $O/profiles_main
EXTRA="--voices 4194304" tools/gpu_profile_all.sh r03 cfg2 > $O/prof_cfg2.log 2>&1
mv gpurun_out/profiles_r03 $O/profiles_cfg2_1GiB
tools/gpu_profile_all.sh r03 cfg2 > $O/prof_cfg2s.log 2>&1
mv gpurun_out/profiles_r03 $O/profiles_cfg2_32MiB
MLGPU_DELAY_WINDOWS=1 tools/gpu_profile_all.sh r03 strings > $O/prof_strings.log 2>&1
mv gpurun_out/profiles_r03 $O/profiles_strings_windows
EXTRA="--voices 262144" tools/gpu_profile_all.sh r03 cfg4 > $O/prof_cfg4big.log 2>&1
mv gpurun_out/profiles_r03 $O/profiles_cfg4_262144
cp profiles/pmc_workloads.json $O/pmc_workloads.json

# config 4: the forms of the cascade on this box; the lab with the real streams and with the rows collapsed (no HBM), and the
# counters of both (GRBM_GUI_ACTIVE = cycles: the same with and without the streams)
{ echo "# config 4 (131072 channels x 32 DSPVectors): bench.py --cascade-lanes, one box";
  for l in -1 1 2 4 0; do echo "## lanes $l"; $B --workload cfg4 --cascade-lanes $l 2>/dev/null | tail -1 | line; done
  echo "# the same at other bank sizes (lanes 0 = what the launcher picks)";
  for V in 4096 16384 32768 65536 262144; do for l in -1 0; do echo "## $V channels, lanes $l"; $B --workload cfg4 --voices $V --cascade-lanes $l 2>/dev/null | tail -1 | line; done; done
} > $O/cfg4_forms.txt 2>&1
X=$PWD/tools/bin/exp_cascade2
{ for m in 0 1 2 3; do timeout 300 $X 131072 32 7 $m; done; for V in 4096 16384 32768 49152 65536 262144; do timeout 300 $X $V 32 5; done; } > $O/cascade_lab.txt 2>&1
tools/bin/lab_pmc.sh $O/cascade_lab_pmc_mode0.txt $X 131072 32 1 0
tools/bin/lab_pmc.sh $O/cascade_lab_pmc_mode1.txt $X 131072 32 1 1
tools/bin/bankbench > $O/bankbench.txt 2>&1
tools/bin/instbench > $O/instbench.txt 2>&1

# the headline, sustained
$B --workload cfg3 --steps 3000 2>/dev/null | tail -1 > $O/cfg3_steps3000_bench.json
$B --workload cfg3 --sustained $O/cfg3_sustained.json --sustained-seconds 30 2>/dev/null | tail -1 > $O/cfg3_sustained_line.json

# strict SVF
{ echo "# mlgpu_engine_set_strict_svf against the default arithmetic, same box";
  for w in cfg3 cfg4 cfg5; do for s in "" "--strict-svf"; do echo "## $w $s"; $B --workload $w $s 2>/dev/null | tail -1 | line; done; done
} > $O/strict_svf.txt 2>&1

# hiprtc with the ahead-of-time build's scheduling strategy
{ echo "# hiprtc kernels with and without -mllvm -amdgpu-sched-strategy=max-ilp (MLGPU_JIT_EXTRA_OPTS), same box";
  for w in cfg5 cfg5full synth synthfused; do for o in "" "-mllvm -amdgpu-sched-strategy=max-ilp"; do
    export MLGPU_CACHE_DIR=$(mktemp -d /tmp/mlgpu_cache_XXXXXX); echo "## $w [$o]"; MLGPU_JIT_EXTRA_OPTS="$o" $B --workload $w 2>/dev/null | tail -1 | line; rm -rf $MLGPU_CACHE_DIR; unset MLGPU_CACHE_DIR; done; done
} > $O/jit_maxilp.txt 2>&1

# synthfused: where the voice sum is made
{ echo "# instrument bank: the voice sum as a kernel of its own or inside the voice kernel (MLGPU_BENCH_MIXDOWN)";
  for w in synth synthfused; do for mix in kernel graph; do echo "## $w mixdown=$mix"; MLGPU_BENCH_MIXDOWN=$mix $B --workload $w 2>/dev/null | tail -1 | line; done; done
} > $O/synth_mixdown.txt 2>&1

# instrument bank: EventsToSignals on a stream of its own (fences), and the voice sum alone
{ echo "# bench.py --workload synth with and without --two-streams; --workload mixgroups; same box";
  for i in 1 2; do for m in "" "--two-streams"; do echo "## synth $m"; $B --workload synth $m --steps 30 --warmup 5 2>/dev/null | tail -1 | line; done; done
  echo "## mixgroups"; $B --workload mixgroups 2>/dev/null | tail -1 | line
} > $O/two_streams_lines.txt 2>&1

# graph kernels: the oscillators' polyBLEP per sample (0) or once per zone per trip of 1 / 2 / 4 quads (MLGPU_GRAPH_OSC_TRIP; default 2)
{ echo "# MLGPU_GRAPH_OSC_TRIP: SawGen / PulseGen nodes with per-voice frequency, polyBLEP per sample (0) or per trip of 1, 2, 4 quads; same box";
  echo "# columns: units/s, kernel, ms per launch, fraction of 8 TB/s, clock";
  for i in 1 2; do for t in 0 1 2 4; do echo "## cfg5 trip=$t"; MLGPU_GRAPH_OSC_TRIP=$t $B --workload cfg5 2>/dev/null | tail -1 | line; done; done
  for t in 0 2; do echo "## cfg5full trip=$t"; MLGPU_GRAPH_OSC_TRIP=$t $B --workload cfg5full 2>/dev/null | tail -1 | line; done
} > $O/osc_trips.txt 2>&1

tools/multi_gpu_dry_run.sh $O/multi_gpu_launch_paths.txt 8
python tools/node_costs.py 2 10 > $O/node_costs.txt 2>&1
for dd in $O/profiles_*; do python tools/summarize_profiles.py $dd r03 > $dd/summary.md 2>/dev/null; done
cat $O/profiles_*/summary.md | grep -v "^|---\|^| bench file" 
tail -3 $O/cfg4_forms.txt
