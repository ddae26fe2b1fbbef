# The round's closing call, at the frozen device code: every workload profiled with rocprofv3 kernel stats and six counter passes
# (tools/gpu_profile_all.sh), the single-class micro-kernels of tools/instbench.hip under the occupancy counters (what calibrates
# roofline.valu.busy_measured), the paced 2^20-voice target under kernel stats, the bench lines with the fresh records in place, the
# default bench line (which carries every config as flat keys) and the GPU test suite.
# Output under gpurun_out/r06f/; tools/collect_round5.sh copies it into profiles/.
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06f; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; v=r.get('valu') or {}; print(d['value'], r['kernel'], r['kernel_ms'], round(r['frac'],4), r['bound'], r.get('frac_of_ceiling'), 'busy measured', v.get('busy_measured'), 'model', v.get('busy_frac'), 'at live clock', v.get('frac_at_live_clock'))"; }
fin() { rm -rf $O/$1; mv gpurun_out/profiles_r06 $O/$1; }
mkdir -p tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/instbench.hip -o tools/bin/instbench 2> $O/instbench_build.log
tools/gpu_profile_all.sh r06 cfg3 cfg4 > $O/prof_main.log 2>&1; fin profiles_main
tools/gpu_profile_all.sh r06 cfg5 cfg5full > $O/prof_graphs.log 2>&1; fin profiles_graphs
EXTRA="--voices 4194304" tools/gpu_profile_all.sh r06 cfg2 > $O/prof_cfg2.log 2>&1; fin profiles_cfg2_1GiB
tools/gpu_profile_all.sh r06 cfg2 > $O/prof_cfg2s.log 2>&1; fin profiles_cfg2_32MiB
tools/gpu_profile_all.sh r06 synth synthrows events resample > $O/prof_wide.log 2>&1; fin profiles_wide
MLGPU_DELAY_WINDOWS=3 tools/gpu_profile_all.sh r06 strings > $O/prof_strings.log 2>&1; fin profiles_strings_best
MLGPU_DELAY_WINDOWS=3 tools/gpu_profile_all.sh r06 allpass4 > $O/prof_allpass4.log 2>&1; fin profiles_allpass4_best
tools/gpu_profile_all.sh r06 reverb > $O/prof_reverb.log 2>&1; fin profiles_reverb
cp profiles/pmc_workloads.json $O/pmc_workloads.json
python tools/check_pmc_fresh.py > $O/pmc_fresh.txt 2>&1
# the north_star target, paced, under kernel stats
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rt -- python $GRAFT_REPO_ROOT/bench.py --workload rt --steps 5 --warmup 1 > $GRAFT_REPO_ROOT/$O/rt_under_rocprof.json 2> /tmp/prof_rt.log )
f=$(find /tmp/prof_rt -name '*_kernel_stats.csv' | head -1); { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --workload rt --steps 5 --warmup 1 (r06, MI355X)"; [ -n "$f" ] && head -8 $f; } > $O/rt_kernel_stats.csv
python bench.py --workload rt 2>/dev/null | tail -1 > $O/rt_bench.json
# calibration of the measured vector-issue fraction: single-class kernels under the same counters
( cd /tmp && rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU --kernel-trace --output-format csv -d /tmp/prof_ib -- $GRAFT_REPO_ROOT/tools/bin/instbench > $GRAFT_REPO_ROOT/$O/instbench_under_pmc.txt 2> /tmp/prof_ib.log )
python - $(find /tmp/prof_ib -name '*counter_collection.csv' | head -1) > $O/valu_calibration.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append((r["Dispatch_Id"], float(r["Counter_Value"])))
print("# tools/instbench.hip (one instruction class per kernel, 4 independent chains per wave, 4 waves per SIMD, every CU) under")
print("# rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU: what bench.py's")
print("# roofline.valu.busy_measured = (ACTIVE - VALU2) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 / 4) reads on kernels whose vector unit is known to be")
print("# saturated, next to rocprofv3's own VALUBusy (= ACTIVE / 256 CUs / (GRBM_GUI_ACTIVE / 8)) and the instructions per 4-cycle slot.")
print("# kernel                 busy_measured  VALUBusy(rocprof)  instr/slot  active/insts  launches")
for k, c in sorted(acc.items()):
    def tot(n):
        d = collections.defaultdict(float)
        for disp, v in c.get(n, []):
            d[disp] += v
        return d
    A, A2, G, I = tot("SQ_ACTIVE_INST_VALU"), tot("SQ_ACTIVE_INST_VALU2"), tot("GRBM_GUI_ACTIVE"), tot("SQ_INSTS_VALU")
    rows = []
    for disp in A:
        if G.get(disp, 0) <= 0 or I.get(disp, 0) < 1e6:
            continue
        slots = 1024 * (G[disp] / 8.0) / 4.0
        rows.append(((A[disp] - A2.get(disp, 0.0)) / slots, A[disp] / 256.0 / (G[disp] / 8.0), I[disp] / slots, A[disp] / I[disp]))
    if rows:
        m = [sum(x[i] for x in rows) / len(rows) for i in range(4)]
        print(f"{k[:24]:24s} {m[0]:13.3f} {m[1]:18.3f} {m[2]:11.3f} {m[3]:13.3f} {len(rows):9d}")
PY
for w in cfg3 cfg4 cfg5 cfg5full cfg2 synth synthrows; do $B --workload $w 2>/dev/null | tail -1 > $O/${w}_line.json; echo "## $w"; cat $O/${w}_line.json | line; done > $O/lines.txt 2>&1
for w in strings allpass4; do MLGPU_DELAY_WINDOWS=3 $B --workload $w 2>/dev/null | tail -1 > $O/${w}_line.json; echo "## $w, delay layout 3 (the best form)"; cat $O/${w}_line.json | line; done >> $O/lines.txt 2>&1
for w in events resample reverb; do $B --workload $w 2>/dev/null | tail -1 > $O/${w}_line.json; echo "## $w"; cat $O/${w}_line.json | line; done >> $O/lines.txt 2>&1
# the headline kernel's dispatches in order (the record of profiles/r06_cfg3_spread.md at the closing code)
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_trace -- python $GRAFT_REPO_ROOT/bench.py --workload cfg3 --no-cpu-baseline --no-extras > /dev/null 2> /tmp/prof_trace.log )
python - $(find /tmp/prof_trace -name '*kernel_trace.csv' | head -1) $O/cfg3_dispatch_trace.csv <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'chain_kernel' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
with open(sys.argv[2], 'w') as f:
    f.write('dispatch,start_us,end_us,duration_us,gap_before_us\n')
    prev = None
    for i, r in enumerate(rows):
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        f.write(f"{i},{(s - t0) / 1e3:.2f},{(e - t0) / 1e3:.2f},{(e - s) / 1e3:.2f},{((s - prev) / 1e3) if prev else 0:.2f}\n")
        prev = e
PY
# the real-time block across GPUs: two ranks on this one GPU (the launch path and the bits, not a multi-GPU measurement)
python bench.py --workload rt --gpus 2 --oversubscribe --steps 4 --warmup 1 2> $O/rt_group.err | tail -1 > $O/rt_group_2ranks_one_gpu.json
tests/cpp/multi_engine_test > $O/multi_engine_test.txt 2>&1
# random delay graphs, every ring layout against layout 0 (tools/ring_layout_soak.py), outputs and state words bit for bit
{ for lay in 2 4 1; do MLGPU_SOAK_LAYOUT=$lay python tools/ring_layout_soak.py 300 $((60 + lay)) 2>&1 | tail -1; done; } > $O/ring_layout_soak.txt
( time python bench.py ) 2> $O/default_bench.time | tail -1 > $O/default_bench.json
python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > $O/gpu_tests.txt
for dd in $O/profiles_*; do python tools/summarize_profiles.py $dd r06 > $dd/summary.md 2>/dev/null; done
cat $O/profiles_*/summary.md | grep -v "^|---\|^| bench file"
cat $O/lines.txt $O/pmc_fresh.txt $O/gpu_tests.txt $O/valu_calibration.txt
