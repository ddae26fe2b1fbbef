#!/bin/bash
# round 6, run 8: the whole GPU suite; the ring-layout record (profiles/r06_ring_layouts.txt)
export TMPDIR=/tmp
mkdir -p gpurun_out/r06h
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r06h/gpu_tests.txt 2>&1; tail -4 gpurun_out/r06h/gpu_tests.txt
R=gpurun_out/r06h/ring_layouts.txt
: > $R
for lay in 2 4; do MLGPU_SOAK_LAYOUT=$lay timeout 900 python tools/ring_layout_soak.py 250 11 2>&1 | tail -1 >> $R; done
python tools/experiments/r05_rings_multi.py 2>&1 | grep "rings per voice" >> $R
for v in 65536 131072 262144; do for l in 1 4; do
  MLGPU_DELAY_WINDOWS=$l timeout 300 python bench.py --workload allpass4 --voices $v --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('allpass4 (4 x Allpass<PitchbendableDelay>, per-voice delay times 400..3400) V=$v layout=$l:', round(r['kernel_ms'],3), 'ms', round(r['frac'],3), 'of HBM (algorithmic 104 B per voice-sample)')"
done; done >> $R 2>&1
for l in 1 2 4 3; do MLGPU_DELAY_WINDOWS=$l timeout 300 python bench.py --workload strings --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('strings layout=$l:', round(r['kernel_ms'],3), 'ms', round(r['frac'],3), 'of HBM')"; done >> $R 2>&1
cat $R
MLGPU_DELAY_WINDOWS=4 EXTRA="--voices 131072" timeout 1200 bash tools/gpu_profile_all.sh r06h allpass4 > gpurun_out/r06h/profile.log 2>&1
cp gpurun_out/profiles_r06h/*allpass4* gpurun_out/r06h/
grep -A8 "== mlgpu_graph_kernel" gpurun_out/r06h/r06h_allpass4_pmc.txt | grep -E "FETCH_SIZE|WRITE_SIZE|SQ_WAIT_ANY |SQ_WAVE_CYCLES|SQ_INSTS_VALU "
