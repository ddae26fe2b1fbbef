#!/bin/bash
# round 6, run 5: layout 4 with one wavefront per SIMD (no scratch: the overflow lives in AGPRs) against two (spilling)
export TMPDIR=/tmp
mkdir -p gpurun_out/r06e
for w in 1 2; do for v in 65536 131072; do
  MLGPU_SECTOR_WAVES=$w MLGPU_DELAY_WINDOWS=4 timeout 300 python bench.py --workload allpass4 --voices $v --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('allpass4 V=$v layout=4 waves=$w', 'ms', round(r['kernel_ms'],3), 'frac', round(r['frac'],3))"
done; done 2>&1 | tee gpurun_out/r06e/allpass4.txt
MLGPU_SECTOR_WAVES=1 MLGPU_DELAY_WINDOWS=4 EXTRA="--voices 131072" PMC=1 timeout 1200 bash tools/gpu_profile_all.sh r06e allpass4 > gpurun_out/r06e/profile.log 2>&1
grep -A6 "== mlgpu_graph_kernel" gpurun_out/profiles_r06e/r06e_allpass4_pmc.txt | grep -E "FETCH_SIZE|WRITE_SIZE|SQ_WAIT_ANY |SQ_WAVE_CYCLES|SQ_INSTS_VALU "
cp gpurun_out/profiles_r06e/*allpass4* gpurun_out/r06e/
