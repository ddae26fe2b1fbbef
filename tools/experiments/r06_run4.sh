#!/bin/bash
# round 6, run 4: counters of allpass4 (8 rings, per-voice delay times) in ring layouts 1 and 4 at 131072 voices
export TMPDIR=/tmp
mkdir -p gpurun_out/r06d
for l in 4 1; do
  MLGPU_DELAY_WINDOWS=$l EXTRA="--voices 131072" timeout 1200 bash tools/gpu_profile_all.sh r06d_l$l allpass4 > gpurun_out/r06d/profile_l$l.log 2>&1
  tail -4 gpurun_out/r06d/profile_l$l.log
  cp gpurun_out/profiles_r06d_l$l/*allpass4* gpurun_out/r06d/ 2>/dev/null
done
