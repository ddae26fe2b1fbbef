#!/bin/bash
# round 6, run 1: the new parity cases, the default line with the widened legs, the reverb workload with counters
export TMPDIR=/tmp
mkdir -p gpurun_out/r06a
timeout 900 python -m pytest tests/test_gpu_widened_parity.py tests/test_gpu_events.py tests/test_graph_async.py -x -q -m gpu > gpurun_out/r06a/tests.txt 2>&1; tail -5 gpurun_out/r06a/tests.txt
timeout 900 python bench.py > gpurun_out/r06a/default_bench.json 2> gpurun_out/r06a/default_bench.err; tail -c 6000 gpurun_out/r06a/default_bench.json; tail -5 gpurun_out/r06a/default_bench.err
timeout 1500 bash tools/gpu_profile_all.sh r06a reverb > gpurun_out/r06a/profile_reverb.log 2>&1; tail -8 gpurun_out/r06a/profile_reverb.log
cp gpurun_out/profiles_r06a/* gpurun_out/r06a/ 2>/dev/null
