set -u
cd $GRAFT_REPO_ROOT
tools/gpu_profile_all.sh r02 cfg3 cfg4 cfg5 cfg5full synth synthfused events resample > gpurun_out/prof_r02_main.log 2>&1
mv gpurun_out/profiles_r02 gpurun_out/profiles_r02_main
EXTRA="--voices 4194304" tools/gpu_profile_all.sh r02 cfg2 > gpurun_out/prof_r02_cfg2.log 2>&1
mv gpurun_out/profiles_r02 gpurun_out/profiles_r02_cfg2_1GiB
tools/gpu_profile_all.sh r02 cfg2 > gpurun_out/prof_r02_cfg2s.log 2>&1
mv gpurun_out/profiles_r02 gpurun_out/profiles_r02_cfg2_32MiB
MLGPU_DELAY_WINDOWS=1 tools/gpu_profile_all.sh r02 strings > gpurun_out/prof_r02_strings.log 2>&1
mv gpurun_out/profiles_r02 gpurun_out/profiles_r02_strings_windows
EXTRA="--voices 262144" tools/gpu_profile_all.sh r02 cfg4 > gpurun_out/prof_r02_cfg4big.log 2>&1
mv gpurun_out/profiles_r02 gpurun_out/profiles_r02_cfg4_262144
tools/bin/instbench > gpurun_out/r02_instbench.txt 2>&1
tools/bin/copybench > gpurun_out/r02_copybench.txt 2>&1
tools/bin/valubench > gpurun_out/r02_valubench.txt 2>&1
cp profiles/pmc_workloads.json gpurun_out/r02_pmc_workloads.json
cat gpurun_out/prof_r02_*.log | grep -v "^\[" | tail -60
