cd $GRAFT_REPO_ROOT
free -g | head -2; cat /sys/fs/cgroup/memory.max 2>/dev/null; nproc
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_graph.py -x -q -m gpu -k "every_voice or every_channel or baseline_length or bench_step" --durations=8 2>&1 | tail -16
