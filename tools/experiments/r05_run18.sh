cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_events.py -x -q -m gpu --durations=5 2>&1 | tail -12
