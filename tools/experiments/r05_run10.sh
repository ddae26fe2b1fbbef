cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_graph.py -x -q -m gpu -k patch_swap -s 2>&1 | tail -30
