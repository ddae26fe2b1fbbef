cd $GRAFT_REPO_ROOT
export MLGPU_CACHE_DIR=off
echo "normal:"; timeout 300 python tools/experiments/r05_miss_probe.py 2>&1 | tail -3
echo "poisoned misses:"; MLGPU_JIT_EXTRA_OPTS=-DMLGPU_RING_X_MISSPOISON timeout 300 python tools/experiments/r05_miss_probe.py 2>&1 | tail -3
