cd $GRAFT_REPO_ROOT
for f in 0 1 2 4; do for vt in "1048576 1" "262144 16" "4194304 1" "65536 8"; do echo "form $f  V T = $vt: $(MLGPU_MIXDOWN_STREAM=$f python tools/mixdown_bench.py $vt 2>&1 | head -1)"; done; done
python -m pytest tests/test_gpu_processbuffer.py -q -x -k mixdown 2>&1 | tail -2
