python -m pytest tests/test_gpu_delays.py -m gpu -x -q 2>&1 | tail -3
MLGPU_SOAK_LAYOUT=2 python tools/ring_layout_soak.py 250 21 2>&1 | tail -2
MLGPU_SOAK_LAYOUT=4 python tools/ring_layout_soak.py 250 22 2>&1 | tail -2
MLGPU_SOAK_LAYOUT=1 python tools/ring_layout_soak.py 150 23 2>&1 | tail -2
