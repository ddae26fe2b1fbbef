"""Developer tool: mlgpu_mixdown (all voices -> one channel) and mlgpu_mixdown_groups read rates."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import madronalib_amd as ml  # noqa: E402
from madronalib_amd.constants import Layout  # noqa: E402

V, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (262144, 16)
eng = ml.Engine(0)
n = V * T * 64
d_x = eng.alloc(4 * n)
d_x.upload(np.random.default_rng(0).standard_normal(n).astype(np.float32))
d_o = eng.alloc(4 * T * 64)
d_g = eng.alloc(4 * (V // 16) * T * 64)
eng.mixdown_reserve(V, T)
for name, fn in (("mixdown", lambda: eng.mixdown(d_x, Layout.QUAD, V, T, d_o)), ("mixdown_groups P=16", lambda: eng.mixdown_groups(d_x, Layout.QUAD, V // 16, 16, T, d_g))):
    for _ in range(3):
        fn()
    eng.sync()
    eng.timer_start()
    for _ in range(30):
        fn()
    ms = eng.timer_stop_ms() / 30
    print(f"{name}: {ms:.3f} ms, {4 * n / ms / 1e9:.2f} TB/s read")
