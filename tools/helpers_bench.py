"""Developer tool: bandwidth of the helper kernels (layout conversion, row ops) at 262144 voices x 16 vectors."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import madronalib_amd as ml  # noqa: E402
from madronalib_amd.constants import Layout, Op, RowOp  # noqa: E402

V, T = 262144, 16
eng = ml.Engine(0)
n = V * T * 64
a, b = eng.alloc(4 * n), eng.alloc(4 * n)
a.upload(np.random.default_rng(0).standard_normal(n).astype(np.float32))
row = eng.alloc(4 * 64)
red = eng.alloc(4 * V * T)


def timed(name, fn, nbytes):
    for _ in range(3):
        fn()
    eng.sync()
    eng.timer_start()
    for _ in range(20):
        fn()
    ms = eng.timer_stop_ms() / 20
    print(f"{name:34s} {ms:7.3f} ms  {nbytes / ms / 1e9:6.2f} TB/s")


for src, dst in ((Layout.VOICE_MAJOR, Layout.QUAD), (Layout.QUAD, Layout.VOICE_MAJOR), (Layout.ROWS, Layout.QUAD), (Layout.QUAD, Layout.ROWS)):
    timed(f"layout_convert {src}->{dst}", lambda: eng.layout_convert(a, src, b, dst, V, T), 8.0 * n)
timed("op_apply_rows1 ADD1", lambda: eng.L.mlgpu_op_apply_rows1(eng.h, int(Op.ADD), a.ptr, row.ptr, b.ptr, V * T), 8.0 * n)
timed("row_reduce SUM", lambda: eng.L.mlgpu_row_reduce(eng.h, int(RowOp.SUM), a.ptr, red.ptr, V * T), 4.0 * n)
