"""Ring layouts 0 / 1 / 2 on graphs with more than one ring per voice (layout 2 costs 40 KiB of LDS per ring and workgroup: 2 rings ->
two workgroups per CU, 3-4 rings -> one): is layout 3's choice (2 wherever it fits) right there too?
    python tools/experiments/r05_rings_multi.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import madronalib_amd as ml
from madronalib_amd.constants import Proc, Op, Layout

V, T, reps = 262144, 16, 12
eng = ml.Engine(0)
n = V * T * 64
nb = eng.bank([Proc.NOISE_GEN], V)
nb.set_state(0, 0, np.arange(1, 1 + V, dtype=np.uint32))
d_x = eng.alloc(4 * n)
nb.process(T, d_x, Layout.QUAD)
d_y = eng.alloc(4 * n)
length = (48000.0 / (46.0 * 2.0 ** (4.0 * ((np.arange(V) * 7919) % V) / V)) - 64.0).astype(np.float32)


def case(name, build, rings):
    res = []
    for layout in (1, 2, 4):
        try:
            g = build(layout)
        except ml.MlgpuError as ex:
            res.append(f"layout {layout}: {ex.status.name if hasattr(ex.status, 'name') else ex.status}")
            continue
        for _ in range(3):
            g.process(T, [d_x], [d_y])
        eng.sync()
        eng.timer_start()
        for _ in range(reps):
            g.process(T, [d_x], [d_y])
        ms = eng.timer_stop_ms() / reps
        alg = (8.0 + 8.0 * rings) * n
        res.append(f"layout {layout}: {ms:.3f} ms, {alg / (ms * 1e-3) / 8e12:.3f} of HBM, {g.workgroups_per_cu()} workgroups per CU")
        g.close()
    print(f"{name} ({rings} rings per voice; algorithmic {8 + 8 * rings} B per voice-sample):  " + "  |  ".join(res), flush=True)


def pitchbend(layout):
    g = ml.Graph(eng, V, delay_windows=layout)
    g.add("x", "input")
    g.add("dt", "param")
    g.add("d", "proc", Proc.PITCHBENDABLE_DELAY, ["x", "dt"], max_delay=1024.0)
    g.add_output("d")
    g.compile()
    g.set_param("dt", np.maximum(length, 1.0))
    return g


def series(k):
    def build(layout):
        g = ml.Graph(eng, V, delay_windows=layout)
        g.add("x", "input")
        src = "x"
        for j in range(k):
            g.add(f"d{j}", "proc", Proc.FRACTIONAL_DELAY, [src], max_delay=1024.0)
            src = f"d{j}"
        g.add_output(src)
        g.compile()
        for j in range(k):
            ln = np.maximum(np.roll(length, 1000 * j), 1.0)
            st = {float(d): ml.FractionalDelay.makeState(float(d)) for d in np.unique(np.round(ln, 2))}
            words = np.stack([st[float(d)] for d in np.round(ln, 2)], 1).astype(np.float32)
            g.set_state(f"d{j}", 3, words[0].view(np.uint32))
            g.set_state(f"d{j}", 4, words[1].view(np.uint32))
        return g
    return build


case("one PitchbendableDelay, per-voice delay times", pitchbend, 2)
for k in (1, 2, 3, 4):
    case(f"{k} FractionalDelay in series, per-voice delay times", series(k), k)
