"""Which samples of the strings workload does ring layout 2 fetch outside its windows? (run with and without
MLGPU_JIT_EXTRA_OPTS=-DMLGPU_RING_X_MISSPOISON, MLGPU_CACHE_DIR=off; compares against layout 1)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import madronalib_amd as ml
from madronalib_amd.constants import Proc, Op, Layout
V, T = 4096, 16
eng = ml.Engine(0)
def run(windows):
    desc = [dict(name="x", type="input"), dict(name="g", type="const", value=0.995),
            dict(name="fb", type="feedback", source="damp"),
            dict(name="fbg", type="op", kind=Op.MULTIPLY, inputs=["fb", "g"]),
            dict(name="sum", type="op", kind=Op.ADD, inputs=["x", "fbg"]),
            dict(name="line", type="proc", kind=Proc.FRACTIONAL_DELAY, inputs=["sum"], max_delay=1024.0),
            dict(name="damp", type="proc", kind=Proc.ONE_POLE, inputs=["line"])]
    g = ml.Graph(eng, V, desc, ["damp"], delay_windows=windows)
    g.set_coeffs("damp", ml.OnePole.makeCoeffs(0.3))
    length = 48000.0 / (46.0 * 2.0 ** (4.0 * ((np.arange(V) * 7919) % V) / V)) - 64.0
    st = {float(d): ml.FractionalDelay.makeState(float(d)) for d in np.unique(np.round(length, 2))}
    words = np.stack([st[float(d)] for d in np.round(length, 2)], 1).astype(np.float32)
    g.set_state("line", 3, words[0].view(np.uint32))
    g.set_state("line", 4, words[1].view(np.uint32))
    nb = eng.bank([Proc.NOISE_GEN], V)
    nb.set_state(0, 0, np.arange(1, 1 + V, dtype=np.uint32))
    n = V * T * 64
    d_x = eng.alloc(4 * n)
    d_y = eng.alloc(4 * n)
    outs = []
    for k in range(4):
        nb.process(T, d_x, Layout.QUAD)
        g.process(T, [d_x], [d_y])
        outs.append(d_y.download(np.float32, n).copy())
    return np.stack(outs)
a = run(1)
b = run(2)
bad = a.view(np.uint32) != b.view(np.uint32)
print("layout 2 vs layout 1: differing samples", int(bad.sum()), "of", bad.size, " per launch:", [int(x.sum()) for x in bad])
if bad.any():
    # QUAD layout [T*16][V][4]: sample index s = (q*4+k), voice v
    bb = bad[0].reshape(T * 16, V, 4).transpose(1, 0, 2).reshape(V, T * 64)
    first = np.where(bb.any(1), bb.argmax(1), -1)
    length = 48000.0 / (46.0 * 2.0 ** (4.0 * ((np.arange(V) * 7919) % V) / V)) - 64.0
    dint = np.floor(length).astype(int)
    hit = first >= 0
    print("voices with a poisoned sample in launch 0:", int(hit.sum()), "of", V)
    for v in np.nonzero(hit)[0][:24]:
        print(f"  voice {v:5d} lane {v % 64:2d} delay {length[v]:8.2f}  first poisoned output sample {first[v]}")
    print("delays of clean voices: min %.1f max %.1f; of poisoned: min %.1f max %.1f" % (length[~hit].min() if (~hit).any() else -1, length[~hit].max() if (~hit).any() else -1, length[hit].min(), length[hit].max()))
    fh = np.bincount(first[hit] % 16, minlength=16)
    print("first poisoned sample mod 16:", fh.tolist())
    waves = np.unique(np.nonzero(hit)[0] // 64)
    print("wavefronts with a poisoned lane:", len(waves), "of", V // 64)
