#!/usr/bin/env python3
"""What one node of a fused graph costs on the device: for every processor / op kind, a graph with N copies in series (filters,
ops) or summed (generators) behind a fixed chain of PRE Lopass filters, against the same graph with the PRE filters only;
262 144 voices x 16 DSPVectors, HIP-event time per launch. The PRE filters make the base instruction-issue bound (without them
the first ~90 ns of arithmetic per wavefront-sample hide under the HBM time of the input and output streams), and N is kept
small so that the kernel stays under 128 VGPRs (4 wavefronts per SIMD, like the real patches). The difference per copy is
issue time per wavefront-sample: ns per node-sample per SIMD.
    python tools/node_costs.py [N [PRE]]          (on the GPU box)"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import madronalib_amd as ml  # noqa: E402
from madronalib_amd.constants import Layout, Op, Proc  # noqa: E402

V, T = 262144, 16
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
PRE = int(sys.argv[2]) if len(sys.argv) > 2 else 10


def series(kind, is_proc, extra=()):
    d = [dict(name="x", type="input"), dict(name="f", type="param"), dict(name="c", type="const", value=0.37), dict(name="lo", type="const", value=-1.0),
         dict(name="hi", type="const", value=1.0)]
    src = "x"
    for i in range(PRE):
        d.append(dict(name=f"pre{i}", type="proc", kind=Proc.LOPASS, inputs=[src]))
        src = f"pre{i}"
    for i in range(N):
        if is_proc:
            d.append(dict(name=f"n{i}", type="proc", kind=kind, inputs=[src] + list(extra)))
        else:
            ins = {1: [src], 2: [src, "c"], 3: [src, "lo", "hi"]}[1 if kind < 32 else (2 if kind < 64 else 3)]
            d.append(dict(name=f"n{i}", type="op", kind=kind, inputs=ins))
        src = f"n{i}"
    if N == 0 and PRE == 0:
        d.append(dict(name="n0", type="op", kind=Op.ADD, inputs=["x", "c"]))
        src = "n0"
    return d, [src]


def summed(kind, inputs):
    d = [dict(name="x", type="input"), dict(name="f", type="param"), dict(name="w", type="param")]
    acc = "x"
    for i in range(PRE):
        d.append(dict(name=f"pre{i}", type="proc", kind=Proc.LOPASS, inputs=[acc]))
        acc = f"pre{i}"
    for i in range(N):
        d.append(dict(name=f"g{i}", type="proc", kind=kind, inputs=inputs))
        d.append(dict(name=f"s{i}", type="op", kind=Op.ADD, inputs=[acc, f"g{i}"]))
        acc = f"s{i}"
    return d, [acc]


def time_graph(eng, desc, outs, d_x, d_y):
    g = ml.Graph(eng, V, desc, outs)
    names = {n["name"] for n in desc}
    if "f" in names:
        g.set_param("f", (55.0 * 2.0 ** (5.0 * np.arange(V) / V) / 48000.0).astype(np.float32))
    if "w" in names:
        g.set_param("w", np.full(V, 0.3, np.float32))
    for n in desc:
        if n["type"] == "proc" and n["kind"] in (Proc.LOPASS, Proc.BANDPASS):
            g.set_coeffs(n["name"], ml.Lopass.makeCoeffs(0.1, 0.7))
        if n["type"] == "proc" and n["kind"] == Proc.HIPASS:
            g.set_coeffs(n["name"], ml.Hipass.makeCoeffs(0.01, 0.7))
        if n["type"] == "proc" and n["kind"] == Proc.ONE_POLE:
            g.set_coeffs(n["name"], ml.OnePole.makeCoeffs(0.2))
        if n["type"] == "proc" and n["kind"] == Proc.DC_BLOCKER:
            g.set_coeffs(n["name"], [ml.DCBlocker.makeCoeffs(0.05)])
        if n["type"] == "proc" and n["kind"] == Proc.ADSR:
            g.set_coeffs(n["name"], ml.ADSR.calcCoeffs(0.005, 0.01, 0.6, 0.02, 48000.0))
    for _ in range(6):
        g.process(T, [d_x], [d_y])
    eng.timer_start()
    reps = 30
    for _ in range(reps):
        g.process(T, [d_x], [d_y])
    ms = eng.timer_stop_ms() / reps
    g.close()
    return ms


def main():
    eng = ml.Engine(0)
    n = V * T * 64
    gate = np.zeros((16 * T, V, 4), np.float32)
    gate[8:40] = 0.8                      # a gate that opens and closes inside the launch; otherwise a plain signal
    d_x = eng.to_device(gate + np.float32(0.001) * np.random.default_rng(1).standard_normal((16 * T, V, 4)).astype(np.float32))
    d_y = eng.alloc(4 * n)
    wave_samples_per_simd = V * T * 64 / 64 / 1024
    global N
    keep = N
    N = 0
    base = time_graph(eng, *series(Op.ADD, False), d_x, d_y)
    N = keep
    rows = []
    cases = [("adsr", series(Proc.ADSR, True)), ("lopass", series(Proc.LOPASS, True)), ("hipass", series(Proc.HIPASS, True)),
             ("bandpass", series(Proc.BANDPASS, True)), ("one_pole", series(Proc.ONE_POLE, True)), ("dc_blocker", series(Proc.DC_BLOCKER, True)),
             ("multiply (op)", series(Op.MULTIPLY, False)), ("clamp const bounds (op)", series(Op.CLAMP, False)), ("exp2_approx (op)", series(Op.EXP2_APPROX, False)),
             ("sin (op)", series(Op.SIN, False)), ("saw_gen, per-voice freq", summed(Proc.SAW_GEN, ["f"])), ("pulse_gen, per-voice freq", summed(Proc.PULSE_GEN, ["f", "w"])),
             ("sine_gen", summed(Proc.SINE_GEN, ["f"])), ("noise_gen", summed(Proc.NOISE_GEN, [])), ("saw_gen, streamed freq", summed(Proc.SAW_GEN, ["x"]))]
    for name, (desc, outs) in cases:
        ms = time_graph(eng, desc, outs, d_x, d_y)
        per = (ms - base) * 1e6 / wave_samples_per_simd / N
        extra_add = 1.09 if name.startswith(("saw", "pulse", "sine_gen", "noise")) else 0.0     # the accumulating add of the summed form
        rows.append(dict(node=name, ms=ms, ns_per_node_sample_per_simd=per - extra_add))
        print(f"{name:28s} {ms:7.3f} ms   {per - extra_add:6.1f} ns per node-sample per SIMD", flush=True)
    print(json.dumps(dict(tool="node_costs", copies=N, pre_lopass=PRE, voices=V, vectors=T, base_ms=base, rows=rows)))


if __name__ == "__main__":
    main()
