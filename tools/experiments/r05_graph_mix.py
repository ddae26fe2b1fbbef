"""Config 5's voice graph (262 144 voices x 16 DSPVectors per launch) mixed to one channel: mlgpu_graph_process + mlgpu_mixdown against
the graph with mlgpu_graph_set_output_mixdown (the first stage inside the voice kernel).   python tools/experiments/r05_graph_mix.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import madronalib_amd as ml
from madronalib_amd import patches
from madronalib_amd.constants import Layout
from madronalib_amd.sharding import cfg5_gate_quad, cfg5_voice_params

V, T, reps = 262144, 16, 20
eng = ml.Engine(0)
eng.mixdown_reserve(V, T)
desc, outs = patches.synth16()
params, coeffs, seeds = cfg5_voice_params(0, V, V, ml)
d_gate = eng.to_device(cfg5_gate_quad(0, V, T))
d_voices, d_mix = eng.alloc(4 * V * T * 64), eng.alloc(4 * T * 64)
res = {}
for mixed in (False, True):
    g = ml.Graph(eng, V, desc, outs, compile_now=False)
    if mixed:
        g.set_output_mixdown(0)
    g.compile()
    g.clear()
    for k, v in params.items():
        g.set_param(k, v if np.ndim(v) else float(v))
    for k, c in coeffs.items():
        g.set_coeffs(k, [np.ascontiguousarray(r) for r in c])
    g.set_state("noise", 0, seeds)

    def step():
        if mixed:
            g.process(T, [d_gate], [d_mix])
        else:
            g.process(T, [d_gate], [d_voices])
            eng.mixdown(d_voices, Layout.QUAD, V, T, d_mix)
    for _ in range(3):
        step()
    eng.sync()
    eng.timer_start()
    for _ in range(reps):
        step()
    res[mixed] = eng.timer_stop_ms() / reps
    g.close()
print(f"config 5 voices -> one channel, 262144 voices x 16 DSPVectors: graph_process + mixdown {res[False]:.3f} ms per launch; "
      f"output mixed down in the voice kernel {res[True]:.3f} ms ({V * T * 64 / (res[True] * 1e-3):.3e} voice-samples/s)")
