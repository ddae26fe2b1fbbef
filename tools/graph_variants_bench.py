"""Developer tool: time the synth16 graph kernel for voices-per-lane x unroll x (constant | streamed pitch) on the GPU.
    python tools/graph_variants_bench.py            (prints ms per launch of 16 DSPVectors, 262144 voices)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import madronalib_amd as ml  # noqa: E402
from madronalib_amd import patches  # noqa: E402
from madronalib_amd.sharding import cfg5_gate_quad, cfg5_voice_params  # noqa: E402

V, T = 262144, 16
eng = ml.Engine(0)
n = V * T * 64
d_gate = eng.to_device(cfg5_gate_quad(0, V, T))
d_pitch = eng.to_device(np.full(n, 0.25, np.float32))
d_out = eng.alloc(4 * n)
params, coeffs, seeds = cfg5_voice_params(0, V, V, ml)
for pitch_in in (False, True):
    for vpl in (1, 2):
        for u in ("1", "2"):
            os.environ["MLGPU_GRAPH_UNROLL"] = u
            desc, outs = patches.synth16(pitch_input=pitch_in)
            g = ml.Graph(eng, V, desc, outs, voices_per_lane=vpl)
            g.clear()
            for k, v in params.items():
                if not (pitch_in and k == "pitch"):
                    g.set_param(k, v if np.ndim(v) else float(v))
            for k, c in coeffs.items():
                g.set_coeffs(k, [np.ascontiguousarray(r) for r in c])
            names = [d["name"] for d in desc if d["type"] == "input"]
            ins = [d_gate if nm == "gate" else d_pitch for nm in names]
            for _ in range(5):
                g.process(T, ins, [d_out])
            eng.sync()
            eng.timer_start()
            for _ in range(20):
                g.process(T, ins, [d_out])
            ms = eng.timer_stop_ms() / 20
            print(f"pitch_in={pitch_in} vpl={vpl} unroll={u}: {ms:.3f} ms  ({V * T * 64 / ms / 1e6:.1f} G voice-samples/s)", flush=True)
            g.close()
