#!/usr/bin/env python3
"""Developer tool (GPU): A/B of generator knobs on BASELINE's graph workloads. Every variant starts from the same cleared state, runs
the same launches and must produce the same bits (CRC of the last output and of every state word); prints ms per launch.
    python tools/graph_ab.py "cfg5,cfg5full,synthpitch" "MLGPU_GRAPH_PREFETCH=0" "MLGPU_GRAPH_PREFETCH=1" ...
A variant is a comma-separated list of NAME=VALUE pairs ("-" = no knob)."""
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import madronalib_amd as ml  # noqa: E402
from madronalib_amd import patches  # noqa: E402
from madronalib_amd.sharding import cfg5_gate_quad, cfg5_voice_params  # noqa: E402

V, T = int(os.environ.get("AB_V", 262144)), 16
REPS = int(os.environ.get("AB_REPS", 40))
workloads = sys.argv[1].split(",")
variants = sys.argv[2:] or ["-"]
eng = ml.Engine(0)
n = V * T * 64
GATE_SCALE = int(os.environ.get("AB_GATE_SCALE", 1))   # > 1: notes that many times longer than BASELINE's 6-48 ms (the bench's gate is a stress pattern)


def gate_quad(n_vectors, scale):
    if scale == 1:
        return cfg5_gate_quad(0, V, n_vectors)
    v = np.arange(V, dtype=np.int64)[None, :, None]
    s = (np.arange(16 * n_vectors, dtype=np.int64)[:, None, None] * 4 + np.arange(4, dtype=np.int64)[None, None, :])
    half = (300 + (v * 131) % 2000) * scale
    on = (((s + (v * 977) % 4096) // half) & 1) == 0
    amp = (0.2 + 0.8 * ((v * 41) % 64) / 63.0).astype(np.float32)
    return np.ascontiguousarray(np.where(on, amp, np.float32(0.0)).astype(np.float32))


d_gate = eng.to_device(gate_quad(T, GATE_SCALE))
d_pitch = eng.to_device((0.25 + 0.5 * np.sin(np.arange(n, dtype=np.float64) * 1e-3)).astype(np.float32))
d_out = eng.alloc(4 * n)
knobs = set()
for v in variants:
    for kv in v.split(","):
        if "=" in kv:
            knobs.add(kv.split("=")[0])
for w in workloads:
    full = w == "cfg5full"
    pitch_in = w == "synthpitch"
    ref = None
    for rnd in range(2):   # two rounds over the variants: drift of the box shows as a difference between the rounds
        for v in variants:
            for k in knobs:
                os.environ.pop(k, None)
            for kv in v.split(","):
                if "=" in kv:
                    os.environ[kv.split("=")[0]] = kv.split("=")[1]
            desc, outs = patches.synth16(full=full, pitch_input=pitch_in)
            g = ml.Graph(eng, V, desc, outs, voices_per_lane=1)
            g.clear()
            params, coeffs, seeds = cfg5_voice_params(0, V, V, ml, full=full)
            for k, x in params.items():
                if not (pitch_in and k == "pitch"):
                    g.set_param(k, x if np.ndim(x) else float(x))
            for k, c in coeffs.items():
                g.set_coeffs(k, [np.ascontiguousarray(r) for r in c])
            g.set_state("noise", 0, seeds)
            names = [d["name"] for d in desc if d["type"] == "input"]
            ins = [d_gate if nm == "gate" else d_pitch for nm in names]
            for _ in range(4):
                g.process(T, ins, [d_out])
            eng.sync()
            eng.timer_start()
            for _ in range(REPS):
                g.process(T, ins, [d_out])
            ms = eng.timer_stop_ms() / REPS
            crc = zlib.crc32(d_out.download(np.uint32, n).tobytes())
            for nd in desc:
                if nd["type"] == "proc":
                    for i in range(g.num_state(nd["name"])):
                        crc = zlib.crc32(np.ascontiguousarray(g.get_state(nd["name"], i)).tobytes(), crc)
            if ref is None:
                ref = crc
            print(f"{w:11s} round {rnd} {v:40s} {ms:8.4f} ms  {V * T * 64 / ms / 1e9:7.2f} G voice-samples/s  crc {crc:08x} {'same bits' if crc == ref else 'DIFFERENT BITS'}", flush=True)
            g.close()
