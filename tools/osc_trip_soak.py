#!/usr/bin/env python3
"""Soak of the graph kernels' oscillator trips (mldsp_procs.hpp: trip_u): SawGen / PulseGen nodes with per-voice frequency and width,
free-running random phases, the trip form (MLGPU_GRAPH_OSC_TRIP = 1 / 2 / 4) against the per-sample form (0) of the same graph on the
same inputs, bit for bit, outputs and final phases. The per-sample form is the one the oracle tests pin; this run adds volume: natural
knife-edge trips (a phase within 2^-22 of 0 or 1) occur about once per 10^6 samples.   usage: tools/osc_trip_soak.py [seeds] [voices] [vectors]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import madronalib_amd as ml  # noqa: E402
from madronalib_amd.constants import Layout, Proc  # noqa: E402


def build(eng, V, trip):
    os.environ["MLGPU_GRAPH_OSC_TRIP"] = str(trip)
    desc = [dict(name="f", type="param"), dict(name="w", type="param"),
            dict(name="saw", type="proc", kind=Proc.SAW_GEN, inputs=["f"]),
            dict(name="pw", type="proc", kind=Proc.PULSE_GEN, inputs=["f", "w"])]
    g = ml.Graph(eng, V, desc, ["saw", "pw"])
    assert (".trip_u<" in g.source) == (trip != 0)
    return g


def main():
    # (both oscillators sit on one frequency node and start from the same counters: since round 4 that is the LOCKED pair, one trip
    # for both, mldsp_procs.hpp: trip_locked. `unlock` as a last argument gives the PulseGen counters of its own: two trips.)
    unlock = sys.argv[-1] == "unlock"
    if unlock:
        sys.argv.pop()
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    V = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    eng = ml.Engine(0)
    forms = {t: build(eng, V, t) for t in (0, 1, 2, 4)}
    total, bad = 0, 0
    for seed in range(seeds):
        rng = np.random.default_rng(1000 + seed)
        lim = [1.0 / 32.0, 1.0 / 16.0, 1.0 / 8.0][seed % 3]             # inside every form's range / the default's / the shortest trip's
        freq = (2e-5 * ((lim / 2e-5) ** rng.random(V))).astype(np.float32)
        width = rng.uniform(0.0, 1.0, V).astype(np.float32)
        width[rng.random(V) < 0.05] = rng.choice(np.array([0.0, 1.0, 0.5, 0.25], np.float32), int(V))[: int((rng.random(V) < 0.05).sum()) or 1][0]
        phases = rng.integers(0, 2 ** 32, V, dtype=np.uint64).astype(np.uint32)
        outs = {}
        for t, g in forms.items():
            g.set_param("f", freq)
            g.set_param("w", width)
            g.set_state("saw", 0, phases)
            g.set_state("pw", 0, (phases * np.uint32(2654435761) + np.uint32(12345)) if unlock else phases)
            a = g.process_host(T, {}, Layout.QUAD)
            b = g.process_host(T, {}, Layout.QUAD)   # resumed
            outs[t] = [x.view(np.uint32) for x in a + b] + [g.get_state("saw", 0), g.get_state("pw", 0)]
        for t in (1, 2, 4):
            for x, y in zip(outs[t], outs[0]):
                bad += int((x != y).sum())
        total += 2 * V * 64 * T
        print(f"seed {seed}: frequencies up to {lim:.4f}, {2 * V * 64 * T} samples per oscillator and form, mismatching words so far {bad}", flush=True)
    print(f"{'unlocked' if unlock else 'locked'} pair, {total} samples per oscillator per form, forms 1 / 2 / 4 quads per trip against the per-sample form: {bad} words differ "
          f"(a phase lands within 2^-22 of 0 or 1 about {total * 2.0 ** -21:.0f} times in that many samples)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
