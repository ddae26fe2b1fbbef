#!/usr/bin/env python3
"""Downsampler / Upsampler (HalfBandFilter cascades, MLDSPFilters.h:1245-1473) on the device against the oracle: random octaves 0 .. 6, voice
counts (whole and ragged wavefronts), input lengths, 2 .. 5 launches with carried state, every layout; output and filter state bit for bit.
    python tools/resample_soak.py [cases] [seed]"""
import os
import sys
import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import madronalib_amd as ml                       # noqa: E402
from madronalib_amd.constants import Layout        # noqa: E402
from cpu_checkers import Oracle                    # noqa: E402
from inputs import lcg_noise                       # noqa: E402


def run(cases, seed, eng=None):
    rng = np.random.default_rng(seed)
    eng = eng or ml.Engine(0)
    orc = Oracle()
    bad = 0
    for case in range(cases):
        octaves = int(rng.integers(0, 7))
        up = bool(rng.integers(0, 2))
        V = int(rng.integers(1, 700)) if rng.random() < 0.6 else 64 * int(rng.integers(1, 12))
        launches = int(rng.integers(2, 6))
        layout = [Layout.QUAD, Layout.VOICE_MAJOR, Layout.ROWS][int(rng.integers(0, 3))]
        unit = 1 if up else (1 << octaves)
        Tin = unit * int(rng.integers(1, max(2, 16 // unit + 1)))
        if up and Tin * (1 << octaves) > 64:
            Tin = max(1, 64 >> octaves)
        x = lcg_noise(np.arange(V, dtype=np.uint32) + np.uint32(case * 31 + 7), 64 * Tin * launches) * np.float32(rng.choice([1.0, 1e-3, 1e3]))
        r = ml.Resampler(eng, V, octaves, up)
        st = np.zeros((octaves * 9, V), np.float32)
        diff = 0
        for call in range(launches):
            xs = np.ascontiguousarray(x[:, call * 64 * Tin:(call + 1) * 64 * Tin])
            got = r.process_host(xs, layout)
            want = orc.resample(octaves, up, st, xs)
            diff += int((np.ascontiguousarray(got).view(np.uint32) != np.ascontiguousarray(want).view(np.uint32)).sum())
            diff += int((r.get_state().view(np.uint32) != st.view(np.uint32)).sum())
        r.close()
        if diff:
            bad += 1
            print(f"case {case}: octaves {octaves} up {up} V {V} Tin {Tin} launches {launches} layout {int(layout)}: {diff} words differ")
    print(f"{cases} resamplers (seed {seed}) against the oracle: {bad} with a difference")
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
