#!/usr/bin/env python3
"""mlgpu_transport (AudioContext::ProcessTime: updateTime / processVector / getBeatPhase) against the reference's own AudioContext
(oracle/_ref/libdropin_ref.so), random host sessions (tests/test_gpu_transport.py: host_session - positions, tempi, starts and stops, loops,
relocations, rubbish reports, clears; blocks of 1 .. 8 DSPVectors) at random sample rates: the beat phase of every frame and
samplesSinceStart after every step, bit for bit.     python tools/transport_soak.py [sessions] [first seed]"""
import os
import sys
import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import madronalib_amd as ml          # noqa: E402
import test_gpu_transport as tt      # noqa: E402


def run(sessions, first, eng=None):
    eng = eng or ml.Engine(0)
    bad = 0
    for seed in range(first, first + sessions):
        rng = np.random.default_rng(90000 + seed)
        sr = float(rng.choice([8000.0, 22050.0, 44100.0, 48000.0, 88200.0, 96000.0, 192000.0]))
        script = tt.host_session(seed, blocks=int(rng.integers(10, 80)), sr=sr)
        want, since = tt.ref_run(script)
        tr = ml.Transport(eng, 2, 8)
        got, gsince = tt.gpu_run(tr, script, 1)
        tr.close()
        d = int((got[1].view(np.uint32) != want.view(np.uint32)).sum()) + int((gsince != since).sum())
        if d:
            bad += 1
            w = np.argwhere(got[1].view(np.uint32) != want.view(np.uint32))
            print(f"seed {seed} (sr {sr}): {d} words differ, first frame {w[0].tolist() if len(w) else None}")
    print(f"{sessions} host sessions (seeds {first} .. {first + sessions - 1}) against the reference's AudioContext: {bad} with a difference")
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
