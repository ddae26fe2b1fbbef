#!/usr/bin/env python3
"""What the per-instrument context signals cost at bank size: 16 384 instruments x 16 voices, 16 DSPVectors per launch.
  * mlgpu_events_process (pitch + gate rows) without and with four watched controllers, a tenth of the instruments receiving a
    controller event per launch (ctl_kernel: one lane per instrument and controller)
  * mlgpu_transport_process, every context reporting its time before every launch (transport_kernel: one lane per context)
    python tools/context_bench.py          (on the GPU box)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import madronalib_amd as ml  # noqa: E402
from madronalib_amd.constants import Layout  # noqa: E402

N, P, T = 16384, 16, 16
CTRL, NOTE_ON = 6, 1


def run(eng, watch):
    ev = ml.Events(eng, N, P)
    ev.set_wanted_rows([0, 1])
    if watch:
        ev.watch_controllers(watch, T)
    rows = [eng.alloc(4 * N * P * T * 64), eng.alloc(4 * N * P * T * 64)] + [None] * 6
    rng = np.random.default_rng(1)
    ev.add_events(list(range(N)), [ml.Event(NOTE_ON, 1, 60, 0, 0.0, 0.8)] * N)
    ev.process(T, 0, rows, Layout.QUAD)
    ev.clear_events()
    times = []
    for rep in range(12):
        who = rng.choice(N, N // 10, replace=False)
        ev.add_events([int(i) for i in who], [ml.Event(CTRL, 1, int(rng.choice([1, 7, 74, 16])), int(rng.integers(0, 64 * T)), float(rng.random()), 0.0) for _ in who])
        eng.sync()
        t0 = time.perf_counter()
        ev.process(T, 0, rows, Layout.QUAD)
        eng.sync()
        times.append(time.perf_counter() - t0)
        ev.clear_events()
    ev.close()
    return 1e3 * float(np.median(times[2:]))


def main():
    eng = ml.Engine(0)
    base = run(eng, None)
    with_ctl = run(eng, [1, 7, 74, 16])
    print(f"events_process, pitch + gate rows, {N} x {P} voices x {T} vectors, host routing included: {base:7.3f} ms per launch")
    print(f"  the same with 4 watched controllers ({4 * N} controller lanes, {N // 10} controller events per launch): {with_ctl:7.3f} ms  (+{with_ctl - base:.3f})")
    tr = ml.Transport(eng, N, T)
    times, ppq = [], 0.0
    for rep in range(12):
        tr.update_time(ppq, 120.0, True, 48000.0)
        eng.sync()
        t0 = time.perf_counter()
        tr.process(T)
        eng.sync()
        times.append(time.perf_counter() - t0)
        ppq += T * 64 * 120.0 / 60.0 / 48000.0
    print(f"transport_process, {N} contexts x {T} vectors, every context reporting before every launch: {1e3 * float(np.median(times[2:])):7.3f} ms per launch")
    print(f"  (the voice kernel of such a bank: ~1.5 ms per launch; a context signal is 1/{P} of a voice signal: {4 * N * T * 64 / 1e6:.1f} MB)")


if __name__ == "__main__":
    main()
