#!/usr/bin/env python3
"""Random delay graphs in ring layout 2 against layout 0 on the GPU, bit for bit (outputs and state words, several launches with
carried state): IntegerDelay / FractionalDelay / PitchbendableDelay with delay times that are constant per voice, stepped, swept,
or jump every sample; ring sizes from 64 samples up; 1 .. 9 DSPVectors per launch; write indices 0, anywhere in a chunk, or
different in one wavefront. Layout 0 is the form the oracle tests pin (tests/test_gpu_delays.py); this checks that layout 2
follows it everywhere a seeded search reaches.     python tools/ring_layout_soak.py [cases] [seed]"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import madronalib_amd as ml                      # noqa: E402
from madronalib_amd.constants import Proc, Layout   # noqa: E402
from inputs import lcg_noise, stepped            # noqa: E402

# the layout under test against layout 0 (MLGPU_SOAK_LAYOUT=4: the sector trips of round 6)
TEST_LAYOUT = int(os.environ.get("MLGPU_SOAK_LAYOUT", "2"))
LAYOUTS = (0, TEST_LAYOUT)
ONLY = int(os.environ.get("MLGPU_SOAK_ONLY", "-1"))   # run this case alone (the random draws of the others are still made) and say where it differs


def delays(rng, kind, V, S, dmax):
    mode = int(rng.integers(0, 5))
    lo, hi = 0.0, dmax + 0.9
    if mode == 0:    # one value per voice
        d = np.repeat(rng.uniform(lo, hi, (V, 1)).astype(np.float32), S, 1)
    elif mode == 1:  # steps at random moments
        d = stepped(V, S, int(rng.integers(1 << 30)), lo, hi, 1, 200)
    elif mode == 2:  # sweeps through the short / long boundary (32 .. 48 samples) and beyond
        ph = rng.uniform(0, 6.28, (V, 1))
        rate = rng.uniform(0.001, 0.2, (V, 1))
        mid = min(40.0, dmax / 2)
        d = (mid + mid * np.sin(ph + rate * np.arange(S)[None, :])).astype(np.float32)
    elif mode == 3:  # a new delay time every sample
        d = rng.uniform(lo, hi, (V, S)).astype(np.float32)
    else:            # short ones only
        d = stepped(V, S, int(rng.integers(1 << 30)), 0.0, min(hi, 60.0), 1, 40)
    return mode, np.ascontiguousarray(d)


def run(cases, seed, eng=None):
    rng = np.random.default_rng(seed)
    eng = eng or ml.Engine(0)
    bad, nonzero, total = 0, 0, 0
    for case in range(cases):
        kind = [Proc.INTEGER_DELAY, Proc.FRACTIONAL_DELAY, Proc.PITCHBENDABLE_DELAY][int(rng.integers(0, 3))]
        V = int(rng.integers(1, 400)) if rng.random() < 0.5 else 64 * int(rng.integers(1, 7))   # whole wavefronts, and banks whose last one is not full
        T = int(rng.integers(1, 10))
        launches = int(rng.integers(2, 6))
        dmax = float([0.0, 40.0, 100.0, 192.0, 700.0, 3000.0][int(rng.integers(0, 6))])
        S = 64 * T * launches
        x = lcg_noise(np.arange(V, dtype=np.uint32) + np.uint32(case * 977 + 5), S)
        mode, d = delays(rng, kind, V, S, dmax)
        wmode = int(rng.integers(0, 3))
        if ONLY >= 0 and case != ONLY:
            continue
        outs, states = {}, {}
        for layout in LAYOUTS:
            g = ml.Graph(eng, V, delay_windows=layout)
            g.add("x", "input")
            g.add("dt", "input")
            g.add("d", "proc", kind, ["x", "dt"], max_delay=dmax)
            g.add_output("d")
            g.compile()
            ns = g.num_state("d") if hasattr(g, "num_state") else None
            ring = 1 << int(np.ceil(np.log2(max(64, int(dmax) + 64))))
            if wmode == 1:
                g.set_state("d", 0, np.full(V, int(case * 7 + 3) % ring, np.uint32))
            elif wmode == 2:
                w = np.zeros(V, np.uint32)
                w[:64] = ((np.arange(64, dtype=np.uint32) * 5 + case) % ring)[:min(64, V)]
                g.set_state("d", 0, w)
            o = []
            for k in range(launches):
                sl = slice(k * 64 * T, (k + 1) * 64 * T)
                o.append(g.process_host(T, {"x": np.ascontiguousarray(x[:, sl]), "dt": np.ascontiguousarray(d[:, sl])}, Layout.QUAD)[0])
            outs[layout] = np.concatenate(o, 1)
            st, i = [], 0
            while True:
                try:
                    st.append(g.get_state("d", i))
                    i += 1
                except ml.MlgpuError:
                    break
            states[layout] = np.stack(st)
            g.close()
        a, b = outs[0].view(np.uint32), outs[TEST_LAYOUT].view(np.uint32)
        nan = np.isnan(outs[0]) & np.isnan(outs[TEST_LAYOUT])
        diff = int(((a != b) & ~nan).sum()) + int((states[0] != states[TEST_LAYOUT]).sum())
        nonzero += int((outs[0] != 0).sum())
        total += outs[0].size
        if diff:
            bad += 1
            print(f"case {case}: kind {int(kind)} V {V} T {T} launches {launches} max delay {dmax} delay mode {mode} write-index mode {wmode}: {diff} words differ")
            if ONLY >= 0:
                where = np.argwhere((a != b) & ~nan)
                for v, n in where[:24]:
                    print(f"  voice {v} sample {n} (vector {n // 64}, launch {n // (64 * T)}, in trip {n % 8}): layout 0 {outs[0][v, n]!r} layout {TEST_LAYOUT} {outs[TEST_LAYOUT][v, n]!r}; delay times around it {d[v, max(0, n - 20):n + 2].tolist()}")
                print("  state words that differ:", np.argwhere(states[0] != states[TEST_LAYOUT]).tolist()[:20])
    print(f"{cases} cases (seed {seed}), {total} output samples, {nonzero / max(1, total):.3f} of them nonzero: {bad} cases with a difference between ring layout {TEST_LAYOUT} and layout 0")
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 120, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
