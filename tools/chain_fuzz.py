#!/usr/bin/env python3
"""Random voice-bank CHAINS (mlgpu_bank: a head - generator, noise or a streamed signal - and 0 .. 5 processors behind it, per-voice
coefficients) on the device against the oracle's chain evaluator: ragged and whole banks, 1 .. 40 DSPVectors, 2 .. 3 launches with carried
state, every output layout; the chains outside the ahead-of-time catalogue are fused at run time (hiprtc), every fifth case also runs
processor by processor through memory (mlgpu_engine_set_jit off). Output and state bit for bit (Peak / RMS: the hardware-reciprocal tolerance).
    python tools/chain_fuzz.py [cases] [seed]"""
import os
import sys
import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import madronalib_amd as ml                                    # noqa: E402
from madronalib_amd.constants import Layout, Proc               # noqa: E402
from cpu_checkers import Oracle                                # noqa: E402
from inputs import chain_coeffs, chain_input                   # noqa: E402

TAIL = [k for k in Proc.ALL if k not in Proc.GENERATORS and k not in Proc.HW_APPROX and k != Proc.ADSR and k != Proc.SAMPLE_ACCURATE_LINEAR_GLIDE]
HEADS = list(Proc.GENERATORS) + [Proc.ADSR, Proc.SAMPLE_ACCURATE_LINEAR_GLIDE] + TAIL


def run(cases, seed, eng=None):
    rng = np.random.default_rng(seed)
    eng = eng or ml.Engine(0)
    orc = Oracle()
    bad, fusedN = 0, 0
    for case in range(cases):
        procs = [int(rng.choice(HEADS))] + [int(rng.choice(TAIL)) for _ in range(int(rng.integers(0, 6)))]
        V = int(rng.integers(1, 600)) if rng.random() < 0.6 else 64 * int(rng.integers(1, 10))
        T = int(rng.integers(1, 41))
        calls = int(rng.integers(2, 4))
        layout = [Layout.QUAD, Layout.ROWS, Layout.VOICE_MAJOR][int(rng.integers(0, 3))]
        co = chain_coeffs(orc, procs, V, seed=case + seed * 1000)
        sig, const = chain_input(procs, V, T, seed=case)
        st = orc.chain_clear(procs, V)
        if procs[0] == Proc.NOISE_GEN:
            st[0] = np.arange(V, dtype=np.uint32)
        if procs[0] == Proc.ONE_SHOT_GEN:
            st[1] = 1
        nojit = case % 5 == 4
        eng.set_jit(not nojit)
        try:
            bank = eng.bank(procs, V)
            bank.set_all_coeffs(co)
            bank.set_all_state(st.copy())
            if const is not None:
                bank.set_input_const(const)
            outs = [bank.process_host(T, sig, layout) for _ in range(calls)]
            gst = bank.get_all_state()
            fusedN += int(bank.fused)
            bank.close()
        finally:
            eng.set_jit(True)
        diff = 0
        for got in outs:
            want = orc.chain_process(procs, T, co, st, sig, const, n_threads=4)
            nan = np.isnan(got) & np.isnan(want)
            diff += int(((np.ascontiguousarray(got).view(np.uint32) != np.ascontiguousarray(want).view(np.uint32)) & ~nan).sum())
        diff += int((np.ascontiguousarray(gst).view(np.uint32) != np.ascontiguousarray(st).view(np.uint32)).sum())
        if diff:
            bad += 1
            print(f"case {case}: chain {procs} V {V} T {T} calls {calls} layout {int(layout)} run-time fusion {not nojit}: {diff} words differ")
    print(f"{cases} chains (seed {seed}), {fusedN} of them as one fused kernel: {bad} with a difference from the oracle")
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
