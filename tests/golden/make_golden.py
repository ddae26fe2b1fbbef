#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the COMPILED REFERENCE (oracle/_ref/libmlref.so).

Run in the build container (where /root/reference exists):   python tests/golden/make_golden.py
The vectors travel with the repo so the oracle and the HIP path can be checked against the
reference's own outputs on boxes where /root/reference does not exist.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from cpu_checkers import Ref  # noqa: E402
from inputs import MULTI_CASES, chain_coeffs, chain_input, multi_case, multi_inputs_audio, op_inputs  # noqa: E402
from madronalib_amd.constants import Op, Proc, RowOp, Vop  # noqa: E402

GOLDEN_CHAINS = {
    "cfg1_sine_lopass": [Proc.SINE_GEN, Proc.LOPASS],
    "cfg3_saw_bandpass_gain": [Proc.SAW_GEN, Proc.BANDPASS, Proc.GAIN],
    "cfg4_noise_lopass8": [Proc.NOISE_GEN] + [Proc.LOPASS] * 8,
    "pulse_hipass_onepole": [Proc.PULSE_GEN, Proc.HIPASS, Proc.ONE_POLE],
    "saw_shelves_bell_dc": [Proc.SAW_GEN, Proc.LO_SHELF, Proc.HI_SHELF, Proc.BELL, Proc.DC_BLOCKER],
}


def synth16full(ref):
    """tests/golden/synth16full.npz: patches.synth16(full=True) (the patch of SURVEY 8d) run with the reference's own objects."""
    from inputs import gate_signal
    from madronalib_amd.sharding import cfg5_voice_params
    import madronalib_amd as ml
    V, T = 6, 12
    params, coeffs, seeds = cfg5_voice_params(0, V, V, ml, full=True)
    gate = gate_signal(V, 64 * T, seed=21)
    out, _ = ref.synth16full_run(params, coeffs, seeds, gate)
    np.savez_compressed(os.path.join(HERE, "synth16full.npz"), gate=gate, out=out, seeds=seeds,
                        **{"p_" + k: np.asarray(v, np.float32) for k, v in params.items()}, **{"c_" + k: v for k, v in coeffs.items()})


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "synth16full":   # add this one file without touching the others
        return synth16full(Ref())
    ref = Ref()
    # ---- elementwise ops: 512 inputs each ----
    d = {}
    for op in Op.UNARY + Op.BINARY + Op.TERNARY:
        a, b, c = op_inputs(op, 64 * 8)
        d[f"op{op}_a"] = np.ascontiguousarray(a).view(np.uint32)
        if b is not None:
            d[f"op{op}_b"] = np.ascontiguousarray(b).view(np.uint32)
        if c is not None:
            d[f"op{op}_c"] = np.ascontiguousarray(c).view(np.uint32)
        d[f"op{op}_out"] = ref.op(op, a, b, c)
    rows = (np.random.default_rng(3).standard_normal(64 * 6) * 100).astype(np.float32)
    rows[64:128] = -np.abs(rows[64:128])
    d["rows"] = rows
    for ro in (RowOp.SUM, RowOp.MEAN, RowOp.MAX, RowOp.MIN):
        d[f"rowop{ro}"] = ref.row_reduce(ro, rows)
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **d)

    # ---- single processors and chains: V=8 voices, T=6 vectors, two consecutive calls ----
    d = {}
    cases = {f"proc{k}": [k] for k in Proc.ALL}
    cases.update(GOLDEN_CHAINS)
    V, T = 8, 6
    for name, procs in cases.items():
        co = chain_coeffs(ref, procs, V, seed=5)
        sig, const = chain_input(procs, V, T, seed=int(procs[0]))
        st = ref.chain_clear(procs, V)
        if procs[0] == Proc.NOISE_GEN:
            st[0] = np.arange(V, dtype=np.uint32)
        if procs[0] == Proc.ONE_SHOT_GEN:
            st[1] = 1
        d[name + "_procs"] = np.asarray(procs, np.int32)
        d[name + "_coeffs"] = co
        d[name + "_state0"] = st.copy()
        if sig is not None:
            d[name + "_in_signal"] = sig
        if const is not None:
            d[name + "_in_const"] = const
        d[name + "_out1"] = ref.chain_process(procs, T, co, st, sig, const)
        d[name + "_state1"] = st.copy()
        d[name + "_out2"] = ref.chain_process(procs, T, co, st, sig, const)
        d[name + "_state2"] = st.copy()
    d["impulse_table"] = ref.impulse_table()
    np.savez_compressed(os.path.join(HERE, "chains.npz"), **d)
    # ---- the other operator() forms, vector-rate ramps, index generators: V=6, 2 calls of T=16 vectors ----
    d = {}
    V, T = 6, 16
    for name in MULTI_CASES:
        case = multi_case(ref, name, V, 2 * T, seed=8)
        kind = case["kind"]
        st = ref.chain_clear([kind], V)
        d[name + "_kind"] = np.int32(kind)
        d[name + "_coeffs"] = case["coeffs"]
        d[name + "_state0"] = st.copy()
        d[name + "_rates"] = np.array([r == "control" for r, _ in case["inputs"]])
        for i, (_, a) in enumerate(case["inputs"]):
            d[f"{name}_in{i}"] = a
        ins = multi_inputs_audio(case, 2 * T)
        for call in range(2):
            sl = slice(call * 64 * T, (call + 1) * 64 * T)
            d[f"{name}_out{call + 1}"] = ref.proc_multi(kind, T, case["coeffs"], st, [np.ascontiguousarray(x[:, sl]) for x in ins])
            d[f"{name}_state{call + 1}"] = st.copy()
    rng = np.random.default_rng(12)
    d["vop_a"] = (rng.standard_normal((V, T)) * 10.0 ** rng.integers(-3, 4, (V, T))).astype(np.float32)
    d["vop_b"] = (rng.standard_normal((V, T)) * 10.0 ** rng.integers(-3, 4, (V, T))).astype(np.float32)
    for vop in (Vop.COLUMN_INDEX, Vop.RANGE_OPEN, Vop.RANGE_CLOSED, Vop.INTERPOLATE_LINEAR):
        d[f"vop{vop}_out"] = ref.vop(vop, V, T, np.repeat(d["vop_a"], 64, 1), np.repeat(d["vop_b"], 64, 1))
    np.savez_compressed(os.path.join(HERE, "multi.npz"), **d)
    # ---- row plumbing and routing: the reference's template instantiations of tests/rows_cases.py ----
    from rows_cases import ROWS_CASES, case_inputs
    d = {}
    for name in ROWS_CASES:
        ins = case_inputs(name)
        for i, x in enumerate(ins):
            d[f"{name}|in{i}"] = x
        d[f"{name}|out"] = ref.rows_case(name, ins)
    np.savez_compressed(os.path.join(HERE, "rows.npz"), **d)
    # ---- delay lines and the feedback composites ----
    from graph_oracle import ring_len
    from inputs import DELAY_CASES, delay_case, lcg_noise
    d = {}
    V, T = 6, 10
    for name in DELAY_CASES:
        c = delay_case(ref, name, V, 2 * T, seed=5)
        rings = 2 if c["kind"] == Proc.PITCHBENDABLE_DELAY else 1
        st, mem = c["state0"].copy(), np.zeros((V, rings, ring_len(c["max_delay"])), np.float32)
        d[name + "_kind"], d[name + "_max_delay"], d[name + "_state0"] = np.int32(c["kind"]), np.float32(c["max_delay"]), c["state0"]
        for i, a in enumerate(c["inputs"]):
            d[f"{name}_in{i}"] = a
        for call in range(2):
            sl = slice(call * 64 * T, (call + 1) * 64 * T)
            d[f"{name}_out{call + 1}"] = ref.delay_process(c["kind"], T, st, mem, [np.ascontiguousarray(a[:, sl]) for a in c["inputs"]])
            d[f"{name}_state{call + 1}"] = st.copy()
    T = 30
    x = lcg_noise(np.array([5], np.uint32), 64 * T)[0]
    dsig = (150.0 + 60.0 * np.sin(np.arange(64 * T) * 0.003)).astype(np.float32)
    d["comp_x"], d["comp_dsig"] = x, dsig
    d["allpass0"] = ref.allpass_run(0, 0.7, 400.0, 101.0, None, x)
    d["allpass1"] = ref.allpass_run(1, 0.7, 400.0, 77.37, None, x)
    d["allpass2"] = ref.allpass_run(2, 0.7, 400.0, 0.0, dsig, x)
    d["fdnL"], d["fdnR"] = ref.fdn4_run([133.0, 201.0, 307.0, 419.0], [0.2, 0.15, 0.1, 0.05], [0.8, 0.75, 0.7, 0.65], 512.0, x * np.float32(0.1))
    d["fbdelay"] = ref.feedback_delay_run(0.6, 1000.0, ref.make_coeffs("lopass", 0.08, 0.9), dsig + np.float32(150.0), x)
    np.savez_compressed(os.path.join(HERE, "delays.npz"), **d)
    # ---- Downsampler / Upsampler ----
    d = {"x": lcg_noise(np.arange(4, dtype=np.uint32) + 21, 64 * 8)}
    for octaves in (1, 3):
        d[f"down{octaves}"] = ref.resample(octaves, False, np.zeros((octaves * 9, 4), np.float32), d["x"])
        d[f"up{octaves}"] = ref.resample(octaves, True, np.zeros((octaves * 9, 4), np.float32), d["x"])
    np.savez_compressed(os.path.join(HERE, "resample.npz"), **d)
    # ---- rate regions: Upsample2xFunction / Downsample2xFunction around a stateful fn ----
    from inputs import region_case
    x, m, freq = region_case(4, 10)
    co = ref.make_coeffs("lopass", 0.2, 0.8)
    opc = ref.make_coeffs("onepole", 0.3)
    np.savez_compressed(os.path.join(HERE, "regions.npz"), x=x, m=m, freq=freq, co=co, up=ref.rate_function_run(True, freq, co, x, m),
                        down=ref.rate_function_run(False, freq, co, x, m), opc=opc,
                        ap_up=ref.rate_allpass_run(True, 0.6, 300.0, 171.0, opc, x), ap_down=ref.rate_allpass_run(False, 0.6, 300.0, 171.0, opc, x),
                        **{f"nested_{'u' if o else 'd'}{'u' if i else 'd'}": ref.rate_nested_run(o, i, co, opc, x) for o in (True, False) for i in (True, False)})
    # ---- BASELINE configs[4]: the synth16 voice written with the reference's objects ----
    from inputs import gate_signal
    from madronalib_amd.sharding import cfg5_voice_params
    import madronalib_amd as ml
    V, T = 6, 12
    params, coeffs, seeds = cfg5_voice_params(0, V, V, ml)
    gate = gate_signal(V, 64 * T, seed=21)
    out, _ = ref.synth16_run(params, coeffs, seeds, gate)
    np.savez_compressed(os.path.join(HERE, "synth16.npz"), gate=gate, out=out, seeds=seeds, **{"p_" + k: np.asarray(v, np.float32) for k, v in params.items()},
                        **{"c_" + k: v for k, v in coeffs.items()})
    synth16full(ref)
    for f in ("ops.npz", "chains.npz", "multi.npz", "rows.npz", "delays.npz", "resample.npz", "regions.npz", "synth16.npz", "synth16full.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
