"""Pin the plain-C oracle (oracle/ml_oracle.c) against the compiled reference itself.

Runs wherever oracle/_ref/libmlref.so exists (this container builds it from /root/reference;
the GPU box gets the prebuilt .so). Bit-exact except the two hardware-approximate ops.
"""
import numpy as np
import pytest

from inputs import (MULTI_CASES, assert_bits_equal, assert_rel_close, chain_coeffs, chain_input, is_float_result,
                    multi_case, multi_inputs_audio, op_inputs)
from madronalib_amd.constants import Op, Proc, RowOp, Vop

HW_REL = 2.0 ** -11  # rcpps/rsqrtps: |rel err| <= 1.5 * 2^-12 per Intel; SURVEY 8d's 2^-11


@pytest.mark.parametrize("op", Op.UNARY + Op.BINARY + Op.TERNARY)
def test_op_matches_reference(oracle, ref, op):
    a, b, c = op_inputs(op)
    want = ref.op(op, a, b, c)
    got = oracle.op(op, a, b, c)
    if op in Op.HW_APPROX:
        # exclude inputs where the approximation itself is ill-defined (0, inf, nan, denormal)
        ok = np.isfinite(want.view(np.float32)) & np.isfinite(got.view(np.float32)) & (want.view(np.float32) != 0)
        assert_rel_close(got.view(np.float32)[ok], want.view(np.float32)[ok], HW_REL, f"op {op}")
    else:
        assert_bits_equal(got, want, is_float_result(op), f"op {op}")


@pytest.mark.parametrize("op", [Op.ADD, Op.SUBTRACT, Op.MULTIPLY, Op.DIVIDE, Op.POW, Op.POW_APPROX, Op.MIN, Op.MAX])
def test_rows1_matches_reference(oracle, ref, op):
    a, b, _ = op_inputs(op, 64 * 8)
    a = np.abs(a) if op in (Op.POW, Op.POW_APPROX) else a
    assert_bits_equal(oracle.op_rows1(op, a, b[:64]), ref.op_rows1(op, a, b[:64]), True, f"rows1 {op}")


@pytest.mark.parametrize("rowop", [RowOp.SUM, RowOp.MEAN, RowOp.MAX, RowOp.MIN])
def test_row_reduce_matches_reference(oracle, ref, rowop):
    rng = np.random.default_rng(3)
    rows = rng.standard_normal(64 * 33).astype(np.float32) * np.float32(100)
    rows[64:128] = -np.abs(rows[64:128])  # all-negative row exercises the FLT_MIN seed quirk of max()
    assert_bits_equal(oracle.row_reduce(rowop, rows), ref.row_reduce(rowop, rows), True, f"rowop {rowop}")


@pytest.mark.parametrize("kind", Proc.ALL)
def test_single_proc_matches_reference(oracle, ref, kind):
    V, T = 24, 12
    procs = [kind]
    co = chain_coeffs(ref, procs, V, seed=5)
    assert_bits_equal(chain_coeffs(oracle, procs, V, seed=5), co, True, "makeCoeffs")
    sig, const = chain_input(procs, V, T, seed=kind)
    st_r = ref.chain_clear(procs, V)
    st_o = oracle.chain_clear(procs, V)
    assert_bits_equal(st_o, st_r, False, "clear() state")
    assert_bits_equal(oracle.chain_default_state(procs, V), ref.chain_default_state(procs, V), False, "default state")
    if kind == Proc.NOISE_GEN:
        st_r[0] = st_o[0] = np.arange(V, dtype=np.uint32)
    if kind == Proc.ONE_SHOT_GEN:
        st_r[1] = st_o[1] = 1  # trigger(): gate = 1
    want = ref.chain_process(procs, T, co, st_r, sig, const)
    got = oracle.chain_process(procs, T, co, st_o, sig, const)
    if kind in Proc.HW_APPROX:
        assert_rel_close(got, want, HW_REL, f"proc {kind}")
    else:
        assert_bits_equal(got, want, True, f"proc {kind} output")
    assert_bits_equal(st_o, st_r, False, f"proc {kind} final state")


@pytest.mark.parametrize("kind", [k for k in Proc.ALL if k not in Proc.HW_APPROX and k != Proc.NOISE_GEN])
def test_single_proc_hostile_input_matches_reference(oracle, ref, kind):
    """Pins the oracle on infinities, NaNs, denormals, huge values and raw bit patterns as the input of every processor
    (tests/test_gpu_parity.py::test_single_proc_hostile_input drives the device with the same kind of signal)."""
    from inputs import general_floats, lcg_noise
    V, T = 24, 8
    procs = [kind]
    co = chain_coeffs(ref, procs, V, seed=5)
    S = 64 * T
    x = lcg_noise(np.arange(V, dtype=np.uint32) + np.uint32(kind + 3), S)
    g = general_floats(V * S, int(kind) + 3).reshape(V, S)
    rng = np.random.default_rng(int(kind) + 3)
    mask = rng.random((V, S)) < 0.05
    x[mask] = g[mask]
    x[1::7, S // 3:S // 3 + 40] = g[1::7, :40]
    st_r = ref.chain_clear(procs, V)
    if kind == Proc.ONE_SHOT_GEN:
        st_r[1] = 1
    st_o = st_r.copy()
    want = ref.chain_process(procs, T, co, st_r, x, None)
    got = oracle.chain_process(procs, T, co, st_o, x, None)
    assert_bits_equal(got, want, True, f"hostile proc {kind}")
    bothnan = np.isnan(st_o.view(np.float32)) & np.isnan(st_r.view(np.float32))
    assert ((st_o == st_r) | bothnan).all(), f"hostile proc {kind} state"


@pytest.mark.parametrize("garbage_state", [False, True])
def test_adsr_hostile_gates_match_reference(oracle, ref, garbage_state):
    """Pins the oracle's ADSR on the inputs tests/test_gpu_parity.py::test_adsr_hostile_gates_and_states drives the device with:
    negative / -0 / NaN / inf / denormal gates, and states the envelope never produces, set into the reference's own object."""
    from inputs import hostile_adsr_state, hostile_gate
    V, T = 48, 12
    procs = [Proc.ADSR]
    co = chain_coeffs(ref, procs, V, seed=3)
    rng = np.random.default_rng(78)
    st_r = hostile_adsr_state(ref.chain_clear(procs, V), rng) if garbage_state else ref.chain_clear(procs, V)
    st_o = st_r.copy()
    for c in range(3):
        gate = hostile_gate(V, 64 * T, seed=100 * c + 1)
        want = ref.chain_process(procs, T, co, st_r, gate, None)
        got = oracle.chain_process(procs, T, co, st_o, gate, None)
        assert_bits_equal(got, want, True, f"hostile ADSR call {c}")
        assert_bits_equal(st_o, st_r, False, f"hostile ADSR state call {c}")


CHAINS = {
    "cfg1_sine_lopass": [Proc.SINE_GEN, Proc.LOPASS],
    "cfg3_saw_bandpass_gain": [Proc.SAW_GEN, Proc.BANDPASS, Proc.GAIN],
    "cfg4_noise_lopass8": [Proc.NOISE_GEN] + [Proc.LOPASS] * 8,
    "pulse_hipass_onepole": [Proc.PULSE_GEN, Proc.HIPASS, Proc.ONE_POLE],
    "saw_shelves_bell_dc": [Proc.SAW_GEN, Proc.LO_SHELF, Proc.HI_SHELF, Proc.BELL, Proc.DC_BLOCKER],
}


@pytest.mark.parametrize("name", list(CHAINS))
def test_chain_matches_reference(oracle, ref, name):
    procs = CHAINS[name]
    V, T = 16, 20
    co = chain_coeffs(ref, procs, V, seed=11)
    sig, const = chain_input(procs, V, T, seed=2)
    st_r = ref.chain_clear(procs, V)
    st_o = st_r.copy()
    if procs[0] == Proc.NOISE_GEN:
        st_r[0] = st_o[0] = np.arange(V, dtype=np.uint32)
    # two calls: the second resumes from carried state
    for _ in range(2):
        want = ref.chain_process(procs, T, co, st_r, sig, const)
        got = oracle.chain_process(procs, T, co, st_o, sig, const, n_threads=3)
        assert_bits_equal(got, want, True, name)
        assert_bits_equal(st_o, st_r, False, name + " state")


def test_coefficient_makers_match_reference(oracle, ref):
    rng = np.random.default_rng(0)
    for _ in range(200):
        om, k, A = rng.uniform(0.0005, 0.49), rng.uniform(0.01, 3), rng.uniform(0.1, 8)
        for name, args in [("lopass", (om, k)), ("hipass", (om, k)), ("bandpass", (om, k)), ("loshelf", (om, k, A)),
                           ("hishelf", (om, k, A)), ("bell", (om, k, A)), ("onepole", (om,)),
                           ("adsr", (rng.uniform(0, .1), rng.uniform(0, .1), rng.random(), rng.uniform(0, .1), 44100.0))]:
            assert_bits_equal(oracle.make_coeffs(name, *args), ref.make_coeffs(name, *args), True, name)
        assert oracle.dcblocker_coeffs(om) == ref.dcblocker_coeffs(om)
        assert oracle.db_to_gain(A) == ref.db_to_gain(A)
    assert_bits_equal(oracle.impulse_table(), ref.impulse_table(), True, "ImpulseGen table")
    assert_bits_equal(oracle.range_closed(-np.pi, np.pi), ref.range_closed(-np.pi, np.pi), True, "rangeClosed")
    assert_bits_equal(oracle.range_open(0.25, 9.0), ref.range_open(0.25, 9.0), True, "rangeOpen")


@pytest.mark.parametrize("name", MULTI_CASES)
def test_multi_input_forms_match_reference(oracle, ref, name):
    """PulseGen(freq, width), Lopass(x, omega, k), shelves with coefficient signals, Interpolator1, LinearGlide:
    two consecutive calls, outputs and final state."""
    V, T = 12, 40
    case_r = multi_case(ref, name, V, 2 * T, seed=4)
    case_o = multi_case(oracle, name, V, 2 * T, seed=4)
    assert_bits_equal(case_o["coeffs"], case_r["coeffs"], True, name + " coeffs")
    ins_r, ins_o = multi_inputs_audio(case_r, 2 * T), multi_inputs_audio(case_o, 2 * T)
    kind = case_r["kind"]
    st_r, st_o = ref.chain_clear([kind], V), oracle.chain_clear([kind], V)
    assert_bits_equal(st_o, st_r, False, name + " clear() state")
    assert_bits_equal(oracle.chain_default_state([kind], V), ref.chain_default_state([kind], V), False, name + " default state")
    for call in range(2):
        sl = slice(call * 64 * T, (call + 1) * 64 * T)
        a_r = [np.ascontiguousarray(x[:, sl]) for x in ins_r]
        a_o = [np.ascontiguousarray(x[:, sl]) for x in ins_o]
        for x, y in zip(a_r, a_o):
            assert_bits_equal(y, x, True, name + " inputs")
        want = ref.proc_multi(kind, T, case_r["coeffs"], st_r, a_r)
        got = oracle.proc_multi(kind, T, case_o["coeffs"], st_o, a_o)
        assert_bits_equal(got, want, True, f"{name} call {call}")
        assert_bits_equal(st_o, st_r, False, f"{name} state after call {call}")


@pytest.mark.parametrize("name", ("pulse2", "interp1", "linear_glide", "linear_glide_long", "tempo_lock"))
def test_multi_input_forms_hostile_inputs_match_reference(oracle, ref, name):
    """Pins the oracle on tests/test_gpu_graph.py::test_multi_input_forms_hostile_inputs' kind of input."""
    from test_gpu_graph import hostile_multi_case
    V, T = 18, 12
    case_r = hostile_multi_case(ref, name, V, T, seed=33)
    case_o = hostile_multi_case(oracle, name, V, T, seed=33)
    ins_r, ins_o = multi_inputs_audio(case_r, T), multi_inputs_audio(case_o, T)
    kind = case_r["kind"]
    st_r = ref.chain_clear([kind], V)
    st_o = st_r.copy()
    want = ref.proc_multi(kind, T, case_r["coeffs"], st_r, ins_r)
    got = oracle.proc_multi(kind, T, case_o["coeffs"], st_o, ins_o)
    assert_bits_equal(got, want, True, f"hostile {name}")
    bothnan = np.isnan(st_o.view(np.float32)) & np.isnan(st_r.view(np.float32))
    assert ((st_o == st_r) | bothnan).all(), f"hostile {name} state"


def test_pulse_gen_absurd_widths_match_reference(oracle, ref):
    """PulseGen(freq, width) with widths whose shifted phase leaves the int32 range (fractionalPart through cvttps2dq's
    0x80000000), NaN and infinite widths: pins the oracle on what tests/test_gpu_graph.py::test_pulse_gen_with_absurd_widths
    drives the device with."""
    V, T = 16, 6
    rng = np.random.default_rng(9)
    f = (20.0 * (400.0 ** rng.random(V)) / 48000.0).astype(np.float32)
    w = np.array([3.0e9, -5.0e9, 2.0 ** 30, -(2.0 ** 30), 2.0 ** 31, np.inf, -np.inf, np.nan, 1.0e20, -0.0, 0.0, 1.0, 1.5, -0.25, 2.0 ** 29, 0.3], np.float32)
    freq, width = np.repeat(f[:, None], 64 * T, 1), np.repeat(w[:, None], 64 * T, 1)
    co = np.full((1, V), 0.5, np.float32)
    st_r = ref.chain_clear([Proc.PULSE_GEN], V)
    st_r[0] = rng.integers(0, 2 ** 32, V, dtype=np.uint64).astype(np.uint32)
    st_o = st_r.copy()
    for call in range(2):
        want = ref.proc_multi(Proc.PULSE_GEN, T, co, st_r, [freq, width])
        got = oracle.proc_multi(Proc.PULSE_GEN, T, co, st_o, [freq, width])
        assert_bits_equal(got, want, True, f"pulse absurd widths call {call}")
        assert_bits_equal(st_o, st_r, False, "state")


@pytest.mark.parametrize("vop", [Vop.COLUMN_INDEX, Vop.RANGE_OPEN, Vop.RANGE_CLOSED, Vop.INTERPOLATE_LINEAR])
def test_vector_generators_match_reference(oracle, ref, vop):
    V, T = 9, 17
    rng = np.random.default_rng(vop)
    a = np.repeat((rng.standard_normal((V, T)) * 10.0 ** rng.integers(-3, 4, (V, T))).astype(np.float32), 64, 1)
    b = np.repeat((rng.standard_normal((V, T)) * 10.0 ** rng.integers(-3, 4, (V, T))).astype(np.float32), 64, 1)
    assert_bits_equal(oracle.vop(vop, V, T, a, b), ref.vop(vop, V, T, a, b), True, f"vop {vop}")


def test_glide_coefficient_makers_match_reference(oracle, ref):
    for t in (0.0, 1.0, 63.9, 64.0, 100.0, 1000.0, 4096.0, 12345.6, -5.0):
        assert_bits_equal(oracle.make_coeffs("linear_glide", t), ref.make_coeffs("linear_glide", t), False, "LinearGlide")
        assert_bits_equal(oracle.make_coeffs("sample_glide", t), ref.make_coeffs("sample_glide", t), False, "SampleAccurate")


from rows_cases import ROWS_CASES, case_inputs, run as run_rows_case  # noqa: E402


@pytest.mark.parametrize("name", list(ROWS_CASES))
def test_row_plumbing_and_routing_match_reference(oracle, ref, name):
    """repeatRows ... separateRows, addRows, normalize, rowIndex, mix, (de)multiplex(Linear): the rule-based calls
    composed as the host does == the reference's own template instantiations."""
    ins = case_inputs(name)
    assert_bits_equal(run_rows_case(oracle, name, ins), ref.rows_case(name, ins), True, name)


from graph_oracle import evaluate_stream, new_stream_state, ring_len  # noqa: E402
from inputs import DELAY_CASES, delay_case, lcg_noise, stepped  # noqa: E402
from madronalib_amd import patches  # noqa: E402


@pytest.mark.parametrize("name", DELAY_CASES)
def test_delay_lines_match_reference(oracle, ref, name):
    """IntegerDelay / FractionalDelay / PitchbendableDelay: the per-sample restatement against the reference objects'
    own operator() forms (incl. the block form for constant delays), outputs, state and ring contents, two calls."""
    V, T = 10, 12
    c = delay_case(oracle, name, V, 2 * T, seed=3)
    cr = delay_case(ref, name, V, 2 * T, seed=3)
    assert_bits_equal(c["state0"], cr["state0"], False, name + " setDelayInSamples state")
    rings = 2 if c["kind"] == Proc.PITCHBENDABLE_DELAY else 1
    L = ring_len(c["max_delay"])
    st_o, st_r = c["state0"].copy(), c["state0"].copy()
    mem_o, mem_r = np.zeros((V, rings, L), np.float32), np.zeros((V, rings, L), np.float32)
    for call in range(2):
        sl = slice(call * 64 * T, (call + 1) * 64 * T)
        ins = [np.ascontiguousarray(a[:, sl]) for a in c["inputs"]]
        want = ref.delay_process(c["kind"], T, st_r, mem_r, ins)
        got = oracle.delay_process(c["kind"], T, st_o, mem_o, ins)
        assert_bits_equal(got, want, True, f"{name} call {call}")
        assert_bits_equal(st_o, st_r, False, f"{name} state after call {call}")
        assert_bits_equal(mem_o, mem_r, True, f"{name} ring contents after call {call}")


def test_allpass1_and_delay_coefficients_match_reference(oracle, ref):
    for d in (0.0, 0.3, 0.618, 0.9999, 1.0, 1.618, 5.5, 63.99, 64.0, 100.617, 1234.25):
        assert oracle.allpass1_coeffs(d) == ref.allpass1_coeffs(d)
        assert_bits_equal(oracle.fractional_delay_state(d), ref.fractional_delay_state(d), False, f"setDelayInSamples({d})")


@pytest.mark.parametrize("which,kind,d", [(0, Proc.INTEGER_DELAY, 101.0), (1, Proc.FRACTIONAL_DELAY, 77.37), (2, Proc.PITCHBENDABLE_DELAY, 0.0)])
def test_allpass_composite_matches_reference(oracle, ref, which, kind, d):
    """Allpass<IntegerDelay / FractionalDelay / PitchbendableDelay> as a graph with a feedback node == the class."""
    T, gain, max_delay = 30, 0.7, 400.0
    x = lcg_noise(np.array([5], np.uint32), 64 * T)
    dsig = (150.0 + 60.0 * np.sin(np.arange(64 * T) * 0.003))[None, :].astype(np.float32)
    desc = [dict(name="x", type="input")] + ([dict(name="dl", type="input")] if which == 2 else [])
    sub, out = patches.allpass("ap_", "x", kind, max_delay, "dl" if which == 2 else None)
    desc += sub
    st = new_stream_state(oracle, desc, 1)
    if which == 0:
        st["ap_delay"][1] = np.uint32(int(d - 64))                 # Allpass::setDelayInSamples(d) -> mDelay.setDelayInSamples(d - 64)
    if which == 1:
        st["ap_delay"][3:5, 0] = oracle.fractional_delay_state(float(np.float32(d) - np.float32(64.0))).view(np.uint32)
    (got,) = evaluate_stream(oracle, desc, [out], 1, T, {"x": x, "dl": dsig}, {"ap_gain": gain}, {}, st)
    want = ref.allpass_run(which, gain, max_delay, d, dsig[0] if which == 2 else None, x[0])
    assert_bits_equal(got[0], want, True, f"Allpass<{kind}>")


def test_fdn_composite_matches_reference(oracle, ref):
    T = 40
    x = lcg_noise(np.array([9], np.uint32), 64 * T) * np.float32(0.1)
    times, omegas, gains = [133.0, 201.0, 307.0, 419.0], [0.2, 0.15, 0.1, 0.05], [0.8, 0.75, 0.7, 0.65]
    desc = [dict(name="x", type="input")]
    sub, outs = patches.fdn(4, "x", 512.0)
    desc += sub
    st = new_stream_state(oracle, desc, 1)
    coeffs = {}
    for n in range(4):
        st[f"fdn_delay{n}"][1] = np.uint32(max(1, int(times[n] - 64)))   # setDelaysInSamples, MLDSPFilters.h:1172-1181
        coeffs[f"fdn_filter{n}"] = oracle.make_coeffs("onepole", omegas[n]).reshape(2, 1)
    got = evaluate_stream(oracle, desc, outs, 1, T, {"x": x}, {f"fdn_gain{n}": gains[n] for n in range(4)}, coeffs, st)
    wantL, wantR = ref.fdn4_run(times, omegas, gains, 512.0, x[0])
    assert_bits_equal(got[0][0], wantL, True, "FDN<4> sumL")
    assert_bits_equal(got[1][0], wantR, True, "FDN<4> sumR")
    assert np.abs(wantL).max() > 1e-3


def test_feedback_delay_function_matches_reference(oracle, ref):
    """FeedbackDelayFunction (MLDSPFunctional.h:262-290) around a Lopass."""
    T, fbGain, max_delay = 36, 0.6, 1000.0
    x = lcg_noise(np.array([2], np.uint32), 64 * T)
    dsig = (300.0 + 100.0 * np.sin(np.arange(64 * T) * 0.002))[None, :].astype(np.float32)
    co = oracle.make_coeffs("lopass", 0.08, 0.9)
    desc = [dict(name="x", type="input"), dict(name="dl", type="input"), dict(name="g", type="const", value=fbGain),
            dict(name="c64", type="const", value=64.0), dict(name="vy1", type="feedback", source="delay"),
            dict(name="fb", type="op", kind=Op.MULTIPLY, inputs=["vy1", "g"]), dict(name="sum", type="op", kind=Op.ADD, inputs=["x", "fb"]),
            dict(name="fn", type="proc", kind=Proc.LOPASS, inputs=["sum"]), dict(name="dt", type="op", kind=Op.SUBTRACT, inputs=["dl", "c64"]),
            dict(name="delay", type="proc", kind=Proc.PITCHBENDABLE_DELAY, inputs=["fn", "dt"], max_delay=max_delay)]
    st = new_stream_state(oracle, desc, 1)
    (got,) = evaluate_stream(oracle, desc, ["fn"], 1, T, {"x": x, "dl": dsig}, {}, {"fn": co.reshape(3, 1)}, st)
    want = ref.feedback_delay_run(fbGain, max_delay, co, dsig[0], x[0])
    assert_bits_equal(got[0], want, True, "FeedbackDelayFunction")


@pytest.mark.parametrize("octaves", [0, 1, 2, 3, 5])
@pytest.mark.parametrize("up", [False, True])
def test_resamplers_match_reference(oracle, ref, octaves, up):
    """Downsampler / Upsampler: the stream cascade == the reference classes' block schedule; two calls, carried state."""
    V = 5
    Tin = 2 * (1 << octaves) if not up else 3
    x = lcg_noise(np.arange(V, dtype=np.uint32) + 70, 64 * Tin * 2)
    st_o, st_r = np.zeros((octaves * 9, V), np.float32), np.zeros((octaves * 9, V), np.float32)
    for call in range(2):
        xs = np.ascontiguousarray(x[:, call * 64 * Tin:(call + 1) * 64 * Tin])
        assert_bits_equal(oracle.resample(octaves, up, st_o, xs), ref.resample(octaves, up, st_r, xs), True, f"resample call {call}")
        assert_bits_equal(st_o, st_r, True, "HalfBandFilter state")


@pytest.mark.parametrize("up", [True, False])
def test_rate_functions_match_reference(oracle, ref, up):
    """Upsample2xFunction<2> / Downsample2xFunction<2> (MLDSPFunctional.h:114-213) around a stateful fn: the oracle's block
    restatement == the reference classes, odd vector count included."""
    from inputs import region_case
    x, m, freq = region_case(7, 11, seed=4)
    co = ref.make_coeffs("lopass", 0.2, 0.8)
    assert_bits_equal(oracle.rate_function_run(up, freq, co, x, m), ref.rate_function_run(up, freq, co, x, m), True, f"rate function up={up}")


def test_synth16_graph_evaluator_matches_reference_objects(oracle, ref):
    """BASELINE configs[4]: the node-by-node evaluator the GPU graph is checked against (tests/graph_oracle.py) gives the same
    bits as the patch written in C++ with the reference's own objects (oracle/ref_wrapper.cpp mlref_synth16_run)."""
    from graph_oracle import evaluate
    from inputs import gate_signal
    from madronalib_amd import patches
    from madronalib_amd.sharding import cfg5_voice_params
    import madronalib_amd as ml
    V, T = 50, 20
    params, coeffs, seeds = cfg5_voice_params(0, V, V, ml)
    desc, outs = patches.synth16()
    states = {n["name"]: oracle.chain_clear([n["kind"]], V) for n in desc if n["type"] == "proc"}
    states["noise"][0] = seeds
    gate = gate_signal(V, 64 * T, seed=5)
    (got,) = evaluate(oracle, desc, outs, V, T, {"gate": gate}, params, coeffs, states)
    want, _ = ref.synth16_run(params, coeffs, seeds, gate)
    assert_bits_equal(got, want, True, "synth16 evaluator vs reference objects")
    assert np.abs(want).max() > 0.05


def test_band_limited_oscillators_on_the_knife_edges_match_reference(oracle, ref):
    """SawGen(freq) and PulseGen(freq, width) with phases placed where tests/test_gpu_graph.py::test_oscillator_trips_on_the_knife_edges
    places them - a sample 0 .. 23 units of 2^-32 after a wrap, 0 .. 368 before one, within -128 .. +608 of the pulse's falling
    step - and the same special widths (0, 1, dt, 1 - dt, dt / 2): the device's trip form is compared with the oracle there, so the
    oracle is pinned to the compiled reference on exactly those inputs."""
    V, T = 4096, 2
    rng = np.random.default_rng(77)
    f = (1e-4 * (300.0 ** rng.random(V))).astype(np.float32)
    f[V // 2:] = (1e-3 * (200.0 ** rng.random(V - V // 2))).astype(np.float32)
    f[5::64] = np.float32(1.0 / 16.0)
    w = rng.uniform(0.0, 1.0, V).astype(np.float32)
    w[0::16], w[1::16] = 0.0, 1.0
    w[2::16], w[3::16], w[4::16] = f[2::16], np.float32(1.0) - f[3::16], f[4::16] * np.float32(0.5)
    istep = np.rint(f.astype(np.float64) * 2.0 ** 32).astype(np.uint64)
    v = np.arange(V, dtype=np.uint64)
    k, j = v % 37 + 1, (v // 8) % 24
    wave = (v // 64) % 3
    om = rng.integers(0, 2 ** 32, V, dtype=np.uint64)
    om = np.where((wave == 0) & (v % 4 == 1), (2 ** 32 * 64 - k * istep + j), om)
    om = np.where((wave == 0) & (v % 4 == 3), (2 ** 32 * 64 - k * istep - 16 * j), om)
    wq = np.rint(w.astype(np.float64) * 2.0 ** 32).astype(np.uint64)
    om = np.where((wave == 1) & (v % 2 == 1), (2 ** 32 * 64 + wq - k * istep + 32 * j - 128), om)
    phases = (om % (2 ** 32)).astype(np.uint32)
    freq, width = np.repeat(f[:, None], 64 * T, 1), np.repeat(w[:, None], 64 * T, 1)
    # PulseGen(freq, width)
    co = np.full((1, V), 0.5, np.float32)
    st_r = ref.chain_clear([Proc.PULSE_GEN], V)
    st_r[0] = phases
    st_o = st_r.copy()
    for call in range(2):
        want = ref.proc_multi(Proc.PULSE_GEN, T, co, st_r, [freq, width])
        got = oracle.proc_multi(Proc.PULSE_GEN, T, co, st_o, [freq, width])
        assert_bits_equal(got, want, True, f"pulse on the edges, call {call}")
        assert_bits_equal(st_o, st_r, False, "pulse phases")
    # SawGen(freq)
    st_r = ref.chain_clear([Proc.SAW_GEN], V)
    st_r[0] = phases
    st_o = st_r.copy()
    none = np.zeros((0, V), np.float32)
    for call in range(2):
        want = ref.chain_process([Proc.SAW_GEN], T, none, st_r, None, f)
        got = oracle.chain_process([Proc.SAW_GEN], T, none, st_o, None, f)
        assert_bits_equal(got, want, True, f"saw on the edges, call {call}")
        assert_bits_equal(st_o, st_r, False, "saw phases")
