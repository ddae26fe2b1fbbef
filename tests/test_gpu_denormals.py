"""The denormal regime, pinned on the device in BOTH floating-point modes.

Reference: by default its SSE code honours denormals (default MXCSR); inside an ml::UsingFlushDenormalsToZero scope
(source/DSP/MLDSPUtils.h:51-96, e.g. examples/audio-and-midi/fdtd.cpp:161) MXCSR has FZ | DAZ set. The engine mirrors
both: default mode, and mlgpu_engine_set_flush_denormals(e, 1) (gfx950 MODE.fp_denorm = 0 at kernel entry).

Every case excites a recurrence for a few DSPVectors and then starves it for >= 2000 DSPVectors, so its state decays
through 1e-38 .. 1e-45 to zero — the range where a GPU and SSE disagree if a mode bit is wrong — and compares every
output sample and the final state with the CPU oracle run in the same mode (its MXCSR set the way the reference sets
it), bit for bit. Ahead-of-time kernels (chain / cascade kernels) and hiprtc-fused ones (chains that are not in the
catalogue, graphs with delay rings in both ring layouts) are both covered: they are built with different compiler
invocations. tests/test_denormals_cpu.py pins the oracle itself to the compiled reference in both modes."""
import numpy as np
import pytest

from graph_oracle import evaluate_stream, new_stream_state
from inputs import assert_bits_equal, chain_coeffs, general_floats, is_float_result, lcg_noise, op_inputs
from madronalib_amd import patches
from madronalib_amd.constants import Layout, Op, Proc

pytestmark = pytest.mark.gpu

MODES = [pytest.param(False, id="ieee"), pytest.param(True, id="flush")]


@pytest.fixture(scope="module")
def eng():
    import madronalib_amd as ml
    e = ml.Engine(0)
    yield e
    e.set_flush_denormals(False)
    e.close()


def starved_input(V, live, T, seed, amp=1.0):
    """Noise for `live` DSPVectors, then silence: [V][64 T]."""
    x = np.zeros((V, 64 * T), np.float32)
    x[:, :64 * live] = lcg_noise(np.arange(V, dtype=np.uint32) + seed, 64 * live) * np.float32(amp)
    return x


def denormal_stats(arrs):
    a = np.concatenate([np.abs(np.ascontiguousarray(x).view(np.float32).ravel()) for x in arrs])
    return int(((a > 0) & (a < np.float32(1.17549435e-38))).sum())


# name: (processors, aot-or-jit note). Lopass x8 and NoiseGen-free cascades are the stage-skewed AOT kernel (packed FP32);
# Lopass x3 and the mixed chains have no catalogue entry and are fused by hiprtc when the bank is created.
DECAY_CHAINS = {
    "aot_cascade_lopass8": [Proc.LOPASS] * 8,
    "aot_cascade_hipass4": [Proc.HIPASS] * 4,
    "aot_onepole": [Proc.ONE_POLE],
    "aot_bandpass": [Proc.BANDPASS],
    "aot_dcblocker": [Proc.DC_BLOCKER],
    "aot_integrator_leaky": [Proc.INTEGRATOR],
    "jit_lopass3": [Proc.LOPASS] * 3,
    "jit_onepole_loshelf_bell": [Proc.ONE_POLE, Proc.LO_SHELF, Proc.BELL],
    "jit_hishelf_dc_onepole": [Proc.HI_SHELF, Proc.DC_BLOCKER, Proc.ONE_POLE],
}


@pytest.mark.parametrize("flush", MODES)
@pytest.mark.parametrize("name", list(DECAY_CHAINS))
def test_filters_decay_through_the_denormal_range(eng, oracle, name, flush):
    procs = DECAY_CHAINS[name]
    V, live, chunk, chunks = 96, 3, 256, 8            # 2048 DSPVectors = 131072 samples per voice
    co = chain_coeffs(oracle, procs, V, seed=31)
    if procs == [Proc.INTEGRATOR]:
        co[0] = np.linspace(0.002, 0.2, V).astype(np.float32)    # mLeak: a leaky integrator decays, a perfect one would not
    st = oracle.chain_clear(procs, V)
    eng.set_flush_denormals(flush)
    try:
        bank = eng.bank(procs, V)
        assert bank.fused
        bank.set_all_coeffs(co)
        bank.set_all_state(st)
        seen = 0
        for c in range(chunks):
            x = starved_input(V, live, chunk, seed=5) if c == 0 else np.zeros((V, 64 * chunk), np.float32)
            got = bank.process_host(chunk, x, Layout.QUAD)
            with oracle.flush_denormals(flush):
                want = oracle.chain_process(procs, chunk, co, st, x, None, n_threads=8)
            assert_bits_equal(got, want, True, f"{name} chunk {c} flush={flush}")
            assert_bits_equal(bank.get_all_state(), st, False, f"{name} state after chunk {c} flush={flush}")
            seen += denormal_stats([want, st])
        bank.close()
    finally:
        eng.set_flush_denormals(False)
    if flush:
        # (a flushed recurrence may also park just above FLT_MIN, where every further update rounds to zero: the reference's too)
        assert seen == 0, "an arithmetic result was denormal in flush mode"
    else:
        assert seen > 0, "the case never reached the denormal range: it proves nothing"


@pytest.mark.parametrize("flush", MODES)
def test_adsr_release_and_glides_reach_zero(eng, oracle, flush):
    """ADSR: gate on for two vectors, then off — the release segment is an exponential approach to zero."""
    V, chunk, chunks = 128, 256, 8
    procs = [Proc.ADSR]
    rng = np.random.default_rng(4)
    co = np.stack([oracle.make_coeffs("adsr", rng.uniform(0.0005, 0.01), rng.uniform(0.002, 0.02), rng.uniform(0.2, 0.9), rng.uniform(0.002, 0.03), 48000.0)
                   for _ in range(V)], 1)
    st = oracle.chain_clear(procs, V)
    eng.set_flush_denormals(flush)
    try:
        bank = eng.bank(procs, V)
        bank.set_all_coeffs(co)
        bank.set_all_state(st)
        seen = 0
        for c in range(chunks):
            gate = np.zeros((V, 64 * chunk), np.float32)
            if c == 0:
                gate[:, 10:140] = np.linspace(0.2, 1.0, V, dtype=np.float32)[:, None]
            got = bank.process_host(chunk, gate, Layout.QUAD)
            with oracle.flush_denormals(flush):
                want = oracle.chain_process(procs, chunk, co, st, gate, None, n_threads=8)
            assert_bits_equal(got, want, True, f"ADSR chunk {c} flush={flush}")
            assert_bits_equal(bank.get_all_state(), st, False, f"ADSR state chunk {c} flush={flush}")
            seen += denormal_stats([want])
        bank.close()
    finally:
        eng.set_flush_denormals(False)
    assert (seen == 0) if flush else True


@pytest.mark.parametrize("flush", MODES)
@pytest.mark.parametrize("op", Op.UNARY + Op.BINARY + Op.TERNARY)
def test_ops_on_denormal_operands(eng, oracle, op, flush):
    """Every elementwise op over operands rich in denormals, tiny normals and results that underflow."""
    n = 64 * 96
    a, b, c = op_inputs(op, n, seed=17)
    rng = np.random.default_rng(100 + int(op))
    tiny = (rng.integers(0, 1 << 24, n, dtype=np.uint64).astype(np.uint32) | (rng.integers(0, 2, n).astype(np.uint32) << np.uint32(31)))
    small = np.float32(1e-19) * rng.standard_normal(n).astype(np.float32)
    if op not in Op.INT_INPUT and op not in (Op.SELECT_INT,):
        a = np.where(rng.random(n) < 0.4, tiny.view(np.float32), np.where(rng.random(n) < 0.5, small, a)).astype(np.float32)
        if b is not None:
            b = np.where(rng.random(n) < 0.3, tiny[::-1].view(np.float32), np.where(rng.random(n) < 0.5, small[::-1], b)).astype(np.float32)
    eng.set_flush_denormals(flush)
    try:
        got = eng.op(op, a, b, c)
    finally:
        eng.set_flush_denormals(False)
    with oracle.flush_denormals(flush):
        want = oracle.op(op, a, b, c)
    if op in Op.HW_APPROX:
        g, w = got.view(np.float32), want.view(np.float32)
        ok = np.isfinite(w) & np.isfinite(g) & (np.abs(w) > 1e-30)
        assert np.allclose(g[ok], w[ok], rtol=1.5 * 2.0 ** -11, atol=0)
    else:
        assert_bits_equal(got, want, is_float_result(op), f"op {op} flush={flush}")


def _string_graph(eng, V, windows):
    """Karplus-Strong voice: FractionalDelay -> OnePole -> feedback (bench.py's strings workload), per-voice lengths."""
    import madronalib_amd as ml
    desc = [dict(name="x", type="input"), dict(name="g", type="const", value=0.8),
            dict(name="fb", type="feedback", source="damp"),
            dict(name="fbg", type="op", kind=Op.MULTIPLY, inputs=["fb", "g"]),
            dict(name="sum", type="op", kind=Op.ADD, inputs=["x", "fbg"]),
            dict(name="line", type="proc", kind=Proc.FRACTIONAL_DELAY, inputs=["sum"], max_delay=256.0),
            dict(name="damp", type="proc", kind=Proc.ONE_POLE, inputs=["line"])]
    return desc, ml.Graph(eng, V, desc, ["damp"], delay_windows=windows)


@pytest.mark.parametrize("flush", MODES)
@pytest.mark.parametrize("windows", [pytest.param(False, id="rows"), pytest.param(True, id="windows")])
def test_feedback_delay_graph_decays_to_zero(eng, oracle, windows, flush):
    """A hiprtc-fused graph with a delay ring in HBM and one-vector feedback, both ring layouts: the loop gain is 0.8 x
    the damping filter, so the string dies away through the denormal range within the run."""
    V, chunk, chunks = 64, 150, 8     # 1200 DSPVectors; the vector-by-vector Python evaluator sets the size
    desc, g = _string_graph(eng, V, windows)
    co = np.repeat(oracle.make_coeffs("onepole", 0.2).reshape(2, 1), V, 1)
    g.set_coeffs("damp", [np.ascontiguousarray(r) for r in co])
    st = new_stream_state(oracle, desc, V)
    length = np.linspace(9.3, 120.7, V)
    words = np.stack([oracle.fractional_delay_state(float(np.float32(d))) for d in length], 1)   # [2][V]: delayInt bits, allpass coefficient
    st["line"][3:5, :] = words.view(np.uint32)
    for i in range(st["line"].shape[0]):
        g.set_state("line", i, st["line"][i])
    eng.set_flush_denormals(flush)
    seen = 0
    try:
        for c in range(chunks):
            x = starved_input(V, 2, chunk, seed=9, amp=0.5) if c == 0 else np.zeros((V, 64 * chunk), np.float32)
            (got,) = g.process_host(chunk, {"x": x}, Layout.QUAD)
            with oracle.flush_denormals(flush):
                (want,) = evaluate_stream(oracle, desc, ["damp"], V, chunk, {"x": x}, {}, {"damp": co}, st)
            assert_bits_equal(got, want, True, f"string graph chunk {c} windows={windows} flush={flush}")
            seen += denormal_stats([want])
        for i in range(st["damp"].shape[0]):
            assert_bits_equal(g.get_state("damp", i), st["damp"][i], False, "OnePole state inside the graph")
    finally:
        eng.set_flush_denormals(False)
    assert (seen == 0) if flush else (seen > 0)


@pytest.mark.parametrize("flush", MODES)
def test_fdn_decays_to_zero(eng, oracle, flush):
    """FDN<4> (MLDSPFilters.h:1162-1239): four rings around a Householder matrix, feedback gains < 1."""
    V, chunk, chunks = 32, 150, 8
    desc = [dict(name="x", type="input")]
    sub, outs = patches.fdn(4, "x", 256.0)
    desc += sub
    import madronalib_amd as ml
    g = ml.Graph(eng, V, desc, outs)
    times, omegas, gains = [33.0, 71.0, 107.0, 149.0], [0.2, 0.15, 0.1, 0.05], [0.6, 0.55, 0.5, 0.45]
    coeffs = {f"fdn_filter{n}": np.repeat(oracle.make_coeffs("onepole", omegas[n]).reshape(2, 1), V, 1) for n in range(4)}
    params = {f"fdn_gain{n}": np.full(V, gains[n], np.float32) for n in range(4)}
    for k, v in params.items():
        g.set_param(k, v)
    for k, c in coeffs.items():
        g.set_coeffs(k, [np.ascontiguousarray(r) for r in c])
    st = new_stream_state(oracle, desc, V)
    for n in range(4):
        st[f"fdn_delay{n}"][1] = np.uint32(int(times[n])) + (np.arange(V, dtype=np.uint32) % 13)
        for i in range(st[f"fdn_delay{n}"].shape[0]):
            g.set_state(f"fdn_delay{n}", i, st[f"fdn_delay{n}"][i])
    eng.set_flush_denormals(flush)
    seen = 0
    try:
        for c in range(chunks):
            x = starved_input(V, 2, chunk, seed=21, amp=0.1) if c == 0 else np.zeros((V, 64 * chunk), np.float32)
            got = g.process_host(chunk, {"x": x}, Layout.QUAD)
            with oracle.flush_denormals(flush):
                want = evaluate_stream(oracle, desc, outs, V, chunk, {"x": x}, params, coeffs, st)
            for i in range(2):
                assert_bits_equal(got[i], want[i], True, f"FDN out {i} chunk {c} flush={flush}")
            seen += denormal_stats(want)
    finally:
        eng.set_flush_denormals(False)
    assert (seen == 0) if flush else (seen > 0)


@pytest.mark.parametrize("flush", MODES)
@pytest.mark.parametrize("kind", [Proc.SAW_GEN, Proc.PULSE_GEN])
def test_oscillators_in_both_modes(eng, oracle, kind, flush):
    """polyBLEP's division (the NR sequence and the IEEE fallback) with tiny, denormal and ordinary frequencies."""
    V, T = 256, 6
    f = np.concatenate([np.linspace(1e-4, 0.45, V - 32), np.float32(2.0) ** -np.arange(100, 132, dtype=np.float32) * 1.7,
                        ]).astype(np.float32)[:V]
    f[-8:] = np.array([1e-39, 3e-42, 1.1754944e-38, 2e-38, 1e-45, 0.0, 5e-39, 1.4e-45], np.float32)
    procs = [kind]
    co = chain_coeffs(oracle, procs, V, seed=3)
    st = oracle.chain_clear(procs, V)
    eng.set_flush_denormals(flush)
    try:
        bank = eng.bank(procs, V)
        bank.set_all_coeffs(co)
        bank.set_all_state(st)
        bank.set_input_const(f)
        got = bank.process_host(T, None, Layout.QUAD)
        gst = bank.get_all_state()
        sig = np.repeat(f[:, None], 64 * T, 1).copy()
        bank2 = eng.bank(procs, V)
        bank2.set_all_coeffs(co)
        bank2.set_all_state(oracle.chain_clear(procs, V))
        got2 = bank2.process_host(T, sig, Layout.QUAD)       # the same frequencies streamed per sample
    finally:
        eng.set_flush_denormals(False)
    with oracle.flush_denormals(flush):
        want = oracle.chain_process(procs, T, co, st, None, f)
    assert_bits_equal(got, want, True, f"proc {kind} const-freq flush={flush}")
    assert_bits_equal(got2, want, True, f"proc {kind} streamed-freq flush={flush}")
    assert_bits_equal(gst, st, False, "state")


def test_mode_is_per_engine_and_reported(eng):
    import madronalib_amd as ml
    other = ml.Engine(0)
    eng.set_flush_denormals(True)
    try:
        assert eng.get_flush_denormals() and not other.get_flush_denormals()
        x = np.full(64, 1e-39, np.float32)
        assert (eng.op(Op.MULTIPLY, x, np.ones(64, np.float32)).view(np.float32) == 0).all()
        assert (other.op(Op.MULTIPLY, x, np.ones(64, np.float32)).view(np.float32) == x).all()
        # moves and selects pass denormals through in both modes (as SSE's blends and moves do)
        m = np.full(64, 0xFFFFFFFF, np.uint32)
        assert (eng.op(Op.SELECT, x, np.zeros(64, np.float32), m).view(np.float32) == x).all()
    finally:
        eng.set_flush_denormals(False)
        other.close()
