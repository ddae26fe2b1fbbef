"""GPU parity tests proper: the HIP path (through the C-ABI) against the CPU oracle on the same
seeded inputs, against the committed golden vectors of the compiled reference, and — at full
BASELINE sizes — through size-independent properties. Bit-exact unless stated.

Tolerances (stated once, used below):
  HW_REL   = 2^-11 relative (SURVEY 8d's figure) for sqrtApprox / divideApprox / Peak / RMS: the reference uses the
             x86 rcpps/rsqrtps 12-bit tables (error <= 1.5 * 2^-12), which no other hardware reproduces;
             the largest difference the session measured is printed at its end (conftest.py).
  everything else: 0 ulp (bit-exact; any-NaN == any-NaN for float results).
"""
import numpy as np
import pytest

from golden_cases import ANCHORS, chain_case, chain_case_names, load_chains, load_ops
from inputs import (hostile_adsr_state, hostile_gate, assert_bits_equal, assert_rel_close, chain_coeffs, chain_input, is_float_result, lcg_noise,
                    op_inputs, ramp_pi)
from madronalib_amd.constants import Layout, Op, Proc, RowOp

pytestmark = pytest.mark.gpu

HW_REL = 2.0 ** -11


@pytest.fixture(scope="module")
def eng():
    import madronalib_amd as ml
    e = ml.Engine(0)
    yield e
    e.close()


def _hw_ok(got, want):
    g, w = got.view(np.float32), want.view(np.float32)
    return np.isfinite(w) & np.isfinite(g) & (w != 0) & (np.abs(w) > 1e-30)


# ---- elementwise ops ---------------------------------------------------------------------

@pytest.mark.parametrize("op", Op.UNARY + Op.BINARY + Op.TERNARY)
def test_op_vs_oracle(eng, oracle, op):
    a, b, c = op_inputs(op, 64 * 256 + 4 * 3 + 3, seed=9)  # odd length: exercises the vector + scalar tails
    want = oracle.op(op, a, b, c)
    got = eng.op(op, a, b, c)
    if op in Op.HW_APPROX:
        ok = _hw_ok(got, want)
        assert_rel_close(got.view(np.float32)[ok], want.view(np.float32)[ok], HW_REL, f"op {op}")
    else:
        assert_bits_equal(got, want, is_float_result(op), f"op {op}")


@pytest.mark.parametrize("op", Op.UNARY + Op.BINARY + Op.TERNARY)
def test_op_vs_golden(eng, op):
    d = load_ops()
    g = lambda k: d[f"op{op}_{k}"] if f"op{op}_{k}" in d.files else None  # noqa: E731
    got = eng.op(op, g("a"), g("b"), g("c"))
    want = d[f"op{op}_out"]
    if op in Op.HW_APPROX:
        ok = _hw_ok(got, want)
        assert_rel_close(got.view(np.float32)[ok], want.view(np.float32)[ok], HW_REL, f"op {op}")
    else:
        assert_bits_equal(got, want, is_float_result(op), f"op {op}")


@pytest.mark.parametrize("op", [Op.ADD, Op.SUBTRACT, Op.MULTIPLY, Op.DIVIDE, Op.POW, Op.POW_APPROX, Op.MIN, Op.MAX])
def test_rows1_vs_oracle(eng, oracle, op):
    a, b, _ = op_inputs(op, 64 * 40, seed=4)
    a = np.abs(a) if op in (Op.POW, Op.POW_APPROX) else a
    assert_bits_equal(eng.op_rows1(op, a, b[:64]), oracle.op_rows1(op, a, b[:64]), True, f"rows1 {op}")


@pytest.mark.parametrize("rowop", [RowOp.SUM, RowOp.MEAN, RowOp.MAX, RowOp.MIN])
def test_row_reduce_vs_oracle(eng, oracle, rowop):
    rng = np.random.default_rng(3)
    rows = rng.standard_normal(64 * 300).astype(np.float32) * np.float32(100)
    rows[64:128] = -np.abs(rows[64:128])
    assert_bits_equal(eng.row_reduce(rowop, rows), oracle.row_reduce(rowop, rows), True, f"rowop {rowop}")
    d = load_ops()
    assert_bits_equal(eng.row_reduce(rowop, d["rows"]), d[f"rowop{rowop}"], True, f"rowop {rowop} golden")


def test_config2_ramp_and_noise(eng, oracle):
    """BASELINE config 2 inputs at a size the oracle finishes in seconds."""
    V = 2048
    for x in (ramp_pi(V), (lcg_noise(np.arange(V, dtype=np.uint32), 64) * np.float32(np.pi)).astype(np.float32)):
        for op in (Op.SIN_APPROX, Op.EXP_APPROX, Op.EXP_APPROX_OF_SIN_APPROX, Op.SIN, Op.EXP):
            assert_bits_equal(eng.op(op, x), oracle.op(op, x), True, f"cfg2 op {op}")


def test_config2_full_size_properties(eng):
    """65 536 voices x 64: fused pair == composition of the two kernels, and idempotent re-run."""
    V = 65536
    x = ramp_pi(V)
    s = eng.op(Op.SIN_APPROX, x)
    e1 = eng.op(Op.EXP_APPROX, s.view(np.float32))
    e2 = eng.op(Op.EXP_APPROX_OF_SIN_APPROX, x)
    assert_bits_equal(e1, e2, True, "fused pair")
    # the ramp repeats every 4096 elements: so must the result
    assert (e2.reshape(-1, 4096) == e2.reshape(-1, 4096)[0]).all()


def test_layout_convert_roundtrip(eng):
    V, T = 100, 5  # ragged V (not a multiple of 64)
    x = np.arange(V * T * 64, dtype=np.float32).reshape(V, T * 64)
    d_vm = eng.to_device(x)
    bufs = {Layout.QUAD: eng.alloc(x.nbytes), Layout.ROWS: eng.alloc(x.nbytes), "back": eng.alloc(x.nbytes)}
    eng.layout_convert(d_vm, Layout.VOICE_MAJOR, bufs[Layout.QUAD], Layout.QUAD, V, T)
    q = bufs[Layout.QUAD].download(np.float32).reshape(T * 16, V, 4)
    want_q = x.reshape(V, T * 16, 4).transpose(1, 0, 2)
    assert (q == want_q).all()
    eng.layout_convert(bufs[Layout.QUAD], Layout.QUAD, bufs[Layout.ROWS], Layout.ROWS, V, T)
    r = bufs[Layout.ROWS].download(np.float32).reshape(T, V, 64)
    assert (r == x.reshape(V, T, 64).transpose(1, 0, 2)).all()
    eng.layout_convert(bufs[Layout.ROWS], Layout.ROWS, bufs["back"], Layout.VOICE_MAJOR, V, T)
    assert (bufs["back"].download(np.float32).reshape(V, T * 64) == x).all()


# ---- processors and chains ----------------------------------------------------------------

def _run_gpu(eng, procs, V, T, coeffs, state0, sig, const, layout=Layout.QUAD, calls=1):
    bank = eng.bank(procs, V)
    bank.set_all_coeffs(coeffs)
    bank.set_all_state(state0)
    if const is not None:
        bank.set_input_const(const)
    outs = [bank.process_host(T, sig, layout) for _ in range(calls)]
    st = bank.get_all_state()
    fused = bank.fused
    bank.close()
    return outs, st, fused


@pytest.mark.parametrize("layout", [Layout.QUAD, Layout.ROWS, Layout.VOICE_MAJOR])
@pytest.mark.parametrize("kind", Proc.ALL)
def test_single_proc_vs_oracle(eng, oracle, kind, layout):
    V, T = 200, 9  # ragged: 200 voices = 3 full waves + 8 lanes
    procs = [kind]
    co = chain_coeffs(oracle, procs, V, seed=5)
    sig, const = chain_input(procs, V, T, seed=kind)
    st = oracle.chain_clear(procs, V)
    if kind == Proc.NOISE_GEN:
        st[0] = np.arange(V, dtype=np.uint32)
    if kind == Proc.ONE_SHOT_GEN:
        st[1] = 1
    (got,), gst, _ = _run_gpu(eng, procs, V, T, co, st, sig, const, layout)
    want = oracle.chain_process(procs, T, co, st, sig, const)
    if kind in Proc.HW_APPROX:
        assert_rel_close(got, want, HW_REL, f"proc {kind}")
    else:
        assert_bits_equal(got, want, True, f"proc {kind} output")
    assert_bits_equal(gst, st, False, f"proc {kind} final state")


CHAINS = {
    "cfg1_sine_lopass": [Proc.SINE_GEN, Proc.LOPASS],
    "cfg3_saw_bandpass_gain": [Proc.SAW_GEN, Proc.BANDPASS, Proc.GAIN],
    "cfg4_lopass8": [Proc.LOPASS] * 8,
    "cfg4_noise_lopass8": [Proc.NOISE_GEN] + [Proc.LOPASS] * 8,
    "pulse_hipass_onepole": [Proc.PULSE_GEN, Proc.HIPASS, Proc.ONE_POLE],
    # not in the fused catalogue -> runs processor by processor through HBM scratch
    "unfused_saw_shelves_bell_dc": [Proc.SAW_GEN, Proc.LO_SHELF, Proc.HI_SHELF, Proc.BELL, Proc.DC_BLOCKER],
    "unfused_noise_adsrgate": [Proc.NOISE_GEN, Proc.ONE_POLE, Proc.INTEGRATOR, Proc.DIFFERENTIATOR],
}


@pytest.mark.parametrize("name", list(CHAINS))
def test_chain_vs_oracle_with_resume(eng, oracle, name):
    procs = CHAINS[name]
    V, T = 300, 70  # T > scratch slice for unfused chains at this V? (slice = 64) -> two slices
    co = chain_coeffs(oracle, procs, V, seed=11)
    sig, const = chain_input(procs, V, T, seed=2)
    st = oracle.chain_clear(procs, V)
    if procs[0] == Proc.NOISE_GEN:
        st[0] = np.arange(V, dtype=np.uint32)
    outs, gst, fused = _run_gpu(eng, procs, V, T, co, st.copy(), sig, const, Layout.QUAD, calls=2)
    assert fused, name  # catalogue kernels ahead of time, everything else fused at run time (hiprtc)
    st_j = st.copy()
    for got in outs:  # the second call resumes from the carried state
        want = oracle.chain_process(procs, T, co, st_j, sig, const, n_threads=4)
        assert_bits_equal(got, want, True, name)
    assert_bits_equal(gst, st_j, False, name + " state")
    if name.startswith("unfused"):
        # the same chain with run-time fusion switched off: processor by processor through HBM scratch
        eng.set_jit(False)
        try:
            outs, gst, fused = _run_gpu(eng, procs, V, T, co, st.copy(), sig, const, Layout.ROWS, calls=2)
        finally:
            eng.set_jit(True)
        assert not fused
        for got in outs:
            want = oracle.chain_process(procs, T, co, st, sig, const, n_threads=4)
            assert_bits_equal(got, want, True, name + " (unfused)")
        assert_bits_equal(gst, st, False, name + " (unfused) state")


CASCADES = {
    "lopass2": [Proc.LOPASS] * 2, "lopass4": [Proc.LOPASS] * 4, "lopass8": [Proc.LOPASS] * 8,
    "hipass2": [Proc.HIPASS] * 2, "hipass4": [Proc.HIPASS] * 4,
    "bandpass2": [Proc.BANDPASS] * 2, "bandpass4": [Proc.BANDPASS] * 4,
    "noise_lopass8": [Proc.NOISE_GEN] + [Proc.LOPASS] * 8, "saw_lopass4": [Proc.SAW_GEN] + [Proc.LOPASS] * 4,
}


@pytest.mark.parametrize("layout", [Layout.QUAD, Layout.ROWS, Layout.VOICE_MAJOR])
@pytest.mark.parametrize("T", [1, 2, 9])
@pytest.mark.parametrize("name", list(CASCADES))
def test_skewed_cascade_vs_oracle(eng, oracle, name, T, layout):
    """The stage-skewed cascade kernel fills and drains its pipeline inside every launch: outputs and
    state must equal the sample-at-a-time oracle for any launch length, layout and across launches."""
    procs = CASCADES[name]
    V = 150
    co = chain_coeffs(oracle, procs, V, seed=21)
    sig, const = chain_input(procs, V, T, seed=3)
    st = oracle.chain_clear(procs, V)
    if procs[0] == Proc.NOISE_GEN:
        st[0] = np.arange(V, dtype=np.uint32) + 7
    outs, gst, fused = _run_gpu(eng, procs, V, T, co, st.copy(), sig, const, layout, calls=3)
    assert fused
    for got in outs:
        want = oracle.chain_process(procs, T, co, st, sig, const, n_threads=4)
        assert_bits_equal(got, want, True, f"{name} T={T}")
    assert_bits_equal(gst, st, False, name + " state")


HEADLESS_CASCADES = [n for n, procs in CASCADES.items() if len(set(procs)) == 1]


@pytest.mark.parametrize("T", [1, 2, 5])
@pytest.mark.parametrize("lanes", [-1, 1, 2, 4])
@pytest.mark.parametrize("name", HEADLESS_CASCADES)
def test_cascade_forms_agree(eng, oracle, name, lanes, T):
    """Every form of an SVF cascade (mlgpu_engine_set_cascade_lanes: the round-2 kernel, 1, 2 or 4 wavefront lanes per
    channel — the launcher falls back to the widest form the number of sections allows) equals the oracle bit for bit: same
    operations per section, only the lane that runs them differs. Ragged V (partial lane groups and wavefronts), three
    layouts' worth of strides via ROWS in / QUAD out, state carried over three launches, and a launch-constant input."""
    procs = CASCADES[name]
    V = 150
    co = chain_coeffs(oracle, procs, V, seed=31)
    sig, _ = chain_input(procs, V, T, seed=5)
    eng.set_cascade_lanes(lanes)
    try:
        assert eng.get_cascade_lanes() == lanes
        for layout, use_const in ((Layout.QUAD, False), (Layout.ROWS, False), (Layout.VOICE_MAJOR, True)):
            st = oracle.chain_clear(procs, V)
            const = (np.linspace(-1.0, 1.0, V).astype(np.float32)) if use_const else None
            outs, gst, fused = _run_gpu(eng, procs, V, T, co, st.copy(), None if use_const else sig, const, layout, calls=3)
            assert fused
            for got in outs:
                want = oracle.chain_process(procs, T, co, st, None if use_const else sig, const, n_threads=4)
                assert_bits_equal(got, want, True, f"{name} lanes={lanes} T={T} layout={layout}")
            assert_bits_equal(gst, st, False, f"{name} lanes={lanes} state")
    finally:
        eng.set_cascade_lanes(0)


def test_cascade_kernel_follows_bank_size(eng):
    """The form is chosen from the bank's size unless forced, and the bank reports the kernel a profiler will show."""
    names = {}
    for V in (4096, 32768, 131072):
        bank = eng.bank([Proc.LOPASS] * 8, V)
        names[V] = bank.kernel_name
        bank.close()
    assert names[4096].startswith("cascade_lanes_kernel<16, 8, 4")
    assert names[32768].startswith("cascade_lanes_kernel<16, 8, 2")
    assert names[131072].startswith("cascade_lanes_kernel<16, 8, 1")
    eng.set_cascade_lanes(-1)
    try:
        bank = eng.bank([Proc.LOPASS] * 8, 4096)
        assert bank.kernel_name.startswith("cascade_kernel<mldev::Chain<>, 16, 8")
        bank.close()
    finally:
        eng.set_cascade_lanes(0)
    bank = eng.bank([Proc.HIPASS] * 2, 4096)  # two sections: one lane per channel is all there is
    assert bank.kernel_name.startswith("cascade_lanes_kernel<17, 2, 1") or bank.kernel_name.startswith("cascade_lanes_kernel<")
    bank.close()


@pytest.mark.parametrize("name", chain_case_names())
def test_chain_vs_golden(eng, name):
    c = chain_case(load_chains(), name)
    V = c["state0"].shape[1] if c["state0"].size else c["coeffs"].shape[1]
    T = c["out1"].shape[1] // 64
    outs, gst, _ = _run_gpu(eng, c["procs"], V, T, c["coeffs"], c["state0"], c["in_signal"], c["in_const"], Layout.ROWS, calls=2)
    hw = any(p in Proc.HW_APPROX for p in c["procs"])
    for got, k in zip(outs, ("out1", "out2")):
        if hw:
            assert_rel_close(got, c[k], HW_REL, name)
        else:
            assert_bits_equal(got, c[k], True, f"{name} {k}")
    assert_bits_equal(gst, c["state2"], False, name + " state2")


def test_anchors_cfg1_cfg3(eng):
    """SURVEY Appendix B anchors through the HIP path (1 voice, 1 DSPVector = BASELINE config 1)."""
    import madronalib_amd as ml
    a = ANCHORS["cfg1"]
    bank = eng.bank(a["procs"], 1)
    bank.clear()
    bank.set_coeffs(1, ml.Lopass.makeCoeffs(*a["lopass"]))
    bank.set_input_const(np.array([a["freq"]], np.float32))
    y = bank.process_host(1)[0]
    assert list(y[:4]) == a["y0_3"] and y[63] == a["y63"]
    a = ANCHORS["cfg3"]
    bank = eng.bank(a["procs"], 1)
    bank.clear()
    bank.set_coeffs(1, ml.Bandpass.makeCoeffs(*a["bandpass"]))
    bank.set_coeffs(2, [a["gain"]])
    bank.set_input_const(np.array([a["freq"]], np.float32))
    z = bank.process_host(1)[0]
    assert list(z[:4]) == a["z0_3"] and z[63] == a["z63"]


@pytest.mark.parametrize("kind", [Proc.SAW_GEN, Proc.PULSE_GEN])
@pytest.mark.parametrize("streamed", [False, True])
def test_blep_division_domain(eng, oracle, kind, streamed):
    """The polyBLEP division is evaluated with a hoisted Newton-Raphson reciprocal on the GPU; it must
    equal the reference's IEEE division for every frequency, including absurd ones (0, negative, tiny,
    huge, inf, NaN — those take the full IEEE path per wavefront)."""
    V, T = 16384, 6
    rng = np.random.default_rng(77)
    f = np.exp(rng.uniform(np.log(1e-7), np.log(0.49), V)).astype(np.float32)
    odd = np.array([0.0, -0.0, -0.01, 1e-25, 1e-38, 1e-45, 1e30, np.inf, -np.inf, np.nan, 0.5, 0.7, 1.5, 2.0 ** -64,
                    2.0 ** -65, 2.0 ** 64, 1.0, 0.25, 1.0 / 3.0, 0.49999997], np.float32)
    f[1000:1000 + odd.size] = odd          # inside otherwise ordinary wavefronts
    f[5000:5064] = np.float32(1e-30)       # a whole odd wavefront
    procs = [kind]
    co = chain_coeffs(oracle, procs, V, seed=1)
    st = oracle.chain_clear(procs, V)
    st[0] = rng.integers(0, 2 ** 32, V, dtype=np.uint64).astype(np.uint32)  # random start phases
    sig = np.repeat(f[:, None], 64 * T, 1) if streamed else None
    const = None if streamed else f
    (got,), gst, _ = _run_gpu(eng, procs, V, T, co, st.copy(), sig, const, Layout.QUAD)
    want = oracle.chain_process(procs, T, co, st, sig, const, n_threads=8)
    assert_bits_equal(got, want, True, f"blep kind {kind} streamed {streamed}")
    assert_bits_equal(gst, st, False, "state")


def test_default_state_and_clear(eng, oracle):
    for procs in ([Proc.SINE_GEN, Proc.LOPASS], [Proc.ADSR], [Proc.SAW_GEN, Proc.BANDPASS, Proc.GAIN]):
        bank = eng.bank(procs, 70)
        assert_bits_equal(bank.get_all_state(), oracle.chain_default_state(procs, 70), False, "default state")
        bank.clear()
        assert_bits_equal(bank.get_all_state(), oracle.chain_clear(procs, 70), False, "clear state")


# ---- full BASELINE sizes: size-independent properties --------------------------------------

def _cfg3_setup(eng, V):
    import madronalib_amd as ml
    v = np.arange(V, dtype=np.float64)
    freq = (55.0 * 2.0 ** (5.0 * v / V) / 48000.0).astype(np.float32)
    # coefficients depend on omega only through a smooth map: make them for 1024 distinct omegas
    om = np.minimum(0.45, 4.0 * freq.astype(np.float64)).astype(np.float32)
    uniq, inv = np.unique(om, return_inverse=True)
    table = np.stack([ml.Bandpass.makeCoeffs(float(o), 0.5) for o in uniq])
    co = table[inv]  # [V][3]
    bank = eng.bank([Proc.SAW_GEN, Proc.BANDPASS, Proc.GAIN], V)
    bank.clear()
    for i in range(3):
        bank.set_coeff(1, i, np.ascontiguousarray(co[:, i]))
    bank.set_coeff(2, 0, 0.25)
    bank.set_input_const(freq)
    return bank, freq, co


def test_config3_full_size(eng, oracle):
    """262 144 voices: (a) a strided subset of voices equals the oracle bit-for-bit, (b) T vectors in
    one launch == the same T vectors in two launches (state carry), (c) QUAD and ROWS outputs agree."""
    V, T = 262144, 4
    bank, freq, co = _cfg3_setup(eng, V)
    n = V * T * 64
    d_q = eng.alloc(4 * n)
    bank.process(T, d_q, Layout.QUAD)
    q = d_q.download(np.float32).reshape(T * 16, V, 4)
    st_one = bank.get_all_state()

    sub = np.arange(0, V, 509)
    coeffs = np.ascontiguousarray(np.concatenate([co[sub].T, np.full((1, sub.size), 0.25, np.float32)], 0))
    st = oracle.chain_clear([Proc.SAW_GEN, Proc.BANDPASS, Proc.GAIN], sub.size)
    want = oracle.chain_process([Proc.SAW_GEN, Proc.BANDPASS, Proc.GAIN], T, coeffs, st, None, freq[sub], n_threads=4)
    got = q[:, sub, :].transpose(1, 0, 2).reshape(sub.size, T * 64)
    assert_bits_equal(got, want, True, "cfg3 subset")
    assert_bits_equal(st_one[:, sub], st, False, "cfg3 subset state")

    bank.clear()
    d_r = eng.alloc(4 * n)
    half = T // 2
    bank.process(half, d_r, Layout.ROWS)
    d_r2 = eng.alloc(4 * n // 2)
    bank.process(T - half, d_r2, Layout.ROWS)
    r = np.concatenate([d_r.download(np.float32, n // 2), d_r2.download(np.float32)]).reshape(T, V, 64)
    assert (r.view(np.uint32) == q.reshape(T, 16, V, 4).transpose(0, 2, 1, 3).reshape(T, V, 64).view(np.uint32)).all()
    assert_bits_equal(bank.get_all_state(), st_one, False, "split launches state")


def test_config4_full_size(eng, oracle):
    """131 072 channels x 8 Lopass sections over streamed noise: subset vs oracle + linearity-free
    property: processing [x] then [x'] equals processing the concatenation."""
    import madronalib_amd as ml
    V, T = 131072, 2
    procs = [Proc.LOPASS] * 8
    bank = eng.bank(procs, V)
    cs = [ml.Lopass.makeCoeffs(float(np.float32(0.02) * np.float32(i + 1)), 0.7) for i in range(8)]
    for i in range(8):
        bank.set_coeffs(i, cs[i])
    # input: NoiseGen bank (seed = channel) on the GPU itself
    nb = eng.bank([Proc.NOISE_GEN], V)
    nb.set_state(0, 0, np.arange(V, dtype=np.uint32))
    d_x = eng.alloc(4 * V * T * 64)
    nb.process(T, d_x, Layout.QUAD)
    d_y = eng.alloc(4 * V * T * 64)
    bank.process(T, d_y, Layout.QUAD, d_x, Layout.QUAD)
    y = d_y.download(np.float32).reshape(T * 16, V, 4)
    sub = np.arange(0, V, 257)
    co = np.ascontiguousarray(np.repeat(np.concatenate(cs)[:, None], sub.size, 1))
    x_sub = lcg_noise(sub.astype(np.uint32), T * 64)
    st = oracle.chain_clear(procs, sub.size)
    want = oracle.chain_process(procs, T, co, st, x_sub, None, n_threads=4)
    got = y[:, sub, :].transpose(1, 0, 2).reshape(sub.size, T * 64)
    assert_bits_equal(got, want, True, "cfg4 subset")
    a = ANCHORS["cfg4_vec4"]
    # Appendix-B anchor: channel 0, 4th vector needs T >= 4: run two more vectors
    nb.process(T, d_x, Layout.QUAD)
    bank.process(T, d_y, Layout.QUAD, d_x, Layout.QUAD)
    y2 = d_y.download(np.float32).reshape(T * 16, V, 4)[16:, 0, :].reshape(64).view(np.uint32)
    assert (y2[0], y2[31], y2[63]) == (a["y0"], a["y31"], a["y63"])


# ---- BASELINE lengths and widths, bit-compared (round 3) ------------------------------------------------------------

def test_config3_baseline_length(eng, oracle):
    """BASELINE configs[2] exactly as the bench runs it: 262 144 voices x 750 DSPVectors (one second of audio at 48 kHz) in
    25 launches of 30. A strided subset of 515 voices is compared with the oracle on the outputs of the first, a middle and
    the last launch and on the final state: 48 000 samples of carried phase and filter memory."""
    V, T, launches = 262144, 30, 25
    procs = [Proc.SAW_GEN, Proc.BANDPASS, Proc.GAIN]
    bank, freq, co = _cfg3_setup(eng, V)
    sub = np.arange(0, V, 509)
    assert sub.size >= 512
    coeffs = np.ascontiguousarray(np.concatenate([co[sub].T, np.full((1, sub.size), 0.25, np.float32)], 0))
    st = oracle.chain_clear(procs, sub.size)
    want = oracle.chain_process(procs, T * launches, coeffs, st, None, freq[sub], n_threads=8).reshape(sub.size, launches, T * 64)
    # (every voice on the first launch and on the final state: test_config3_every_voice below - a test of its own, so that a run
    # without the compiled reference SKIPS it and says so instead of passing with less checked)
    d_q = eng.alloc(4 * V * T * 64)
    for k in range(launches):
        bank.process(T, d_q, Layout.QUAD)
        if k in (0, launches // 2, launches - 1):
            q = d_q.download(np.float32).reshape(T * 16, V, 4)
            got = q[:, sub, :].transpose(1, 0, 2).reshape(sub.size, T * 64)
            assert_bits_equal(got, want[:, k], True, f"cfg3 launch {k}")
    state = bank.get_all_state()
    assert_bits_equal(state[:, sub], st, False, "cfg3 state after 750 vectors")
    bank.close()


def _need_reference():
    """The every-voice comparisons run the CPU side on the reference compiled by oracle/Makefile (several times the plain-C port's
    speed). Where it was not built (no /root/reference, oracle/_ref/ not shipped) they are SKIPPED with this reason - loudly - instead
    of shrinking to a subset and passing."""
    from cpu_checkers import fast_checker
    fast = fast_checker()
    if fast is None:
        pytest.skip("every-voice comparison needs oracle/_ref/libmlref.so (the reference compiled by oracle/Makefile); only the strided-subset tests ran")
    return fast


def test_config3_every_voice(eng, oracle):
    """BASELINE configs[2], all 262 144 voices: the first launch's output and the state of every voice after the bench's whole step
    (750 DSPVectors, 25 launches of 30), against the compiled reference run on the host threads."""
    from cpu_checkers import host_threads
    fast = _need_reference()
    V, T, launches = 262144, 30, 25
    procs = [Proc.SAW_GEN, Proc.BANDPASS, Proc.GAIN]
    bank, freq, co = _cfg3_setup(eng, V)
    all_co = np.ascontiguousarray(np.concatenate([co.T, np.full((1, V), 0.25, np.float32)], 0))
    all_st = oracle.chain_clear(procs, V)
    d_q = eng.alloc(4 * V * T * 64)
    for k in range(launches):
        bank.process(T, d_q, Layout.QUAD)
        if k == 0:
            q = d_q.download(np.float32).reshape(T * 16, V, 4)
            want_all = fast.chain_process(procs, T, all_co, all_st, None, freq, n_threads=host_threads())
            for a in range(0, V, 32768):      # (in slices: the transposed copy of everything at once is 2 GiB)
                got_all = q[:, a:a + 32768, :].transpose(1, 0, 2).reshape(-1, T * 64)
                assert_bits_equal(got_all, want_all[a:a + 32768], True, f"cfg3 launch 0, voices {a}..")
            del want_all, q
    fast.chain_process(procs, T * (launches - 1), all_co, all_st, None, freq, n_threads=host_threads(), want_out=False)
    assert_bits_equal(bank.get_all_state(), all_st, False, "cfg3 state of all 262 144 voices after 750 vectors")
    bank.close()


def test_config4_baseline_length(eng, oracle):
    """BASELINE configs[3] at SURVEY 8d's length: 131 072 channels x 8 Lopass x 4 096 DSPVectors in 128 launches of 32 (the
    bench's launch shape), streamed noise in. 257 channels against the oracle: first, middle and last launch, final state."""
    import madronalib_amd as ml
    V, T, launches = 131072, 32, 128
    procs = [Proc.LOPASS] * 8
    bank = eng.bank(procs, V)
    assert bank.kernel_name.startswith("cascade_lanes_kernel<16, 8, 1")
    cs = [ml.Lopass.makeCoeffs(float(np.float32(0.02) * np.float32(i + 1)), 0.7) for i in range(8)]
    for i in range(8):
        bank.set_coeffs(i, cs[i])
    nb = eng.bank([Proc.NOISE_GEN], V)
    nb.set_state(0, 0, np.arange(V, dtype=np.uint32))
    sub = np.arange(0, V, 511)
    co = np.ascontiguousarray(np.repeat(np.concatenate(cs)[:, None], sub.size, 1))
    x_sub = lcg_noise(sub.astype(np.uint32), T * launches * 64)
    st = oracle.chain_clear(procs, sub.size)
    want = oracle.chain_process(procs, T * launches, co, st, x_sub, None, n_threads=8).reshape(sub.size, launches, T * 64)
    # (every channel, first and last launch of a 16-launch run: test_config4_every_channel below)
    d_x = eng.alloc(4 * V * T * 64)
    d_y = eng.alloc(4 * V * T * 64)
    for k in range(launches):
        nb.process(T, d_x, Layout.QUAD)
        bank.process(T, d_y, Layout.QUAD, d_x, Layout.QUAD)
        if k in (0, launches // 2, launches - 1):
            y = d_y.download(np.float32).reshape(T * 16, V, 4)
            got = y[:, sub, :].transpose(1, 0, 2).reshape(sub.size, T * 64)
            assert_bits_equal(got, want[:, k], True, f"cfg4 launch {k}")
    assert_bits_equal(bank.get_all_state()[:, sub], st, False, "cfg4 state after 4096 vectors")
    bank.close()
    nb.close()


def test_config4_every_channel(eng, oracle):
    """BASELINE configs[3], all 131 072 channels x 8 Lopass over a 16-launch run (512 DSPVectors, the bench's step): the output of
    the FIRST and of the LAST launch and the state every channel is left in, against the compiled reference on the host threads -
    the last launch inherits 15 launches of carried filter memory per stage."""
    import madronalib_amd as ml
    from cpu_checkers import host_threads
    fast = _need_reference()
    V, T, launches = 131072, 32, 16
    procs = [Proc.LOPASS] * 8
    bank = eng.bank(procs, V)
    cs = [ml.Lopass.makeCoeffs(float(np.float32(0.02) * np.float32(i + 1)), 0.7) for i in range(8)]
    for i in range(8):
        bank.set_coeffs(i, cs[i])
    nb = eng.bank([Proc.NOISE_GEN], V)
    nb.set_state(0, 0, np.arange(V, dtype=np.uint32))
    d_x, d_y = eng.alloc(4 * V * T * 64), eng.alloc(4 * V * T * 64)
    kept, xs = {}, []
    for k in range(launches):
        nb.process(T, d_x, Layout.QUAD)
        bank.process(T, d_y, Layout.QUAD, d_x, Layout.QUAD)
        xs.append(d_x.download(np.float32).reshape(T * 16, V, 4))   # the NoiseGen bank's stream (itself bit-exact against the oracle: test_single_proc_vs_oracle)
        if k in (0, launches - 1):
            kept[k] = d_y.download(np.float32).reshape(T * 16, V, 4)
    # one channel's stream in numpy, as a cross-check that the downloaded input is the LCG's
    assert_bits_equal(np.concatenate([x[:, 77, :].reshape(-1) for x in xs]), lcg_noise(np.array([77], np.uint32), launches * T * 64)[0], True, "cfg4 input stream")
    state = bank.get_all_state()
    all_co = np.ascontiguousarray(np.repeat(np.concatenate(cs)[:, None], V, 1))
    slab = 8192      # (the reference's output for a slab: slab x 512 DSPVectors x 256 B = 1 GiB)
    for a in range(0, V, slab):
        x_all = np.ascontiguousarray(np.concatenate([x[:, a:a + slab, :].transpose(1, 0, 2).reshape(slab, T * 64) for x in xs], 1))
        st_a = oracle.chain_clear(procs, slab)
        want = fast.chain_process(procs, launches * T, np.ascontiguousarray(all_co[:, a:a + slab]), st_a, x_all, None, n_threads=host_threads())
        want = want.reshape(slab, launches, T * 64)
        for k, y in kept.items():
            got = y[:, a:a + slab, :].transpose(1, 0, 2).reshape(-1, T * 64)
            assert_bits_equal(got, want[:, k], True, f"cfg4 launch {k}, channels {a}..")
        assert_bits_equal(state[:, a:a + slab], st_a, False, f"cfg4 state after {launches} launches, channels {a}..")
    bank.close()
    nb.close()


def test_config2_baseline_width(eng, oracle):
    """BASELINE configs[1] at its stated width: all 65 536 x 64 elements of sinApprox, expApprox and the fused pair against
    the oracle, on the ramp and on noise in [-pi, pi] (Tests/dspOpsTest.cpp:85-105 is the reference's own precision check)."""
    V = 65536
    noise = (lcg_noise(np.arange(V, dtype=np.uint32), 64) * np.float32(np.pi)).astype(np.float32)
    for label, x in (("ramp", ramp_pi(V)), ("noise", noise)):
        for op in (Op.SIN_APPROX, Op.EXP_APPROX, Op.EXP_APPROX_OF_SIN_APPROX):
            assert_bits_equal(eng.op(op, x), oracle.op(op, x), True, f"cfg2 {label} op {op}")


def test_empty_and_error_paths(eng):
    import madronalib_amd as ml
    bank = eng.bank([Proc.LOPASS], 64)
    out = eng.alloc(64 * 64 * 4)
    bank.process(0, out)  # zero vectors: no-op
    with pytest.raises(ml.MlgpuError) as ei:
        bank.set_coeff(0, 7, 1.0)
    assert ei.value.status == ml.Status.ERR_RANGE
    with pytest.raises(ml.MlgpuError):
        eng.bank([999], 64)
    with pytest.raises(ml.MlgpuError):
        eng.bank([Proc.LOPASS], 0)
    assert eng.op(Op.ADD, np.zeros(0, np.float32), np.zeros(0, np.float32)).size == 0


def test_default_constructed_peak_holds_for_44100_samples(eng, oracle):
    """A bank installs the coefficients of a default-constructed reference object: Peak's third coefficient is
    peakHoldSamples{44100} (MLDSPFilters.h:574). A host that only sets makeCoeffs' a0 / b1 must get that hold time."""
    V, T = 70, 8
    procs = [Proc.PEAK]
    bank = eng.bank(procs, V)
    a0b1 = oracle.make_coeffs("onepole", 0.01)          # Peak::makeCoeffs = the one-pole pair (:576-580)
    bank.set_coeff(0, 0, float(a0b1[0]))
    bank.set_coeff(0, 1, float(a0b1[1]))
    assert (bank.get_coeff(0, 2).view(np.int32) == 44100).all()
    co = np.zeros((3, V), np.float32)
    co[0], co[1] = a0b1[0], a0b1[1]
    co[2] = np.array([44100], np.int32).view(np.float32)[0]
    st = oracle.chain_clear(procs, V)
    x = np.zeros((V, 64 * T), np.float32)
    x[:, :64] = lcg_noise(np.arange(V, dtype=np.uint32) + 1, 64)       # one burst, then the hold
    got = bank.process_host(T, x, Layout.QUAD)
    want = oracle.chain_process(procs, T, co, st, x, None)
    assert_rel_close(got, want, HW_REL, "default Peak")
    assert_bits_equal(bank.get_all_state(), st, False, "default Peak state")
    assert (np.abs(got[:, -64:]) > 0).any()                              # still holding at the end (a zero hold would have decayed)
    bank.close()


@pytest.mark.parametrize("flush", [False, True])
@pytest.mark.parametrize("garbage_state", [False, True])
def test_adsr_hostile_gates_and_states(eng, oracle, flush, garbage_state):
    """The ADSR's quiet path decides from carried lane masks whether any lane may change segment (mldsp_procs.hpp); this drives
    it with every gate value its tests could mishandle, from states the envelope itself never produces (an off segment with a
    nonzero y, k or target; segments past off; NaN thresholds), in both floating-point modes, over several launches."""
    V, T, calls = 200, 12, 3
    procs = [Proc.ADSR]
    rng = np.random.default_rng(77 + garbage_state)
    co = chain_coeffs(oracle, procs, V, seed=3)
    st = hostile_adsr_state(oracle.chain_clear(procs, V), rng) if garbage_state else oracle.chain_clear(procs, V)
    eng.set_flush_denormals(flush)
    try:
        bank = eng.bank(procs, V)
        bank.set_all_coeffs(co)
        bank.set_all_state(st)
        for c in range(calls):
            gate = hostile_gate(V, 64 * T, seed=100 * c + int(garbage_state))
            got = bank.process_host(T, gate, Layout.QUAD)
            with oracle.flush_denormals(flush):
                want = oracle.chain_process(procs, T, co, st, gate, None, n_threads=8)
            assert_bits_equal(got, want, True, f"hostile ADSR call {c} flush={flush} garbage={garbage_state}")
            assert_bits_equal(bank.get_all_state(), st, False, f"hostile ADSR state call {c}")
        bank.close()
    finally:
        eng.set_flush_denormals(False)


def _hostile_signal(V, S, seed):
    """[V][S]: ordinary noise with stretches of special values (inf, NaN, denormals, huge, -0) and raw bit patterns mixed in."""
    from inputs import general_floats
    x = lcg_noise(np.arange(V, dtype=np.uint32) + np.uint32(seed), S)
    g = general_floats(V * S, seed).reshape(V, S)
    rng = np.random.default_rng(seed)
    mask = rng.random((V, S)) < 0.05
    x[mask] = g[mask]
    x[1::7, S // 3:S // 3 + 40] = g[1::7, :40]      # whole stretches too, so that a recurrence has to live with them
    return np.ascontiguousarray(x)


SVF_KINDS = (Proc.LOPASS, Proc.HIPASS, Proc.BANDPASS, Proc.LO_SHELF, Proc.HI_SHELF, Proc.BELL)


def _below_the_svf_overflow_regime(x):
    """The state-variable filters' `ic += 2 t` is one fused instruction on the device: the same float unless 2 t overflows while
    the sum does not, |t| > 1.7e38 (include/mlgpu.h, numerical contract: a filter that has blown up may reach inf / NaN a few
    samples apart). Finite inputs are kept below 1e15 for them; infinities and NaNs stay."""
    big = np.isfinite(x) & (np.abs(x) > 1e15)
    y = x.copy()
    y[big] = np.float32(1e15) * np.sign(x[big])
    return y


@pytest.mark.parametrize("procs", [[k] for k in SVF_KINDS] + [[Proc.LOPASS] * 8, [Proc.HIPASS] * 4, [Proc.SAW_GEN, Proc.BANDPASS, Proc.GAIN]],
                         ids=lambda p: "x".join(str(int(k)) for k in p))
def test_strict_svf_hostile_input(oracle, procs):
    """mlgpu_engine_set_strict_svf: with `ic + 2 t` spent as two instructions the state-variable filters follow the
    reference through the overflow corner as well - the hostile signal goes in UNCLAMPED (1e38, FLT_MAX, raw bit patterns next
    to infinities and NaNs), outputs and memories bit for bit over two launches. Banks of a strict engine are generated
    kernels (hiprtc), the plain cascades in their stage-skewed form; the default engine beside it keeps its own kernels."""
    import madronalib_amd as ml
    e = ml.Engine(0)
    try:
        e.set_strict_svf(True)
        assert e.get_strict_svf()
        V, T = 200, 6
        co = chain_coeffs(oracle, procs, V, seed=5)
        has_input = procs[0] not in Proc.GENERATORS
        sig = _hostile_signal(V, 64 * T * 2, seed=7 + len(procs)) if has_input else None
        _, const = chain_input(procs, V, T, seed=3)
        st = oracle.chain_clear(procs, V)
        bank = e.bank(procs, V)
        assert bank.fused and "hiprtc" in bank.kernel_name
        bank.set_all_coeffs(co)
        bank.set_all_state(st)
        if const is not None:
            bank.set_input_const(const)
        for call in range(2):
            part = np.ascontiguousarray(sig[:, call * 64 * T:(call + 1) * 64 * T]) if has_input else None
            got = bank.process_host(T, part, Layout.QUAD)
            want = oracle.chain_process(procs, T, co, st, part, const, n_threads=8)
            assert_bits_equal(got, want, True, f"strict svf {procs} call {call}")
            g32, w32 = bank.get_all_state().view(np.uint32), st.view(np.uint32)
            bothnan = np.isnan(g32.view(np.float32)) & np.isnan(w32.view(np.float32))
            assert ((g32 == w32) | bothnan).all(), f"strict svf {procs} state call {call}"
        bank.close()
    finally:
        e.close()


@pytest.mark.parametrize("kind", [k for k in Proc.ALL if k not in Proc.HW_APPROX and k != Proc.NOISE_GEN])
def test_single_proc_hostile_input(eng, oracle, kind):
    """Every processor with inputs nobody would send on purpose: infinities, NaNs, denormals, 1e38, raw bit patterns - as the
    signal of a filter, the gate of an envelope, the frequency of an oscillator. The reference computes *something* for each of
    them; so must the device, bit for bit (any NaN equals any NaN), output and final state, over two launches."""
    V, T = 200, 6
    procs = [kind]
    co = chain_coeffs(oracle, procs, V, seed=5)
    sig = _hostile_signal(V, 64 * T * 2, seed=int(kind) + 3)
    if kind in SVF_KINDS:
        sig = _below_the_svf_overflow_regime(sig)
    st = oracle.chain_clear(procs, V)
    if kind == Proc.ONE_SHOT_GEN:
        st[1] = 1
    bank = eng.bank(procs, V)
    bank.set_all_coeffs(co)
    bank.set_all_state(st)
    for call in range(2):
        part = np.ascontiguousarray(sig[:, call * 64 * T:(call + 1) * 64 * T])
        got = bank.process_host(T, part, Layout.QUAD)
        want = oracle.chain_process(procs, T, co, st, part, None, n_threads=8)
        assert_bits_equal(got, want, True, f"hostile proc {kind} call {call}")
        gst = bank.get_all_state()
        # state words are floats or integers; a float NaN may differ in payload between the two machines
        g32, w32 = gst.view(np.uint32), st.view(np.uint32)
        bothnan = np.isnan(g32.view(np.float32)) & np.isnan(w32.view(np.float32))
        assert ((g32 == w32) | bothnan).all(), f"hostile proc {kind} state call {call}"
    bank.close()
