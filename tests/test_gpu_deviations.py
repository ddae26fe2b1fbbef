"""The three places where the DEFAULT build does not hand back the reference's bits, each pinned on its exact trigger (INTEGRATION.md,
"Behavioural differences"): (a) the SVF accumulators as one fused operation - differs only when 2 t overflows while ic + 2 t does not;
(b) the libm sinf restated without FMA - differs from an FMA-built glibc in the last bit on twelve arguments, all with 53 < |x| < 120;
(c) the hardware-approximate reciprocal and reciprocal square root - another table than x86's rcpps / rsqrtps, inside its error band.
Everything else is bit-exact; these tests fail if a deviation grows, moves, or quietly disappears."""
import numpy as np
import pytest

from inputs import assert_bits_equal
from madronalib_amd.constants import Layout, Op, Proc

pytestmark = pytest.mark.gpu

# the twelve arguments on which glibc 2.35's x86-64 FMA variant of sinf and the plain-double algorithm disagree (last bit)
SINF_FMA_DIFFERS = [0x4255b0a9, 0x42a35c07, 0x42a35d44, 0x42a97360, 0x42cf5854, 0x42e87a55,
                    0xc255b0a9, 0xc2a35c07, 0xc2a35d44, 0xc2a97360, 0xc2cf5854, 0xc2e87a55]


@pytest.fixture(scope="module")
def eng():
    import madronalib_amd as ml
    e = ml.Engine(0)
    yield e
    e.close()


def _flipping_lopass(engine, oracle, ic1):
    """A Lopass with g0 = g2 = 0, g1 = -1 and memory (ic1, 0), silence in: t1 = -ic1eq, t2 = 0, so every sample does
    ic1eq += 2 * (-ic1eq) - the memory changes sign and keeps its size, as long as 2 * ic1eq can be formed."""
    V = 64
    procs = [Proc.LOPASS]
    co = np.zeros((3, V), np.float32)
    co[1] = -1.0
    st = oracle.chain_clear(procs, V)
    st[0] = np.float32(ic1).view(np.uint32)
    sig = np.zeros((V, 64), np.float32)
    bank = engine.bank(procs, V)
    bank.set_all_coeffs(co)
    bank.set_all_state(st)
    got = bank.process_host(1, sig, Layout.QUAD)
    gst = bank.get_all_state().view(np.float32)
    bank.close()
    want_st = st.copy()
    want = oracle.chain_process(procs, 1, co, want_st, sig, None)
    return got, gst, want, want_st.view(np.float32)


def test_a_svf_accumulator_is_one_fused_operation(eng, oracle):
    """`ic1eq += 2 * t1` (MLDSPFilters.h:128-131) is fma(2, t1, ic1eq) in the default kernels: the same float whenever 2 * t1 is
    finite (doubling is exact), and a different one only when 2 * t1 overflows while the sum does not. Trigger: ic1eq = 2e38,
    t1 = -2e38 -> the reference forms 2 t1 = -inf, ic1eq = -inf, and a sample later inf - inf = NaN; the fused form has
    2e38 - 4e38 = -2e38 and goes on flipping (output 0 where the reference now has NaN). At 1.7e38 (2 t1 = -3.4e38 is still finite) the two agree for all 64 samples;
    mlgpu_engine_set_strict_svf spends the two instructions and follows the reference into the NaN."""
    import madronalib_amd as ml
    got, gst, want, wst = _flipping_lopass(eng, oracle, 2e38)
    assert np.isnan(wst[0, 0])                                            # the reference
    assert gst[0, 0] == np.float32(2e38)                                  # the default build: 64 flips later, the same size
    # the sample of the overflow itself comes out alike (t2 + ic2eq = 0); from the next one on the reference's 0 * inf makes NaN, the
    # default build's memory is still finite and its output 0
    assert_bits_equal(got[:, :1], want[:, :1], True, "sample 0")
    assert np.isnan(want[:, 1:]).all() and (got[:, 1:] == 0).all()
    got, gst, want, wst = _flipping_lopass(eng, oracle, 1.7e38)           # just below the trigger: bit for bit
    assert_bits_equal(gst, wst, True, "2 t1 finite: same memory")
    assert gst[0, 0] == np.float32(1.7e38)
    strict = ml.Engine(0)
    try:
        strict.set_strict_svf(True)
        got, gst, want, wst = _flipping_lopass(strict, oracle, 2e38)
        assert np.isnan(gst[0, 0])
    finally:
        strict.close()


def test_b_sinf_without_fma_on_its_twelve_arguments(eng, oracle):
    """The per-sample sinf of Lopass(x, omega, k) and TestSineGen is glibc's algorithm in plain double operations. An x86-64 glibc
    built with FMA (the ifunc variant this image's hosts select) rounds twelve arguments differently, all with 53 < |x| < 120 -
    reachable only by a negative omega below -16.9 (pi * omega <= -53.4) or a TestSineGen above 7.5 cycles per sample. On each of
    them the device returns the plain-double result, bit for bit; whether the host's libm agrees depends on the host."""
    import madronalib_amd as ml
    pif = np.float32(np.pi)
    args = np.array([a for a in SINF_FMA_DIFFERS if a & 0x80000000], np.uint32).view(np.float32)   # omega <= 0.5: only negative arguments
    omegas = []
    for a in args:                                             # an omega whose pi_f * omega IS the argument
        guess = np.float32(a / pif)
        cands = (guess.view(np.uint32).astype(np.int64) + np.arange(-8, 9)).astype(np.uint32).view(np.float32)
        hit = cands[(pif * cands) == a]
        assert hit.size, float(a)
        omegas.append(hit[0])
    V = 64
    omega = np.zeros((V, 64), np.float32) + np.float32(0.1)
    for i, om in enumerate(omegas):
        omega[i, :] = om
    x = np.ones((V, 64), np.float32)
    k = np.full((V, 64), 0.7, np.float32)
    g = ml.Graph(eng, V)
    g.add("x", "input")
    g.add("om", "input")
    g.add("k", "input")
    g.add("p", "proc", Proc.LOPASS, ["x", "om", "k"])
    g.add_output("p")
    g.compile()
    (got,) = g.process_host(1, {"x": x, "om": omega, "k": k}, Layout.QUAD)
    g.close()
    # sample 0 from cleared memory is c2 * x with c2 = (2 s1 s1) / (2 + k s2): the device's s1 is the restated sinf's
    for i, a in enumerate(args):
        s1 = oracle.libm_sinf(np.array([a], np.float32))[0]
        s2 = oracle.libm_sinf(np.array([np.float32(2.0) * a], np.float32))[0]
        nrm = np.float32(1.0) / (np.float32(2.0) + np.float32(0.7) * s2)
        c2 = (np.float32(2.0) * s1 * s1) * nrm
        assert got[i, 0].view(np.uint32) == np.float32(c2).view(np.uint32), (i, float(a), got[i, 0], c2)
    # lanes with a regular omega in the same wavefront took the general path too and still match the oracle (host libm)
    st = oracle.chain_clear([Proc.LOPASS], V)
    want = oracle.proc_multi(Proc.LOPASS, 1, np.zeros((3, V), np.float32), st, [x, omega, k])
    assert_bits_equal(got[len(args):], want[len(args):], True, "regular lanes beside the irregular ones")
    n, lst = oracle.sinf_check(0, 0xFFFFFFFF)
    assert n in (0, 12) and (n == 0 or sorted(int(v) for v in lst) == sorted(SINF_FMA_DIFFERS))


def test_c_hardware_reciprocal_is_another_table(eng):
    """divideApprox = a * rcpps(b), sqrtApprox = x * rsqrtps(x) (MLDSPMathSSE.h:79, 84-85): Intel's 12-bit tables, error up to
    1.5 * 2^-12, e.g. divideApprox(1, 3) = 0x1.554p-2. gfx950's v_rcp_f32 / v_rsq_f32 are good to 1 ulp: the
    device's divideApprox(1, 3) is the correctly rounded third, 2^-13.4 away from the reference's value - inside the reference's
    own error band, outside bit parity. Contract: 2^-11 relative (tests/test_gpu_parity.py prints the measured maximum)."""
    one, three = np.full(64, 1.0, np.float32), np.full(64, 3.0, np.float32)
    got = eng.op(Op.DIVIDE_APPROX, one, three).view(np.float32)
    assert (got == got[0]).all()
    third = np.float32(1.0) / np.float32(3.0)
    assert abs(int(got[0].view(np.uint32)) - int(third.view(np.uint32))) <= 1
    ref_value = np.float32(float.fromhex("0x1.554p-2"))
    rel = abs(float(got[0]) - float(ref_value)) / float(ref_value)
    assert 0 < rel <= 1.5 * 2.0 ** -12
    four = np.full(64, 4.0, np.float32)
    root = eng.op(Op.SQRT_APPROX, four).view(np.float32)
    assert abs(float(root[0]) - 2.0) <= 2.0 * 2.0 ** -23
