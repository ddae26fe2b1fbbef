"""CPU side of the denormal contract (row a18): the plain-C oracle restatement against the COMPILED REFERENCE in both
floating-point modes — default MXCSR and ml::UsingFlushDenormalsToZero (MLDSPUtils.h:51-96: DAZ | FZ) — on operands and
recurrences that live in the denormal range. The GPU tests (tests/test_gpu_denormals.py) compare the device with the
oracle; this file is what makes the oracle's behaviour in that range the reference's."""
import numpy as np
import pytest

from inputs import assert_bits_equal, chain_coeffs, is_float_result, lcg_noise, op_inputs
from madronalib_amd.constants import Op, Proc

MODES = [pytest.param(False, id="ieee"), pytest.param(True, id="flush")]


def test_flush_scope_sets_and_restores_the_mode(oracle, ref):
    x = np.full(64, 1e-39, np.float32)
    one = np.ones(64, np.float32)
    for chk in (oracle, ref):
        assert (chk.op(Op.MULTIPLY, x, one).view(np.float32) == x).all()
        with chk.flush_denormals():
            assert (chk.op(Op.MULTIPLY, x, one).view(np.float32) == 0).all()        # DAZ: the operand reads as zero
            assert (chk.op(Op.MULTIPLY, np.full(64, 1e-20, np.float32), np.full(64, 1e-20, np.float32)).view(np.float32) == 0).all()  # FZ
            with chk.flush_denormals(False):
                assert (chk.op(Op.MULTIPLY, x, one).view(np.float32) == x).all()
            assert (chk.op(Op.MULTIPLY, x, one).view(np.float32) == 0).all()
        assert (chk.op(Op.MULTIPLY, x, one).view(np.float32) == x).all()


@pytest.mark.parametrize("flush", MODES)
@pytest.mark.parametrize("op", Op.UNARY + Op.BINARY + Op.TERNARY)
def test_ops_oracle_equals_reference_on_denormals(oracle, ref, op, flush):
    n = 64 * 64
    a, b, c = op_inputs(op, n, seed=17)
    rng = np.random.default_rng(100 + int(op))
    tiny = (rng.integers(0, 1 << 24, n, dtype=np.uint64).astype(np.uint32) | (rng.integers(0, 2, n).astype(np.uint32) << np.uint32(31)))
    small = np.float32(1e-19) * rng.standard_normal(n).astype(np.float32)
    if op not in Op.INT_INPUT and op != Op.SELECT_INT:
        a = np.where(rng.random(n) < 0.4, tiny.view(np.float32), np.where(rng.random(n) < 0.5, small, a)).astype(np.float32)
        if b is not None:
            b = np.where(rng.random(n) < 0.3, tiny[::-1].view(np.float32), np.where(rng.random(n) < 0.5, small[::-1], b)).astype(np.float32)
    with oracle.flush_denormals(flush):
        got = oracle.op(op, a, b, c)
    with ref.flush_denormals(flush):
        want = ref.op(op, a, b, c)
    if op in Op.HW_APPROX:   # rcpps / rsqrtps tables: the oracle computes these exactly (2^-11 contract)
        g, w = got.view(np.float32), want.view(np.float32)
        ok = np.isfinite(w) & np.isfinite(g) & (np.abs(w) > 1e-30)
        assert np.allclose(g[ok], w[ok], rtol=1.5 * 2.0 ** -11, atol=0)
    else:
        assert_bits_equal(got, want, is_float_result(op), f"op {op} flush={flush}")


@pytest.mark.parametrize("flush", MODES)
def test_row_reductions_on_denormals(oracle, ref, flush):
    from madronalib_amd.constants import RowOp
    rng = np.random.default_rng(8)
    rows = (rng.integers(0, 1 << 24, 64 * 200, dtype=np.uint64).astype(np.uint32) | (rng.integers(0, 2, 64 * 200).astype(np.uint32) << np.uint32(31))).view(np.float32).copy()
    rows[64 * 100:] = np.abs(rows[64 * 100:])          # all-positive denormal rows: min must not be rescued by a negative
    rows[64 * 150:] = rng.standard_normal(64 * 50).astype(np.float32) * np.float32(1e-37)
    for rowop in (RowOp.SUM, RowOp.MEAN, RowOp.MAX, RowOp.MIN):
        with oracle.flush_denormals(flush):
            got = oracle.row_reduce(rowop, rows)
        with ref.flush_denormals(flush):
            want = ref.row_reduce(rowop, rows)
        assert_bits_equal(got, want, True, f"rowop {rowop} flush={flush}")


CHAINS = {
    "lopass8": [Proc.LOPASS] * 8, "hipass4": [Proc.HIPASS] * 4, "onepole": [Proc.ONE_POLE], "bandpass": [Proc.BANDPASS],
    "dcblocker": [Proc.DC_BLOCKER], "onepole_loshelf_bell": [Proc.ONE_POLE, Proc.LO_SHELF, Proc.BELL],
    "hishelf_dc_onepole": [Proc.HI_SHELF, Proc.DC_BLOCKER, Proc.ONE_POLE], "adsr": [Proc.ADSR],
}


@pytest.mark.parametrize("flush", MODES)
@pytest.mark.parametrize("name", list(CHAINS))
def test_decaying_chains_oracle_equals_reference(oracle, ref, name, flush):
    procs = CHAINS[name]
    V, live, T = 48, 3, 1536
    co = chain_coeffs(oracle, procs, V, seed=31)
    x = np.zeros((V, 64 * T), np.float32)
    x[:, :64 * live] = lcg_noise(np.arange(V, dtype=np.uint32) + 5, 64 * live)
    if name == "adsr":
        x[:] = 0
        x[:, 10:140] = np.linspace(0.2, 1.0, V, dtype=np.float32)[:, None]
    so, sr = oracle.chain_clear(procs, V), ref.chain_clear(procs, V)
    with oracle.flush_denormals(flush):
        yo = oracle.chain_process(procs, T, co, so, x, None, n_threads=8)
    with ref.flush_denormals(flush):
        yr = ref.chain_process(procs, T, co, sr, x, None, n_threads=8)
    assert_bits_equal(yo, yr, True, f"{name} flush={flush}")
    assert_bits_equal(so, sr, False, f"{name} state flush={flush}")
    a = np.abs(yr)
    den = int(((a > 0) & (a < np.float32(1.17549435e-38))).sum())
    assert (den == 0) if flush else (den > 0 or name == "adsr")


@pytest.mark.parametrize("flush", MODES)
@pytest.mark.parametrize("kind", [Proc.SAW_GEN, Proc.PULSE_GEN, Proc.SINE_GEN])
def test_oscillators_with_tiny_frequencies(oracle, ref, kind, flush):
    V, T = 128, 4
    f = np.linspace(1e-4, 0.45, V).astype(np.float32)
    f[-8:] = np.array([1e-39, 3e-42, 1.1754944e-38, 2e-38, 1e-45, 0.0, 5e-39, 1.4e-45], np.float32)
    f[-40:-8] = np.float32(2.0) ** -np.arange(100, 132, dtype=np.float32) * np.float32(1.7)
    procs = [kind]
    co = chain_coeffs(oracle, procs, V, seed=3)
    so, sr = oracle.chain_clear(procs, V), ref.chain_clear(procs, V)
    with oracle.flush_denormals(flush):
        yo = oracle.chain_process(procs, T, co, so, None, f)
    with ref.flush_denormals(flush):
        yr = ref.chain_process(procs, T, co, sr, None, f)
    assert_bits_equal(yo, yr, True, f"proc {kind} flush={flush}")
    assert_bits_equal(so, sr, False, "state")
