"""GPU parity of row plumbing (MLDSPOps.h:1041-1383) and routing (MLDSPRouting.h:59-234): bit-exact against
the compiled reference's golden outputs and against the CPU oracle, through the C-ABI."""
import numpy as np
import pytest

from golden_cases import load_rows, rows_golden_case
from inputs import assert_bits_equal, lcg_noise
from madronalib_amd.constants import Layout, Op, Proc, Route, RowsRule
from rows_cases import ROWS_CASES, case_inputs, run as run_rows_case


@pytest.fixture(scope="module")
def eng():
    import madronalib_amd as ml
    e = ml.Engine(0)
    yield e
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(ROWS_CASES))
def test_rows_case_golden_and_oracle(eng, oracle, name):
    ins, want = rows_golden_case(load_rows(), name)
    got = run_rows_case(eng, name, ins)
    assert_bits_equal(got, want, True, name + " vs reference golden")
    ins2 = case_inputs(name, seed=77)
    assert_bits_equal(run_rows_case(eng, name, ins2), run_rows_case(oracle, name, ins2), True, name + " vs oracle")


@pytest.mark.gpu
@pytest.mark.parametrize("rule,p0,p1,rot,src_rows,dst_rows,off,step,count", [
    (RowsRule.REPEAT, 0, 0, 0, 3, 8, 0, 1, 8), (RowsRule.STRETCH, 0, 0, 0, 5, 9, 0, 1, 9), (RowsRule.STRETCH, 0, 0, 0, 4, 1, 0, 1, 1),
    (RowsRule.SHIFT, -3, 0, 1, 6, 6, 0, 1, 6), (RowsRule.ROTATE, 11, 0, -1, 7, 7, 0, 1, 7), (RowsRule.STRIDED, 1, 3, 0, 10, 9, 2, 2, 4)])
def test_rows_map_many_groups(eng, oracle, rule, p0, p1, rot, src_rows, dst_rows, off, step, count):
    """One launch over many independent arrays (one DSPVectorArray per voice)."""
    groups = 1500
    src = lcg_noise(np.arange(groups * src_rows, dtype=np.uint32), 64)
    dst0 = np.full((groups * dst_rows, 64), 7.0, np.float32)   # rows the call does not touch must keep their content
    got = eng.rows_map(rule, p0, p1, rot, src, src_rows, dst_rows, off, step, count, groups, dst0.copy())
    want = oracle.rows_map(rule, p0, p1, rot, src, src_rows, dst_rows, off, step, count, groups, dst0.copy())
    assert_bits_equal(got, want, True, f"rows_map rule {rule}")


@pytest.mark.gpu
def test_rows_add_normalize_index_many_groups(eng, oracle):
    groups, rpg = 700, 6
    x = lcg_noise(np.arange(groups * rpg, dtype=np.uint32) + 5, 64)
    assert_bits_equal(eng.rows_add(x, rpg, groups), oracle.rows_add(x, rpg, groups), True, "addRows")
    assert_bits_equal(eng.rows_normalize(x), oracle.rows_normalize(x), True, "normalize")
    assert_bits_equal(eng.rows_index(rpg, groups), oracle.rows_index(rpg, groups), True, "rowIndex")


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 5, 8])
@pytest.mark.parametrize("linear", [False, True])
def test_routing_flat_vs_oracle(eng, oracle, n, linear):
    rng = np.random.default_rng(n)
    N = 64 * 333
    sel_full = rng.uniform(0.0, 3.0, N).astype(np.float32)
    sel_full[:6] = [0.0, 0.25, 0.5, 0.99999994, 1.0, 2.75]
    sel_row = rng.uniform(0.0, 1.0, 64).astype(np.float32)
    ins = [lcg_noise(np.arange(1, dtype=np.uint32) + 10 * k, N)[0] for k in range(n)]
    for sel in (sel_full, sel_row):   # a selector per sample, and the reference's one selector row for every row
        assert_bits_equal(eng.multiplex(sel, ins, linear), oracle.multiplex(sel, ins, linear), True, f"multiplex n={n}")
        got, want = eng.demultiplex(sel, ins[0], n, linear), oracle.demultiplex(sel, ins[0], n, linear)
        for j in range(n):
            assert_bits_equal(got[j], want[j], True, f"demultiplex n={n} out {j}")
    if not linear:  # REQUIRE(addRows(demultiplex(sel, x)) == x), the reference author's note MLDSPRouting.h:242
        total = np.sum(np.stack(eng.demultiplex(sel_full, ins[0], n, False)), 0)
        assert (total == ins[0]).all()


@pytest.mark.gpu
def test_routing_nodes_in_a_graph(eng, oracle):
    """multiplex / multiplexLinear / demultiplex / demultiplexLinear as fused graph nodes == the flat routing calls."""
    import madronalib_amd as ml
    V, T = 90, 4
    desc = [dict(name="sel", type="input"), dict(name="a", type="input"), dict(name="b", type="input"),
            dict(name="saw", type="proc", kind=Proc.SAW_GEN, inputs=["f"]) if False else dict(name="c", type="input"),
            dict(name="mux", type="route", kind=Route.MULTIPLEX, inputs=["sel", "a", "b", "c"]),
            dict(name="muxl", type="route", kind=Route.MULTIPLEX_LINEAR, inputs=["sel", "a", "b", "c"]),
            dict(name="d1", type="route", kind=Route.DEMULTIPLEX, inputs=["sel", "a"], index=1, n_outputs=3),
            dict(name="dl2", type="route", kind=Route.DEMULTIPLEX_LINEAR, inputs=["sel", "b"], index=2, n_outputs=3)]
    g = ml.Graph(eng, V, desc, ["mux", "muxl", "d1", "dl2"])
    rng = np.random.default_rng(2)
    sig = {"sel": rng.uniform(0, 2, (V, 64 * T)).astype(np.float32)}
    for k, nm in enumerate("abc"):
        sig[nm] = lcg_noise(np.arange(V, dtype=np.uint32) + 100 * k, 64 * T)
    got = g.process_host(T, sig, Layout.QUAD)
    ins = [sig["a"], sig["b"], sig["c"]]
    assert_bits_equal(got[0], oracle.multiplex(sig["sel"], ins, False), True, "graph multiplex")
    assert_bits_equal(got[1], oracle.multiplex(sig["sel"], ins, True), True, "graph multiplexLinear")
    assert_bits_equal(got[2], oracle.demultiplex(sig["sel"], sig["a"], 3, False)[1], True, "graph demultiplex")
    assert_bits_equal(got[3], oracle.demultiplex(sig["sel"], sig["b"], 3, True)[2], True, "graph demultiplexLinear")


@pytest.mark.gpu
def test_rows_error_paths(eng):
    import madronalib_amd as ml
    x = np.zeros((4, 64), np.float32)
    with pytest.raises(ml.MlgpuError):
        eng.rows_map(99, 0, 0, 0, x, 4, 4, 0, 1, 4, 1)       # unknown rule
    with pytest.raises(ml.MlgpuError):
        eng.rows_map(RowsRule.REPEAT, 0, 0, 0, x, 4, 4, 2, 1, 4, 1)   # destination rows out of range
    with pytest.raises(ml.MlgpuError):
        eng.rows_map(RowsRule.REPEAT, 0, 0, 2, x, 4, 4, 0, 1, 4, 1)   # sample_rotate must be -1, 0, +1
    with pytest.raises(ml.MlgpuError):
        eng.multiplex(x[0], [x[0]] * 9)                       # at most 8 signals


@pytest.mark.gpu
def test_validate_scan(eng):
    """mlgpu_validate = ml::validate (MLDSPOps.h:1430-1445) over a whole device signal: NaN or |x| > 1e8."""
    rng = np.random.default_rng(12)
    for n in (64, 64 * 1000 + 3, 64 * 40000 + 1):
        x = (rng.standard_normal(n) * 1e3).astype(np.float32)
        d = eng.to_device(np.concatenate([x, np.zeros(4, np.float32)]))
        assert eng.validate(d, n) == (0, None)
        bad = sorted(rng.choice(n, 7, replace=False).tolist())
        vals = [np.nan, 1.5e8, -2e8, np.inf, -np.inf, np.nan, 1.0000001e8]
        for i, v in zip(bad, vals):
            x[i] = v
        x[(bad[0] + 1) % n] = 1e8 if (bad[0] + 1) % n not in bad else x[(bad[0] + 1) % n]     # exactly 1e8 is still fine
        d = eng.to_device(np.concatenate([x, np.zeros(4, np.float32)]))
        want = int((np.isnan(x) | (np.abs(x) > np.float32(1e8))).sum())
        assert eng.validate(d, n) == (want, bad[0])
