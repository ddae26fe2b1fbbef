"""GPU parity of run-time defined graphs (BASELINE configs[4]: a 16-node synth patch) and of chains
fused at run time with hiprtc, against the CPU oracle evaluating the same graph node by node.
Bit-exact (no hardware-approximate node is used in these graphs)."""
import numpy as np
import pytest

from graph_oracle import evaluate
from inputs import assert_bits_equal, gate_signal, lcg_noise
from madronalib_amd import patches
from madronalib_amd.constants import Layout, Op, Proc


def test_jit_selftest_compiles_without_gpu():
    """The run-time code generator's output (a chain and a graph) compiles for gfx950 on any box."""
    import madronalib_amd as ml
    st, log = ml.jit_selftest()
    assert st == 0, log


@pytest.fixture(scope="module")
def eng():
    import madronalib_amd as ml
    e = ml.Engine(0)
    yield e
    e.close()


def synth16_setup(oracle, V, seed=0):
    rng = np.random.default_rng(seed)
    params = dict(pitch=rng.uniform(-2.0, 3.0, V).astype(np.float32), baseFreq=np.float32(110.0 / 48000.0),
                  width=rng.uniform(0.1, 0.9, V).astype(np.float32),
                  lfoFreq=(rng.uniform(0.1, 8.0, V) / 48000.0).astype(np.float32),
                  noiseLevel=rng.uniform(0.0, 0.3, V).astype(np.float32))
    coeffs = dict(
        lp=np.stack([oracle.make_coeffs("lopass", rng.uniform(0.01, 0.3), rng.uniform(0.3, 1.5)) for _ in range(V)], 1),
        hp=np.stack([oracle.make_coeffs("hipass", rng.uniform(0.0005, 0.01), rng.uniform(0.7, 1.5)) for _ in range(V)], 1),
        smooth=np.stack([oracle.make_coeffs("onepole", rng.uniform(0.1, 0.4)) for _ in range(V)], 1),
        dc=np.array([[oracle.dcblocker_coeffs(rng.uniform(0.01, 0.1)) for _ in range(V)]], np.float32),
        env=np.stack([oracle.make_coeffs("adsr", rng.uniform(0.0005, 0.01), rng.uniform(0.002, 0.02), rng.uniform(0.2, 0.9),
                                         rng.uniform(0.002, 0.03), 48000.0) for _ in range(V)], 1))
    return params, coeffs


def build_synth16(eng, oracle, V, params, coeffs, seeds):
    import madronalib_amd as ml
    desc, outs = patches.synth16()
    g = ml.Graph(eng, V, desc, outs)
    g.clear()
    for k, v in params.items():
        g.set_param(k, v if np.ndim(v) else float(v))
    for k, c in coeffs.items():
        g.set_coeffs(k, [np.ascontiguousarray(row) for row in c])
    g.set_state("noise", 0, seeds)
    states = {n["name"]: oracle.chain_clear([n["kind"]], V) for n in desc if n["type"] == "proc"}
    states["noise"][0] = seeds
    return g, desc, outs, states


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [Layout.QUAD, Layout.ROWS])
def test_synth16_vs_oracle(eng, oracle, layout):
    V, T = 300, 24
    params, coeffs = synth16_setup(oracle, V, seed=4)
    seeds = np.arange(V, dtype=np.uint32) * np.uint32(2654435761)
    g, desc, outs, states = build_synth16(eng, oracle, V, params, coeffs, seeds)
    assert "mlgpu_graph_kernel" in g.source and "p10.next" in g.source or "next(" in g.source
    gate = gate_signal(V, 64 * T * 2, seed=9)
    for call in range(2):  # second call resumes from carried state
        sig = {"gate": np.ascontiguousarray(gate[:, call * 64 * T:(call + 1) * 64 * T])}
        (got,) = g.process_host(T, sig, layout)
        (want,) = evaluate(oracle, desc, outs, V, T, sig, params, coeffs, states)
        assert_bits_equal(got, want, True, f"synth16 call {call}")
    for n in desc:
        if n["type"] == "proc":
            for i in range(g.num_state(n["name"])):
                assert (g.get_state(n["name"], i) == states[n["name"]][i]).all(), (n["name"], i)
    assert np.abs(got).max() <= 1.0 and np.abs(got).max() > 0.01  # the output clamp works and there is sound


@pytest.mark.gpu
def test_synth16_full_size_subset(eng, oracle):
    """BASELINE configs[4] per-GPU size (262 144 voices): a strided subset against the oracle, and
    launch splitting: 2 x T/2 vectors == T vectors."""
    V, T = 262144, 4
    sub = np.arange(0, V, 1021)
    rng = np.random.default_rng(1)
    base, _ = synth16_setup(oracle, sub.size, seed=2)
    # full-size params by tiling the subset's parameters so the subset voices are known exactly
    idx = np.arange(V) % sub.size
    params = {k: (v[idx] if np.ndim(v) else v) for k, v in base.items()}
    _, co_sub = synth16_setup(oracle, sub.size, seed=2)
    coeffs = {k: np.ascontiguousarray(c[:, idx]) for k, c in co_sub.items()}
    seeds = (np.arange(V, dtype=np.uint32) + np.uint32(17))
    g, desc, outs, _ = build_synth16(eng, oracle, V, params, coeffs, seeds)
    gate_sub = gate_signal(sub.size, 64 * T, seed=3)
    pos = np.full(V, -1, np.int64)
    pos[sub] = np.arange(sub.size)
    # every voice gets the gate of (voice mod subset size); subset voices get gate_sub rows in order
    gate_rows = np.where(pos >= 0, pos, idx)
    d_gate_vm = eng.to_device(gate_sub[gate_rows])
    n = V * T * 64
    d_gate = eng.alloc(4 * n)
    eng.layout_convert(d_gate_vm, Layout.VOICE_MAJOR, d_gate, Layout.QUAD, V, T)
    d_out = eng.alloc(4 * n)
    g.process(T, [d_gate], [d_out])
    q = d_out.download(np.float32).reshape(T * 16, V, 4)
    got = q[:, sub, :].transpose(1, 0, 2).reshape(sub.size, T * 64)
    p_sub = {k: (v[sub] if np.ndim(v) else v) for k, v in params.items()}
    c_sub = {k: np.ascontiguousarray(c[:, sub]) for k, c in coeffs.items()}
    states = {nn["name"]: oracle.chain_clear([nn["kind"]], sub.size) for nn in desc if nn["type"] == "proc"}
    states["noise"][0] = seeds[sub]
    (want,) = evaluate(oracle, desc, outs, sub.size, T, {"gate": gate_sub}, p_sub, c_sub, states)
    assert_bits_equal(got, want, True, "synth16 full-size subset")
    # split launches from a fresh graph (ADSR::clear() only resets the segment, MLDSPFilters.h:698, so a
    # cleared graph is NOT in its initial state; identical source => the compiled module is reused)
    g.close()
    g, _, _, _ = build_synth16(eng, oracle, V, params, coeffs, seeds)
    half = T // 2
    hq = V * half * 64
    d_g1, d_g2, d_o1, d_o2 = eng.alloc(4 * hq), eng.alloc(4 * hq), eng.alloc(4 * hq), eng.alloc(4 * hq)
    gq = d_gate.download(np.float32)
    d_g1.upload(gq[:hq])
    d_g2.upload(gq[hq:])
    g.process(half, [d_g1], [d_o1])
    g.process(half, [d_g2], [d_o2])
    joined = np.concatenate([d_o1.download(np.float32), d_o2.download(np.float32)])
    assert (joined.view(np.uint32) == q.reshape(-1).view(np.uint32)).all()


@pytest.mark.gpu
def test_graph_with_masks_and_two_outputs(eng, oracle):
    """compare -> mask -> select, a 3-input op, a const, two outputs, two inputs."""
    import madronalib_amd as ml
    V, T = 130, 3
    desc = [dict(name="a", type="input"), dict(name="b", type="input"), dict(name="half", type="const", value=0.5),
            dict(name="gt", type="op", kind=Op.GREATER_THAN, inputs=["a", "b"]),
            dict(name="sel", type="op", kind=Op.SELECT, inputs=["a", "b", "gt"]),      # max(a, b) by mask
            dict(name="mix", type="op", kind=Op.LERP, inputs=["a", "b", "half"]),
            dict(name="lp", type="proc", kind=Proc.LOPASS, inputs=["sel"]),
            dict(name="sat", type="op", kind=Op.SIN_APPROX, inputs=["mix"])]
    g = ml.Graph(eng, V, desc, ["lp", "sat"])
    co = np.stack([oracle.make_coeffs("lopass", 0.1, 0.8)] * V, 1)
    g.set_coeffs("lp", [np.ascontiguousarray(r) for r in co])
    sig = {"a": lcg_noise(np.arange(V, dtype=np.uint32), 64 * T), "b": lcg_noise(np.arange(V, dtype=np.uint32) + 999, 64 * T)}
    got = g.process_host(T, sig, Layout.VOICE_MAJOR)
    states = {"lp": oracle.chain_clear([Proc.LOPASS], V)}
    want = evaluate(oracle, desc, ["lp", "sat"], V, T, sig, {}, {"lp": co}, states)
    assert_bits_equal(got[0], want[0], True, "lp(select)")
    assert_bits_equal(got[1], want[1], True, "sinApprox(lerp)")


@pytest.mark.gpu
def test_graph_error_paths(eng):
    import madronalib_amd as ml
    g = ml.Graph(eng, 64)
    a = g.add("a", "input")
    with pytest.raises(ml.MlgpuError):
        g.add("bad", "op", Op.ADD, [a])            # wrong arity
    with pytest.raises(ml.MlgpuError):
        g.add("bad2", "proc", 999, [a])            # unknown processor
    with pytest.raises(ml.MlgpuError):
        g.add("bad3", "proc", Proc.LOPASS, [57])   # unknown input node
    with pytest.raises(ml.MlgpuError):
        g.compile()                                # no outputs
    lp = g.add("lp", "proc", Proc.LOPASS, [a])
    g.add_output(lp)
    g.compile()
    with pytest.raises(ml.MlgpuError):
        g.add("late", "op", Op.ABS, [lp])          # already compiled
    with pytest.raises(ml.MlgpuError):
        g.set_coeff("lp", 5, 1.0)
