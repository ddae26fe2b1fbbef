"""GPU parity of run-time defined graphs (BASELINE configs[4]: a 16-node synth patch) and of chains
fused at run time with hiprtc, against the CPU oracle evaluating the same graph node by node.
Bit-exact (no hardware-approximate node is used in these graphs)."""
import os

import numpy as np
import pytest

from graph_oracle import evaluate
from golden_cases import load_multi, multi_golden_case
from inputs import MULTI_CASES, assert_bits_equal, gate_signal, lcg_noise, multi_case, multi_inputs_audio, stepped
from madronalib_amd import patches
from madronalib_amd.constants import Layout, Op, Proc, Vop


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


def build_synth16(eng, oracle, V, params, coeffs, seeds, voices_per_lane=0):
    import madronalib_amd as ml
    desc, outs = patches.synth16()
    g = ml.Graph(eng, V, desc, outs, voices_per_lane=voices_per_lane)
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
@pytest.mark.parametrize("vpl", [1, 2])
@pytest.mark.parametrize("layout", [Layout.QUAD, Layout.ROWS])
def test_synth16_vs_oracle(eng, oracle, layout, vpl):
    V, T = 300, 24   # 300 voices: with two voices per lane the second half-block is ragged
    params, coeffs = synth16_setup(oracle, V, seed=4)
    seeds = np.arange(V, dtype=np.uint32) * np.uint32(2654435761)
    g, desc, outs, states = build_synth16(eng, oracle, V, params, coeffs, seeds, vpl)
    assert f"({vpl} voice" in g.source
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
@pytest.mark.parametrize("vpl", [1, 2])
def test_synth16full_golden_and_oracle(eng, oracle, vpl):
    """patches.synth16(full=True) = config 5 as SURVEY 8d lists it: Lopass(x, omega, k) with per-sample coefficients (device
    libm sinf), two ADSRs, the cutoff through exp2Approx. Against the golden of the reference's own objects, and against
    the oracle evaluator for more voices with a resumed second launch."""
    import os
    import madronalib_amd as ml
    from madronalib_amd.sharding import cfg5_voice_params
    g0 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "synth16full.npz"))
    desc, outs = patches.synth16(full=True)

    def build(V, params, coeffs, seeds):
        g = ml.Graph(eng, V, desc, outs, voices_per_lane=vpl)
        g.clear()
        for k, v in params.items():
            g.set_param(k, v if np.ndim(v) else float(v))
        for k, c in coeffs.items():
            g.set_coeffs(k, [np.ascontiguousarray(row) for row in c])
        g.set_state("noise", 0, seeds)
        return g
    V, T = g0["gate"].shape[0], g0["gate"].shape[1] // 64
    params = {k[2:]: (g0[k] if g0[k].ndim else float(g0[k])) for k in g0.files if k.startswith("p_")}
    coeffs = {k[2:]: g0[k] for k in g0.files if k.startswith("c_")}
    g = build(V, params, coeffs, g0["seeds"])
    half = T // 2
    a = g.process_host(half, {"gate": np.ascontiguousarray(g0["gate"][:, :64 * half])}, Layout.QUAD)[0]
    b = g.process_host(T - half, {"gate": np.ascontiguousarray(g0["gate"][:, 64 * half:])}, Layout.VOICE_MAJOR)[0]
    assert_bits_equal(np.concatenate([a, b], 1), g0["out"], True, "synth16full golden")
    g.close()
    V, T = 300, 20
    params, coeffs, seeds = cfg5_voice_params(1000, 1000 + V, 4096, ml, full=True)
    g = build(V, params, coeffs, seeds)
    states = {n["name"]: oracle.chain_clear([n["kind"]], V) for n in desc if n["type"] == "proc"}
    states["noise"][0] = seeds
    gate = gate_signal(V, 64 * T * 2, seed=5)
    for call in range(2):
        sig = {"gate": np.ascontiguousarray(gate[:, call * 64 * T:(call + 1) * 64 * T])}
        (got,) = g.process_host(T, sig, Layout.QUAD)
        (want,) = evaluate(oracle, desc, outs, V, T, sig, params, coeffs, states)
        assert_bits_equal(got, want, True, f"synth16full call {call}")
    g.close()


@pytest.mark.gpu
def test_synth16_golden_reference_objects(eng, oracle):
    """tests/golden/synth16.npz: the patch run with the reference's own objects (BASELINE configs[4] parameters)."""
    import os
    g0 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "synth16.npz"))
    V, T = g0["gate"].shape[0], g0["gate"].shape[1] // 64
    params = {k[2:]: (g0[k] if g0[k].ndim else float(g0[k])) for k in g0.files if k.startswith("p_")}
    coeffs = {k[2:]: g0[k] for k in g0.files if k.startswith("c_")}
    for vpl in (1, 2):
        g, _, _, _ = build_synth16(eng, oracle, V, params, coeffs, g0["seeds"], vpl)
        half = T // 2
        a = g.process_host(half, {"gate": np.ascontiguousarray(g0["gate"][:, :64 * half])}, Layout.QUAD)[0]
        b = g.process_host(T - half, {"gate": np.ascontiguousarray(g0["gate"][:, 64 * half:])}, Layout.VOICE_MAJOR)[0]
        assert_bits_equal(np.concatenate([a, b], 1), g0["out"], True, f"synth16 golden, {vpl} voice(s) per lane")
        g.close()


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
@pytest.mark.parametrize("full", [False, True])
def test_config5_bench_step_subset(eng, oracle, full):
    """BASELINE configs[4] exactly as bench.py --workload cfg5 / cfg5full runs it - 262 144 voices, the bench's own per-voice parameters,
    coefficients, noise seeds and gate (madronalib_amd/sharding.py), one bench step = 16 launches of 16 DSPVectors with carried state -
    against the oracle on 257 voices spread over the range: launches 0, 7 and 15, and every processor's final state."""
    import madronalib_amd as ml
    from madronalib_amd.sharding import cfg5_gate_quad, cfg5_voice_params
    V, T, L = 262144, 16, 16
    desc, outs = patches.synth16(full=full)
    g = ml.Graph(eng, V, desc, outs)
    g.clear()
    params, coeffs, seeds = cfg5_voice_params(0, V, V, ml, full=full)
    for k, v in params.items():
        g.set_param(k, v if np.ndim(v) else float(v))
    for k, c in coeffs.items():
        g.set_coeffs(k, [np.ascontiguousarray(r) for r in c])
    g.set_state("noise", 0, seeds)
    gate_q = cfg5_gate_quad(0, V, T)                       # the same gate block every launch, as in the bench
    d_gate = eng.to_device(gate_q)
    d_out = eng.alloc(4 * V * T * 64)
    sub = np.concatenate([np.arange(0, V, 1021), [V - 1]])[:257]
    gate_sub = np.ascontiguousarray(gate_q[:, sub, :].transpose(1, 0, 2).reshape(sub.size, T * 64))
    p_sub = {k: (v[sub] if np.ndim(v) else v) for k, v in params.items()}
    c_sub = {k: np.ascontiguousarray(np.asarray(c)[:, sub]) for k, c in coeffs.items()}
    states = {n["name"]: oracle.chain_clear([n["kind"]], sub.size) for n in desc if n["type"] == "proc"}
    states["noise"][0] = seeds[sub]
    # (every voice over eight launches: test_config5_every_voice below - skipped, loudly, where the compiled reference is absent)
    for launch in range(L):
        g.process(T, [d_gate], [d_out])
        (want,) = evaluate(oracle, desc, outs, sub.size, T, {"gate": gate_sub}, p_sub, c_sub, states)
        if launch in (0, 1, 7, L - 1):
            q = d_out.download(np.float32).reshape(T * 16, V, 4)
            got = q[:, sub, :].transpose(1, 0, 2).reshape(sub.size, T * 64)
            assert_bits_equal(got, want, True, f"config 5 (full={full}) launch {launch}")
            assert np.abs(want).max() > 0.01
    for n in desc:
        if n["type"] == "proc":
            for i in range(g.num_state(n["name"])):
                assert (g.get_state(n["name"], i)[sub] == states[n["name"]][i]).all(), (n["name"], i)
    g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("full", [False, True])
def test_config5_every_voice(eng, full):
    """BASELINE configs[4], ALL 262 144 voices over 8 launches of 16 DSPVectors with carried state (8 192 samples per voice, both
    patches): every output word of every launch against the same voice written with the reference's own objects and run from the
    start on the host threads, slab of voices by slab (what tools/cfg5_soak.py does for 256 launches, here as a test the driver
    runs). A wrong state word of any processor shows in the launch after it. Skipped - not passed - without the compiled reference."""
    import madronalib_amd as ml
    from cpu_checkers import fast_checker, host_threads
    from madronalib_amd.sharding import cfg5_gate_quad, cfg5_voice_params
    fast = fast_checker()
    if fast is None:
        pytest.skip("every-voice comparison needs oracle/_ref/libmlref.so (the reference compiled by oracle/Makefile); only the strided-subset test ran")
    V, T, L = 262144, 16, 8
    desc, outs = patches.synth16(full=full)
    g = ml.Graph(eng, V, desc, outs)
    g.clear()
    params, coeffs, seeds = cfg5_voice_params(0, V, V, ml, full=full)
    for k, v in params.items():
        g.set_param(k, v if np.ndim(v) else float(v))
    for k, c in coeffs.items():
        g.set_coeffs(k, [np.ascontiguousarray(r) for r in c])
    g.set_state("noise", 0, seeds)
    gate_q = cfg5_gate_quad(0, V, T)
    d_gate = eng.to_device(gate_q)
    d_out = eng.alloc(4 * V * T * 64)
    got = []
    for launch in range(L):
        g.process(T, [d_gate], [d_out])
        got.append(d_out.download(np.float32).reshape(T * 16, V, 4))
    g.close()
    run = fast.synth16full_run if full else fast.synth16_run
    slab = 16384
    for a in range(0, V, slab):
        gate = np.ascontiguousarray(gate_q[:, a:a + slab, :].transpose(1, 0, 2).reshape(slab, T * 64))
        p = {k: (np.asarray(v)[a:a + slab] if np.ndim(v) else v) for k, v in params.items()}
        c = {k: np.ascontiguousarray(np.asarray(cc)[:, a:a + slab]) for k, cc in coeffs.items()}
        want = run(p, c, seeds[a:a + slab], np.tile(gate, (1, L)), host_threads())[0].reshape(slab, L, T * 64)
        for launch in range(L):
            gq = got[launch][:, a:a + slab, :].transpose(1, 0, 2).reshape(slab, T * 64)
            assert_bits_equal(gq, want[:, launch], True, f"config 5 (full={full}) launch {launch}, voices {a}..")


@pytest.mark.gpu
def test_patch_swap_with_async_compile(eng):
    """A host edits its patch while it plays: the old graph keeps processing one 512-frame block per 10.67 ms period (48 kHz) while
    the new one - a description no cache has seen, so hiprtc really runs: seconds - compiles on the library's thread
    (mlgpu_graph_compile_async), polled once per block. No block may take longer than its period; the swapped-in graph gives the
    bits of the same description compiled the ordinary way."""
    import gc
    import time
    import madronalib_amd as ml
    from madronalib_amd.sharding import cfg5_gate_quad, cfg5_voice_params
    V, T = 65536, 8
    period = 64 * T / 48000.0

    def setup(g):
        params, coeffs, seeds = cfg5_voice_params(0, V, V, ml)
        g.clear()
        for k, v in params.items():
            g.set_param(k, v if np.ndim(v) else float(v))
        for k, c in coeffs.items():
            g.set_coeffs(k, [np.ascontiguousarray(r) for r in c])
        g.set_state("noise", 0, seeds)
    desc, outs = patches.synth16()
    old = ml.Graph(eng, V, desc, outs)
    setup(old)
    salt = float(np.float32(0.5 + (time.time_ns() % 1000003) * 1e-7))
    desc2 = desc + [dict(name="salt", type="const", value=salt), dict(name="salted", type="op", kind=Op.MULTIPLY, inputs=[outs[0], "salt"])]
    new = ml.Graph(eng, V, desc2, ["salted"], compile_now=False)
    d_gate = eng.to_device(cfg5_gate_quad(0, V, T))
    d_out = eng.alloc(4 * V * T * 64)
    for _ in range(3):
        old.process(T, [d_gate], [d_out])
    eng.sync()
    compiles0 = ml.jit_stats()["compiles"]
    with pytest.raises(ml.MlgpuError):
        new.process(T, [d_gate], [d_out])                 # not compiled yet
    new.compile_async()
    assert eng.L.mlgpu_graph_process(new.h, T, None, 0, None, 0) == ml.BUSY
    gc_was = gc.isenabled()
    gc.disable()
    block_s, ready_after, t_start = [], None, time.perf_counter()
    try:
        nxt = time.perf_counter()
        for b in range(20000):
            while time.perf_counter() < nxt:
                pass
            t0 = time.perf_counter()
            old.process(T, [d_gate], [d_out])
            eng.sync()
            ready = new.compile_poll()
            block_s.append(time.perf_counter() - t0)
            nxt = max(nxt + period, time.perf_counter())
            if ready:
                ready_after = time.perf_counter() - t_start
                break
    finally:
        if gc_was:
            gc.enable()
    assert ready_after is not None, "the compile did not finish in 20 000 blocks"
    assert ml.jit_stats()["compiles"] > compiles0 and len(block_s) > 20, (ready_after, len(block_s))   # hiprtc really ran (once or twice: the register budget), for many blocks
    assert max(block_s) < period, f"a block took {max(block_s) * 1e3:.2f} ms of a {period * 1e3:.2f} ms period while the new patch compiled"
    print(f"\npatch swap: cold compile ready after {ready_after:.2f} s = {len(block_s)} blocks of {period * 1e3:.2f} ms; slowest block {max(block_s) * 1e3:.3f} ms")
    # the new patch takes over; its twin compiled the ordinary way (a memory-cache hit now) gives the same bits
    setup(new)
    twin = ml.Graph(eng, V, desc2, ["salted"])
    setup(twin)
    d_out2 = eng.alloc(4 * V * T * 64)
    for _ in range(2):
        new.process(T, [d_gate], [d_out])
        twin.process(T, [d_gate], [d_out2])
    a, b = d_out.download(np.float32), d_out2.download(np.float32)
    assert_bits_equal(a, b, True, "async-compiled graph vs its synchronously compiled twin")
    assert np.abs(a).max() > 0
    for g in (old, new, twin):
        g.close()


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


# ---- the other operator() forms, control-rate inputs, vector-rate ramps, index generators ---------------------

def single_node_graph(eng, V, case):
    """A graph with one processor node fed by the case's inputs (audio inputs / control inputs)."""
    import madronalib_amd as ml
    g = ml.Graph(eng, V)
    names = []
    for i, (rate, _) in enumerate(case["inputs"]):
        names.append(f"in{i}")
        g.add(names[-1], "input" if rate == "audio" else "control")
    g.add("p", "proc", case["kind"], names)
    g.add_output("p")
    g.compile()
    return g, names


def run_case_calls(g, names, case, T, state0, layout):
    """Two consecutive calls of T vectors; returns outputs and the state after each call."""
    V = state0.shape[1]
    for i in range(state0.shape[0]):
        g.set_state("p", i, state0[i])
    g.set_coeffs("p", [np.ascontiguousarray(r) for r in case["coeffs"]])
    outs, states = [], []
    for call in range(2):
        sig = {}
        for nm, (rate, a) in zip(names, case["inputs"]):
            w = 64 * T if rate == "audio" else T
            sig[nm] = np.ascontiguousarray(a[:, call * w:(call + 1) * w])
        (got,) = g.process_host(T, sig, layout)
        outs.append(got)
        states.append(np.stack([g.get_state("p", i) for i in range(state0.shape[0])]))
    return outs, states


@pytest.mark.gpu
@pytest.mark.parametrize("name", MULTI_CASES)
def test_multi_input_forms_vs_oracle(eng, oracle, name):
    V, T = 200, 24
    case = multi_case(oracle, name, V, 2 * T, seed=21)
    g, names = single_node_graph(eng, V, case)
    st = oracle.chain_clear([case["kind"]], V)
    outs, states = run_case_calls(g, names, case, T, st.copy(), Layout.QUAD)
    ins = multi_inputs_audio(case, 2 * T)
    for call in range(2):
        sl = slice(call * 64 * T, (call + 1) * 64 * T)
        want = oracle.proc_multi(case["kind"], T, case["coeffs"], st, [np.ascontiguousarray(x[:, sl]) for x in ins])
        assert_bits_equal(outs[call], want, True, f"{name} call {call}")
        assert_bits_equal(states[call], st, False, f"{name} state after call {call}")


@pytest.mark.gpu
def test_lopass_per_sample_coefficients_every_sinf_form(eng, oracle):
    """Lopass(x, omega, k) (MLDSPFilters.h:136-152) picks the form of its two sinf per sample for the whole wavefront: the
    sequences proven on [2^-12, pi_f] (the cosine polynomial only when a lane is in quadrant 1) with the range-free division when
    every lane is regular, glibc's own algorithm with the IEEE division otherwise. Wavefronts (64 voices) of every kind, against
    the oracle (host libm): omega swept densely through both quadrant thresholds and the 0.5 clamp, wavefronts that stay in
    quadrant 0, and wavefronts in which a few lanes are irregular - omega negative, zero, tiny, NaN, infinite; k past 1.98, huge,
    NaN, negative - for a few samples or throughout."""
    V, T = 64 * 8, 6
    S = 64 * T
    n = np.arange(S)[None, :]
    v = np.arange(V)[:, None]
    w = v // 64
    x = lcg_noise(np.arange(V, dtype=np.uint32) + 5, S)
    thr1, thr2 = np.uint32(0x3F490FDB).view(np.float32), np.uint32(0x4016CBE4).view(np.float32)
    pif = np.float32(np.pi)
    omega = np.empty((V, S), np.float32)
    k = (0.9 + 0.85 * np.sin(n * 0.013 + v)).astype(np.float32)       # 0.05 .. 1.75: regular
    # wavefront 0: every lane below quadrant 1 (2 pi omega < 0.75); 1: a dense sweep 0 .. 0.5 .. clamp; 2, 3: around the thresholds
    omega[0:64] = (0.002 + 0.1 * (0.5 + 0.5 * np.sin(n * 0.02 + v[0:64]))).astype(np.float32)
    omega[64:128] = ((n + 7 * (v[64:128] - 64)) % 640 / 1200.0).astype(np.float32)
    for base, thr, scale in ((128, thr1, 1.0), (160, thr1, 2.0), (192, thr2, 2.0), (224, pif, 2.0)):
        # omega whose pi * omega (scale 1) or 2 pi * omega (scale 2) lands within a few ulps of thr
        c = np.float32(thr / (scale * pif))
        ulps = (((n + 3 * v[base:base + 32]) % 41) - 20).astype(np.int32)
        omega[base:base + 32] = (c.view(np.uint32).astype(np.int64) + ulps).astype(np.uint32).view(np.float32)
    # wavefronts 4 - 7: regular sweeps with irregular lanes
    omega[256:] = (0.25 + 0.24 * np.sin(n * 0.004 * (1 + v[256:] % 5))).astype(np.float32)
    bad_omega = np.array([-0.1, 0.0, -0.0, 1e-5, 2.0 ** -14, np.nan, np.inf, -np.inf, 7.0, 1e-42], np.float32)
    bad_k = np.array([1.99, 2.0, 2.5, 1e6, np.nan, -3.0, np.inf, 3e38, 1.9800001, 0.0], np.float32)
    for i in range(10):
        omega[256 + 6 * i, 40:70] = bad_omega[i]                       # wavefront 4 (and 5): for 30 samples
        omega[384 + 5 * i, :] = bad_omega[i]                           # wavefront 6: throughout
        k[320 + 6 * i, 100:140] = bad_k[i]                             # wavefront 5
        k[448 + 6 * i, :] = bad_k[i]                                   # wavefront 7
    case = dict(kind=Proc.LOPASS, coeffs=np.zeros((3, V), np.float32), inputs=[("audio", x), ("audio", omega), ("audio", k)])
    g, names = single_node_graph(eng, V, case)
    st = oracle.chain_clear([Proc.LOPASS], V)
    (got,) = g.process_host(T, {names[0]: x, names[1]: omega, names[2]: k}, Layout.QUAD)
    want = oracle.proc_multi(Proc.LOPASS, T, case["coeffs"], st, [x, omega, k])
    gst = np.stack([g.get_state("p", i) for i in range(2)])
    both_nan = np.isnan(got) & np.isnan(want)
    bad = (got.view(np.uint32) != want.view(np.uint32)) & ~both_nan
    assert not bad.any(), (np.argwhere(bad)[:8], got[bad][:8], want[bad][:8])
    sb = (gst.view(np.uint32) != st.view(np.uint32)) & ~(np.isnan(gst.view(np.float32)) & np.isnan(st.view(np.float32)))
    assert not sb.any(), np.argwhere(sb)[:8]
    assert np.isfinite(want[:256]).all() and (np.abs(want[:256]) > 0).mean() > 0.9     # the regular wavefronts really filter


@pytest.mark.gpu
@pytest.mark.parametrize("name", MULTI_CASES)
def test_multi_input_forms_golden(eng, name):
    """Against the compiled reference's own outputs (tests/golden/multi.npz)."""
    c = multi_golden_case(load_multi(), name)
    T = c["out"][0].shape[1] // 64
    V = c["state0"].shape[1]
    g, names = single_node_graph(eng, V, c)
    outs, states = run_case_calls(g, names, c, T, c["state0"].copy(), Layout.VOICE_MAJOR)
    for call in range(2):
        assert_bits_equal(outs[call], c["out"][call], True, f"{name} out{call + 1}")
        assert_bits_equal(states[call], c["state"][call], False, f"{name} state{call + 1}")


@pytest.mark.gpu
def test_vector_generators_and_rates(eng, oracle):
    """columnIndex / rangeOpen / rangeClosed / interpolateDSPVectorLinear from param, const and control nodes; the
    float-mix lerp (MLDSPOps.h:753) as LERP with a control third operand; golden from the compiled reference."""
    import madronalib_amd as ml
    d = load_multi()
    V, T = d["vop_a"].shape
    desc = [dict(name="a", type="control"), dict(name="b", type="control"),
            dict(name="idx", type="vop", kind=Vop.COLUMN_INDEX, inputs=[]),
            dict(name="ro", type="vop", kind=Vop.RANGE_OPEN, inputs=["a", "b"]),
            dict(name="rc", type="vop", kind=Vop.RANGE_CLOSED, inputs=["a", "b"]),
            dict(name="il", type="vop", kind=Vop.INTERPOLATE_LINEAR, inputs=["a", "b"])]
    g = ml.Graph(eng, V, desc, ["idx", "ro", "rc", "il"])
    got = g.process_host(T, {"a": d["vop_a"], "b": d["vop_b"]}, Layout.QUAD)
    for o, vop in zip(got, (Vop.COLUMN_INDEX, Vop.RANGE_OPEN, Vop.RANGE_CLOSED, Vop.INTERPOLATE_LINEAR)):
        assert_bits_equal(o, d[f"vop{vop}_out"], True, f"vop {vop}")
    # voice-rate (param) operands + the scalar-mix lerp
    V2, T2 = 70, 5
    rng = np.random.default_rng(5)
    start, end = rng.standard_normal(V2).astype(np.float32), rng.standard_normal(V2).astype(np.float32)
    mixc = rng.random((V2, T2)).astype(np.float32)
    desc = [dict(name="x", type="input"), dict(name="s", type="param"), dict(name="e", type="param"), dict(name="m", type="control"),
            dict(name="ramp", type="vop", kind=Vop.RANGE_CLOSED, inputs=["s", "e"]),
            dict(name="y", type="op", kind=Op.LERP, inputs=["x", "ramp", "m"])]
    g = ml.Graph(eng, V2, desc, ["y"])
    g.set_param("s", start)
    g.set_param("e", end)
    x = lcg_noise(np.arange(V2, dtype=np.uint32), 64 * T2)
    (got,) = g.process_host(T2, {"x": x, "m": mixc}, Layout.ROWS)
    ramp = oracle.vop(Vop.RANGE_CLOSED, V2, T2, np.repeat(np.repeat(start[:, None], T2, 1), 64, 1), np.repeat(np.repeat(end[:, None], T2, 1), 64, 1))
    want = oracle.op(Op.LERP, x, ramp, np.repeat(mixc, 64, 1))
    assert_bits_equal(got, want.view(np.float32).reshape(V2, -1), True, "lerp(x, ramp, float m)")


@pytest.mark.gpu
def test_glide_patch_vs_oracle(eng, oracle):
    """A voice driven at control rate: pitch control -> LinearGlide -> exp2Approx -> * base = freq -> SawGen ->
    Lopass(x, omega = Interpolator1(cutoff control), k param) -> * SampleAccurateLinearGlide(level input)."""
    import madronalib_amd as ml
    V, T = 160, 30
    desc = [dict(name="pitch", type="control"), dict(name="cutoff", type="control"), dict(name="level", type="input"),
            dict(name="k", type="param"), dict(name="base", type="const", value=110.0 / 48000.0),
            dict(name="pglide", type="proc", kind=Proc.LINEAR_GLIDE, inputs=["pitch"]),
            dict(name="ratio", type="op", kind=Op.EXP2_APPROX, inputs=["pglide"]),
            dict(name="freq", type="op", kind=Op.MULTIPLY, inputs=["ratio", "base"]),
            dict(name="saw", type="proc", kind=Proc.SAW_GEN, inputs=["freq"]),
            dict(name="omega", type="proc", kind=Proc.INTERPOLATOR1, inputs=["cutoff"]),
            dict(name="lp", type="proc", kind=Proc.LOPASS, inputs=["saw", "omega", "k"]),
            dict(name="lvl", type="proc", kind=Proc.SAMPLE_ACCURATE_LINEAR_GLIDE, inputs=["level"]),
            dict(name="out", type="op", kind=Op.MULTIPLY, inputs=["lp", "lvl"])]
    g = ml.Graph(eng, V, desc, ["out"])
    g.clear()
    rng = np.random.default_rng(3)
    kq = rng.uniform(0.2, 1.5, V).astype(np.float32)
    g.set_param("k", kq)
    co_pg = np.stack([oracle.make_coeffs("linear_glide", 64.0 * (1 + v % 9)) for v in range(V)], 1)
    co_lv = np.stack([oracle.make_coeffs("sample_glide", 10.0 + 7 * (v % 40)) for v in range(V)], 1)
    g.set_coeffs("pglide", [np.ascontiguousarray(r) for r in co_pg])
    g.set_coeffs("lvl", [np.ascontiguousarray(r) for r in co_lv])
    states = {n["name"]: oracle.chain_clear([n["kind"]], V) for n in desc if n["type"] == "proc"}
    coeffs = {"pglide": co_pg, "lvl": co_lv}
    for call in range(2):
        sig = {"pitch": stepped(V, T, 40 + call, -1.0, 3.0, 2, 10), "cutoff": stepped(V, T, 50 + call, 0.005, 0.45, 1, 6),
               "level": stepped(V, 64 * T, 60 + call, 0.0, 1.0, 50, 700)}
        (got,) = g.process_host(T, sig, Layout.QUAD)
        (want,) = evaluate(oracle, desc, ["out"], V, T, sig, {"k": kq}, coeffs, states)
        assert_bits_equal(got, want, True, f"glide patch call {call}")
    for n in desc:
        if n["type"] == "proc":
            for i in range(g.num_state(n["name"])):
                assert (g.get_state(n["name"], i) == states[n["name"]][i]).all(), (n["name"], i)
    assert np.abs(got).max() > 1e-3


@pytest.mark.gpu
def test_vector_rate_rules(eng):
    import madronalib_amd as ml
    with pytest.raises(ml.MlgpuError) as ei:
        eng.bank([Proc.LINEAR_GLIDE], 64)                      # vector-rate processors are graph nodes
    assert ei.value.status == ml.Status.ERR_UNSUPPORTED
    g = ml.Graph(eng, 64)
    a = g.add("a", "input")
    c = g.add("c", "control")
    with pytest.raises(ml.MlgpuError):
        g.add("bad", "proc", Proc.INTERPOLATOR1, [a])          # needs a float per vector, not an audio signal
    with pytest.raises(ml.MlgpuError):
        g.add("bad", "vop", Vop.RANGE_OPEN, [a, c])
    with pytest.raises(ml.MlgpuError):
        g.add("bad", "proc", Proc.LOPASS, [a, c])              # Lopass takes 1 or 3 inputs
    with pytest.raises(ml.MlgpuError):
        g.add("bad", "proc", Proc.HI_SHELF, [a] * 6)           # HiShelf takes 1 or 7
    ok = g.add("glide", "proc", Proc.LINEAR_GLIDE, [c])
    g.add_output(ok)
    g.compile()
    with pytest.raises(ml.MlgpuError):
        g.process(1, [eng.alloc(64 * 64 * 4)], [eng.alloc(64 * 64 * 4)])   # control list missing


@pytest.mark.gpu
def test_online_tuning_same_bits_and_settles(eng, oracle):
    """mlgpu_graph_set_autotune: the first big launches take turns through the kernel forms (voices per lane x quads per
    trip); every launch gives the reference's bits whichever form ran, and the graph settles on one form."""
    import madronalib_amd as ml
    V, T = 16384, 4                   # 4 Mi voice-samples per launch: the smallest launch that is timed
    params, coeffs = synth16_setup(oracle, 64, seed=8)
    idx = np.arange(V) % 64
    P = {k: (v[idx] if np.ndim(v) else v) for k, v in params.items()}
    C = {k: np.ascontiguousarray(c[:, idx]) for k, c in coeffs.items()}
    seeds = (np.arange(V, dtype=np.uint32) % 64) + np.uint32(3)
    desc, outs = patches.synth16()
    g = ml.Graph(eng, V, desc, outs, autotune=True)
    g.clear()
    for k, v in P.items():
        g.set_param(k, v if np.ndim(v) else float(v))
    for k, c in C.items():
        g.set_coeffs(k, [np.ascontiguousarray(r) for r in c])
    g.set_state("noise", 0, seeds)
    states = {n["name"]: oracle.chain_clear([n["kind"]], 64) for n in desc if n["type"] == "proc"}
    states["noise"][0] = seeds[:64]
    gate64 = gate_signal(64, 64 * T * 14, seed=2)
    forms = set()
    for call in range(14):
        sl = slice(call * 64 * T, (call + 1) * 64 * T)
        (got,) = g.process_host(T, {"gate": np.ascontiguousarray(gate64[idx][:, sl])}, Layout.QUAD)
        (want,) = evaluate(oracle, desc, outs, 64, T, {"gate": np.ascontiguousarray(gate64[:, sl])}, params, coeffs, states)
        assert_bits_equal(got[:64], want, True, f"call {call}")
        assert (got.reshape(V // 64, 64, -1).view(np.uint32) == want.view(np.uint32)[None]).all()
        forms.add(g.tuning())
    settled, vl, quads = g.tuning()
    assert settled and vl in (1, 2) and quads in (1, 2)
    assert any(not f[0] for f in forms)    # it did go through a measuring phase


@pytest.mark.gpu
@pytest.mark.parametrize("vpl", [0, 2])
def test_const_vector_nodes_keep_their_bits(eng, oracle, vpl):
    """mlgpu_graph_add_const_vector: DSPVector(const float*) / DSPVector(fn) of MLDSPOps.h:140-161 as a node. Every voice and
    every vector sees the same 64 floats, bit patterns untouched (NaN payloads, infinities, denormals, -0)."""
    import madronalib_amd as ml
    V, T = 300, 5
    rng = np.random.default_rng(77)
    tbl = rng.standard_normal(64).astype(np.float32)
    odd = rng.integers(0, 2**32, 64, dtype=np.uint64).astype(np.uint32)
    odd[:6] = [0x7fc00123, 0xffc00001, 0x7f800000, 0xff800000, 0x00000001, 0x80000000]
    desc = [dict(name="x", type="input"),
            dict(name="win", type="const_vector", value=tbl),
            dict(name="bits", type="const_vector", value=odd.view(np.float32)),
            dict(name="y", type="op", kind=Op.MULTIPLY, inputs=["x", "win"]),
            dict(name="lp", type="proc", kind=Proc.ONE_POLE, inputs=["y"])]
    g = ml.Graph(eng, V, desc, ["lp", "bits", "y"], voices_per_lane=vpl)
    g.clear()
    co = np.full((2, V), 0.25, np.float32)
    co[1] = 0.75
    g.set_coeffs("lp", [co[0], co[1]])
    x = lcg_noise(np.arange(V, dtype=np.uint32) + np.uint32(5), 64 * T)
    got = g.process_host(T, {"x": x}, Layout.QUAD)
    from graph_oracle import evaluate
    states = {"lp": oracle.chain_clear([Proc.ONE_POLE], V)}
    want = evaluate(oracle, desc, ["lp", "bits", "y"], V, T, {"x": x}, {}, {"lp": co}, states)
    for o, w, nm in zip(got, want, ("lp", "bits", "y")):
        assert_bits_equal(o, w, True, nm)
    assert np.array_equal(got[1].view(np.uint32), np.tile(odd, (V, T)))


@pytest.mark.gpu
def test_live_constants_change_between_launches(eng, oracle):
    """mlgpu_graph_set_live_constants / set_const / update_constants_from: constants read from a device table give the same
    bits as literals, can be changed between launches without touching state, and a graph with other wiring is refused."""
    import madronalib_amd as ml
    V, T = 300, 4

    def description(gain, offset):
        return [dict(name="x", type="input"), dict(name="g", type="const", value=gain), dict(name="o", type="const", value=offset),
                dict(name="xg", type="op", kind=Op.MULTIPLY, inputs=["x", "g"]),
                dict(name="lp", type="proc", kind=Proc.ONE_POLE, inputs=["xg"]),
                dict(name="y", type="op", kind=Op.ADD, inputs=["lp", "o"])]
    co = np.stack([np.full(V, 0.25, np.float32), np.full(V, 0.75, np.float32)])
    x = lcg_noise(np.arange(V, dtype=np.uint32) + np.uint32(9), 64 * T * 3)
    seg = [np.ascontiguousarray(x[:, 64 * T * i:64 * T * (i + 1)]) for i in range(3)]
    values = [(0.5, 0.125), (1.75, -0.3), (0.1, 7.0)]
    states = {"lp": oracle.chain_clear([Proc.ONE_POLE], V)}
    want = [evaluate(oracle, description(*v), ["y"], V, T, {"x": s}, {}, {"lp": co}, states)[0] for v, s in zip(values, seg)]

    g = ml.Graph(eng, V, description(*values[0]), ["y"], live_constants=True)
    assert "a.consts[" in g.source
    g.clear()
    g.set_coeffs("lp", [co[0], co[1]])
    (got0,) = g.process_host(T, {"x": seg[0]}, Layout.QUAD)
    g.set_const("g", values[1][0])
    g.set_const("o", values[1][1])
    (got1,) = g.process_host(T, {"x": seg[1]}, Layout.QUAD)
    other = ml.Graph(ml.OfflineEngine(), V, description(*values[2]), ["y"])     # never compiled: only its numbers are read
    g.update_constants_from(other)
    (got2,) = g.process_host(T, {"x": seg[2]}, Layout.QUAD)
    for got, w, nm in zip((got0, got1, got2), want, ("literal values", "set_const", "update_constants_from")):
        assert_bits_equal(got, w, True, nm)

    rewired = description(*values[2])
    rewired[3]["inputs"] = ["x", "o"]
    with pytest.raises(ml.MlgpuError):
        g.update_constants_from(ml.Graph(ml.OfflineEngine(), V, rewired, ["y"]))
    plain = ml.Graph(eng, V, description(*values[0]), ["y"])
    assert "a.consts[" not in plain.source
    with pytest.raises(ml.MlgpuError):
        plain.set_const("g", 2.0)                      # literals of the kernel
    with pytest.raises(ml.MlgpuError):
        plain.update_constants_from(other)             # ... so other numbers are another kernel
    plain.update_constants_from(ml.Graph(ml.OfflineEngine(), V, description(*values[0]), ["y"]))   # the same numbers: nothing to do


@pytest.mark.gpu
@pytest.mark.parametrize("flush", [False, True])
def test_clamp_with_constant_bounds_two_instruction_form(eng, oracle, flush):
    """clamp(x, lo, hi) with literal, non-zero bounds on an arithmetic result is emitted as v_max_f32 + v_min_f32
    (clamp_const_bounds) instead of two compare / select / canonicalize triples. Same bits as the reference's
    min(max(x, lo), hi) for every x an arithmetic instruction can produce - infinities, quiet NaNs, +-0, denormals - in both
    floating-point modes; clamps the proof does not cover (a raw input, a zero bound) keep the general form and match too."""
    import madronalib_amd as ml
    from inputs import general_floats
    V, T = 128, 2
    a = general_floats(V * 64 * T, seed=31).reshape(V, 64 * T)
    b = general_floats(V * 64 * T, seed=32)[::-1].reshape(V, 64 * T).copy()
    a[0, :8] = [np.inf, -np.inf, 0.0, -0.0, 1e-30, -1e-30, 3e38, -3e38]
    b[0, :8] = [0.0, 0.0, -1.0, 1.0, 1e-15, 1e-15, 10.0, 10.0]          # inf*0 = NaN, +-0, denormal products, overflow
    desc = [dict(name="a", type="input"), dict(name="b", type="input"),
            dict(name="lo", type="const", value=-0.75), dict(name="hi", type="const", value=2.5), dict(name="z", type="const", value=0.0),
            dict(name="one", type="const", value=1.0),
            dict(name="m", type="op", kind=Op.MULTIPLY, inputs=["a", "b"]),
            dict(name="fast", type="op", kind=Op.CLAMP, inputs=["m", "lo", "hi"]),        # two-instruction form
            dict(name="raw", type="op", kind=Op.CLAMP, inputs=["a", "lo", "hi"]),         # raw input: general form
            dict(name="zero", type="op", kind=Op.CLAMP, inputs=["m", "z", "one"])]        # zero bound: general form
    outs = ["fast", "raw", "zero"]
    g = ml.Graph(eng, V, desc, outs)
    assert g.source.count("clamp_const_bounds(") == 1
    eng.set_flush_denormals(flush)
    try:
        got = g.process_host(T, {"a": a, "b": b}, Layout.QUAD)
    finally:
        eng.set_flush_denormals(False)
    with oracle.flush_denormals(flush):
        want = evaluate(oracle, desc, outs, V, T, {"a": a, "b": b}, {}, {}, {})
    for i, o in enumerate(outs):
        assert_bits_equal(got[i], want[i], True, f"clamp {o} flush={flush}")
    g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("vpl", [1, 2])
def test_pulse_gen_with_absurd_widths(eng, oracle, vpl):
    """PulseGen with a per-voice frequency and a per-voice width takes, per wavefront, either a fast path that assumes a width
    below 2^30 in magnitude (the shifted phase then fits an int32: one conversion instead of the emulation of cvttps2dq's
    out-of-range result) or the general one. Wavefronts of both kinds, widths as a param, as the processor's coefficient and as
    a signal (never the width-regular path), against the oracle."""
    import madronalib_amd as ml
    V, T = 256, 3
    rng = np.random.default_rng(9)
    freq = (20.0 * (400.0 ** rng.random(V)) / 48000.0).astype(np.float32)
    width = rng.uniform(0.05, 0.95, V).astype(np.float32)
    # voices 0..63 stay regular (a fast wavefront); the others get absurd widths scattered in
    odd = np.array([3.0e9, -5.0e9, 2.0 ** 30, -(2.0 ** 30), 2.0 ** 31, np.inf, -np.inf, np.nan, 1.0e20, -0.0, 0.0, 1.0, 1.5, -0.25, 2.0 ** 29], np.float32)
    width[64::3] = odd[rng.integers(0, len(odd), len(width[64::3]))]
    wsig = np.repeat(width[:, None], 64 * T, 1)
    desc = [dict(name="f", type="param"), dict(name="w", type="param"), dict(name="ws", type="input"),
            dict(name="pp", type="proc", kind=Proc.PULSE_GEN, inputs=["f", "w"]),      # width per voice: the regular-width test applies
            dict(name="pc", type="proc", kind=Proc.PULSE_GEN, inputs=["f"]),           # width = the processor's coefficient
            dict(name="ps", type="proc", kind=Proc.PULSE_GEN, inputs=["f", "ws"])]     # width a signal
    outs = ["pp", "pc", "ps"]
    g = ml.Graph(eng, V, desc, outs, voices_per_lane=vpl)
    assert g.source.count("pulse_width_is_odd(") == 2 * vpl and ".next_uw(" in g.source
    g.set_param("f", freq)
    g.set_param("w", width)
    g.set_coeffs("pc", [width])
    phases = rng.integers(0, 2 ** 32, V, dtype=np.uint64).astype(np.uint32)
    for nm in outs:
        g.set_state(nm, 0, phases)
    states = {nm: np.ascontiguousarray(phases[None, :].copy()) for nm in outs}
    for call in range(2):
        got = g.process_host(T, {"ws": wsig}, Layout.QUAD)
        want = evaluate(oracle, desc, outs, V, T, {"ws": wsig}, {"f": freq, "w": width}, {"pc": width[None, :]}, states)
        for i, o in enumerate(outs):
            assert_bits_equal(got[i], want[i], True, f"pulse {o} call {call} vpl={vpl}")
    g.close()


HOSTILE_MULTI = ("pulse2", "interp1", "linear_glide", "linear_glide_long", "tempo_lock")


def hostile_multi_case(mk, name, V, T, seed):
    """A multi-input case with 4 % of every input replaced by special values and raw bit patterns (and whole stretches on some
    voices): frequencies and widths of a PulseGen, the targets of glides and interpolators, a TempoLock's phasor and ratios."""
    from inputs import general_floats
    case = multi_case(mk, name, V, T, seed=seed)
    rng = np.random.default_rng(seed + 77)
    ins = []
    for rate, a in case["inputs"]:
        a = a.copy()
        g = general_floats(a.size + (-a.size) % 64, seed + len(ins))[:a.size].reshape(a.shape)
        mask = rng.random(a.shape) < 0.04
        a[mask] = g[mask]
        w = min(24, a.shape[1] // 3)
        a[2::9, a.shape[1] // 2:a.shape[1] // 2 + w] = g[2::9, :w]
        ins.append((rate, np.ascontiguousarray(a)))
    case["inputs"] = ins
    return case


@pytest.mark.gpu
@pytest.mark.parametrize("name", HOSTILE_MULTI)
def test_multi_input_forms_hostile_inputs(eng, oracle, name):
    """The multi-input / vector-rate forms with infinities, NaNs, denormals, huge values and raw bit patterns in every input
    (tests/test_oracle_vs_ref.py pins the oracle on the same cases against the reference's objects)."""
    V, T = 200, 12
    case = hostile_multi_case(oracle, name, V, 2 * T, seed=33)
    g, names = single_node_graph(eng, V, case)
    st = oracle.chain_clear([case["kind"]], V)
    outs, states = run_case_calls(g, names, case, T, st.copy(), Layout.QUAD)
    ins = multi_inputs_audio(case, 2 * T)
    for call in range(2):
        sl = slice(call * 64 * T, (call + 1) * 64 * T)
        want = oracle.proc_multi(case["kind"], T, case["coeffs"], st, [np.ascontiguousarray(x[:, sl]) for x in ins])
        assert_bits_equal(outs[call], want, True, f"hostile {name} call {call}")
        g32, w32 = states[call].view(np.uint32), st.view(np.uint32)
        bothnan = np.isnan(g32.view(np.float32)) & np.isnan(w32.view(np.float32))
        assert ((g32 == w32) | bothnan).all(), f"hostile {name} state after call {call}"


@pytest.mark.gpu
@pytest.mark.parametrize("group", [2, 4, 8, 16])
@pytest.mark.parametrize("vpl", [1, 2, -1])
def test_voice_sum_inside_the_graph_kernel(eng, oracle, group, vpl):
    """mlgpu_graph_set_output_group_sum: an output that is the sum of groups of adjacent voices in the order Synth::processVector
    adds them ((0 + v0) + v1 + ...), made inside the voice kernel - with lane shifts, or for groups of 16 at one voice per lane
    through an LDS strip per wavefront - the bits mlgpu_mixdown_groups gives for the same voices, next to an ordinary output of
    the same graph. Values of mixed sign and magnitude so that the order matters. vpl -1: one voice per lane and a bank that ends
    in a partial wavefront (37 groups of 16)."""
    import madronalib_amd as ml
    V, T = 512 + 2 * 16 * vpl * 8, 3          # not a whole number of workgroups
    if vpl < 0:
        V, vpl = 16 * 37, 1
    rng = np.random.default_rng(group)
    x = (rng.standard_normal((V, 64 * T)) * 10.0 ** rng.integers(-3, 4, (V, 1))).astype(np.float32)
    x[3, :8] = [np.inf, -np.inf, np.nan, -0.0, 0.0, 1e-40, 3e38, -3e38]
    desc = [dict(name="x", type="input"), dict(name="k", type="const", value=0.75), dict(name="y", type="op", kind=Op.MULTIPLY, inputs=["x", "k"])]
    g = ml.Graph(eng, V, desc, ["y", "y"], voices_per_lane=vpl, output_groups={1: group})
    assert ("group16_sum_store" if (group == 16 and vpl == 1) else f"group_sum_in_order<{group}>") in g.source
    n = V * T * 64
    d_x = eng.to_device(x)
    d_q = eng.alloc(4 * n)
    eng.layout_convert(d_x, Layout.VOICE_MAJOR, d_q, Layout.QUAD, V, T)
    d_all, d_sum, d_ref = eng.alloc(4 * n), eng.alloc(4 * n // group), eng.alloc(4 * n // group)
    g.process(T, [d_q], [d_all, d_sum], out_layout=Layout.VOICE_MAJOR)
    eng.mixdown_groups(d_all, Layout.VOICE_MAJOR, V // group, group, T, d_ref, Layout.VOICE_MAJOR)
    got = d_sum.download(np.float32, n // group)
    want = d_ref.download(np.float32, n // group)
    assert_bits_equal(got, want, True, f"voice sum of {group}, {vpl} voices per lane")
    # and the host's own sequential sum (float32, voice by voice)
    y = (x * np.float32(0.75)).reshape(V // group, group, 64 * T)
    acc = np.zeros((V // group, 64 * T), np.float32)
    with np.errstate(all="ignore"):
        for p in range(group):
            acc = acc + y[:, p]
    assert_bits_equal(got.reshape(V // group, 64 * T), acc, True, "voice sum vs numpy")
    g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("group,vpl", [(1, 0), (4, 0), (16, 0), (3, 0), (4, 2), (192, 0)])
def test_input_shared_by_groups_of_voices(eng, group, vpl):
    """mlgpu_graph_set_input_group: an input with one row per `group` adjacent voices (a controller or transport signal per
    instrument); voice v reads row v // group. QUAD and VOICE_MAJOR, one and two voices per lane, a group that is not a power
    of two, one row for the whole bank."""
    import madronalib_amd as ml
    V, T = 192, 5
    rng = np.random.default_rng(group)
    x = rng.standard_normal((V, 64 * T)).astype(np.float32)
    c = rng.standard_normal((V // group, 64 * T)).astype(np.float32)
    desc = [dict(name="x", type="input"), dict(name="c", type="input"), dict(name="y", type="op", kind=Op.MULTIPLY, inputs=["x", "c"])]
    want = x * np.repeat(c, group, axis=0)
    for layout in (Layout.QUAD, Layout.VOICE_MAJOR):
        g = ml.Graph(eng, V, desc, ["y"], voices_per_lane=vpl, input_groups={1: group})
        def dev(a):
            rows = a.shape[0]
            if layout == Layout.QUAD:
                a = a.reshape(rows, 16 * T, 4).transpose(1, 0, 2)
            return eng.to_device(np.ascontiguousarray(a))
        d_y = eng.alloc(4 * V * T * 64)
        g.process(T, [dev(x), dev(c)], [d_y], in_layout=layout, out_layout=Layout.VOICE_MAJOR)
        assert_bits_equal(d_y.download(np.float32, V * T * 64).reshape(V, 64 * T), want, True, f"group {group} layout {layout}")
        g.close()
    g = ml.Graph(eng, V, desc, ["y"])
    g_err = ml.Graph(eng, V, None)
    for n in desc[:2]:
        g_err.add(**n)
    for bad in (0, 5, V + 1):
        with pytest.raises(ml.MlgpuError):
            g_err.set_input_group(1, bad)
    with pytest.raises(ml.MlgpuError):
        g_err.set_input_group(2, 4)
    with pytest.raises(ml.MlgpuError):
        g.set_input_group(1, 4)                    # already compiled
    g.close()
    g_err.close()


@pytest.mark.gpu
@pytest.mark.parametrize("hostile", [False, True])
def test_tempo_lock_following_a_computed_phasor(eng, oracle, hostile):
    """TempoLock whose input is a signal computed inside the graph (here the streamed clock times a constant 1: the same values, but
    not an input node): taken sample by sample - only a start-up vector needs x[1], and first for its sample 1 - with the results
    of the vector-rate form, i.e. of the reference (the oracle's TempoLock is pinned against it on these cases). Stopped clocks,
    restarts, locking and non-locking ratios; and with infinities, NaNs and raw bit patterns in the clock."""
    import madronalib_amd as ml
    V, T = 200, 16
    case = (hostile_multi_case if hostile else multi_case)(oracle, "tempo_lock", V, 2 * T, seed=77)
    g = ml.Graph(eng, V)
    names = []
    for i, (rate, _) in enumerate(case["inputs"]):
        names.append(f"in{i}")
        g.add(names[-1], "input" if rate == "audio" else "control")
    g.add("one", "const", value=1.0)
    g.add("clock", "op", Op.MULTIPLY, [names[0], "one"])
    g.add("p", "proc", case["kind"], ["clock"] + names[1:])
    g.add_output("p")
    g.compile()
    assert ".next_x(" in g.source
    st = oracle.chain_clear([case["kind"]], V)
    outs, states = run_case_calls(g, names, case, T, st.copy(), Layout.QUAD)
    ins = multi_inputs_audio(case, 2 * T)
    for call in range(2):
        sl = slice(call * 64 * T, (call + 1) * 64 * T)
        want = oracle.proc_multi(case["kind"], T, case["coeffs"], st, [np.ascontiguousarray(x[:, sl]) for x in ins])
        assert_bits_equal(outs[call], want, True, f"computed-input TempoLock call {call} (hostile: {hostile})")
        g32, w32 = states[call].view(np.uint32), st.view(np.uint32)
        bothnan = np.isnan(g32.view(np.float32)) & np.isnan(w32.view(np.float32))
        assert ((g32 == w32) | bothnan).all(), f"computed-input TempoLock state after call {call}"


@pytest.mark.gpu
@pytest.mark.parametrize("trip_quads,vpl,unlock", [(2, 1, False), (2, 2, False), (1, 1, False), (4, 1, False), (0, 1, False), (2, 1, True), (2, 2, True)])
def test_oscillator_trips_on_the_knife_edges(eng, oracle, trip_quads, vpl, unlock, monkeypatch):
    """SawGen / PulseGen nodes with per-voice frequency (and width) get their polyBLEP corrections once per zone per trip of
    `trip_quads` quads (mldsp_procs.hpp: trip_u; MLGPU_GRAPH_OSC_TRIP, default 2; 0 = per sample). The short cut rests on "at most one
    sample of a trip in each zone", which rounding can break next to a wrap - those trips must be recognised and evaluated per sample.
    Voices placed on the edges: phases landing 0..23 units after a wrap, 0..368 units before one, and within -128 .. +608 units of the pulse's
    falling step, some 1..37 samples into the launch; widths 0, 1, dt, 1 - dt, dt / 2 among random ones; frequencies at and around
    the trip forms' limit (1 / 2N), wavefronts entirely below it and mixed ones. Outputs and final phases against the oracle.
    Round 4: the SawGen and the first PulseGen sit on one frequency node, and with equal phase counters (all of them here, unless
    `unlock`) a wavefront makes the two trips as one (mldsp_procs.hpp: trip_locked). `unlock`: the PulseGen's counters differ from
    the SawGen's - by one unit in a single lane of some wavefronts, altogether in a quarter of the voices - so those wavefronts
    make their trips one by one, next to wavefronts that stay locked."""
    import madronalib_amd as ml
    monkeypatch.setenv("MLGPU_GRAPH_OSC_TRIP", str(trip_quads))
    V, T = 16384, 2
    rng = np.random.default_rng(77)
    freq = (1e-4 * (300.0 ** rng.random(V))).astype(np.float32)              # 1e-4 .. 0.03: inside every trip form's range
    freq[V // 2:] = (1e-3 * (200.0 ** rng.random(V - V // 2))).astype(np.float32)   # second half: up to 0.2 - mixed wavefronts
    limit = np.float32(0.5 / (4 * trip_quads)) if trip_quads else np.float32(1.0 / 32.0)
    freq[5:V // 2:64] = limit                                   # the first half's wavefronts stay entirely inside the trip form's range
    freq[V // 2 + 5::64] = np.float32(1.0 / 32.0)
    freq[V // 2 + 6::64] = np.float32(1.0 / 16.0)
    freq[V // 2 + 7::64] = np.float32(1.0 / 8.0)
    freq[V // 2 + 9::128] = np.nextafter(limit, np.float32(1.0))
    width = rng.uniform(0.0, 1.0, V).astype(np.float32)
    width[0::16] = 0.0
    width[1::16] = 1.0
    width[2::16] = freq[2::16]
    width[3::16] = np.float32(1.0) - freq[3::16]
    width[4::16] = freq[4::16] * np.float32(0.5)
    istep = np.rint(freq.astype(np.float64) * 2.0 ** 32).astype(np.uint64)
    v = np.arange(V, dtype=np.uint64)
    k, j = v % 37 + 1, (v // 8) % 24
    om = rng.integers(0, 2 ** 32, V, dtype=np.uint64)
    # (a trip with a suspect lane is evaluated per sample for the whole wavefront: the three kinds of edges go to different wavefronts,
    # or one kind's fall-back would hide what the others' recognition misses)
    wave = (v // 64) % 3
    om = np.where((wave == 0) & (v % 4 == 1), (2 ** 32 * 64 - k * istep + j), om)
    om = np.where((wave == 0) & (v % 4 == 3), (2 ** 32 * 64 - k * istep - 16 * j), om)
    wq = np.rint(width.astype(np.float64) * 2.0 ** 32).astype(np.uint64)
    om = np.where((wave == 1) & (v % 2 == 1), (2 ** 32 * 64 + wq - k * istep + 32 * j - 128), om)
    phases = (om % (2 ** 32)).astype(np.uint32)
    desc = [dict(name="f", type="param"), dict(name="w", type="param"),
            dict(name="saw", type="proc", kind=Proc.SAW_GEN, inputs=["f"]),
            dict(name="pw", type="proc", kind=Proc.PULSE_GEN, inputs=["f", "w"]),
            dict(name="pc", type="proc", kind=Proc.PULSE_GEN, inputs=["f"])]
    outs = ["saw", "pw", "pc"]
    g = ml.Graph(eng, V, desc, outs, voices_per_lane=vpl)
    if trip_quads:
        assert g.source.count(f".trip_u<{4 * trip_quads}>(") == 3 * vpl and f"q2 += {trip_quads}" in g.source
    else:
        assert ".trip_u<" not in g.source and ".next_u(" in g.source
    g.set_param("f", freq)
    g.set_param("w", width)
    g.set_coeffs("pc", [width])
    start = {nm: phases.copy() for nm in outs}
    if trip_quads:
        if os.environ.get("MLGPU_GRAPH_LOCK_OSC", "1") != "0":   # (the developer knob that turns the pairing off: the rest still has to hold)
            assert ("trip_locked<" in g.source) and "locked2" in g.source     # node 2 (the saw) is paired with node 3
    if unlock:
        start["pw"][64 * 5 + 3::64 * 7] ^= np.uint32(1)
        start["pw"][V // 4:V // 2] = rng.integers(0, 2 ** 32, V // 2 - V // 4, dtype=np.uint64).astype(np.uint32)
    for nm in outs:
        g.set_state(nm, 0, start[nm])
    states = {nm: np.ascontiguousarray(start[nm][None, :].copy()) for nm in outs}
    for call in range(2):
        got = g.process_host(T, {}, Layout.QUAD)
        want = evaluate(oracle, desc, outs, V, T, {}, {"f": freq, "w": width}, {"pc": width[None, :]}, states)
        for i, o in enumerate(outs):
            assert_bits_equal(got[i], want[i], True, f"{o} call {call} trip={trip_quads} vpl={vpl}")
    for nm in outs:
        assert (g.get_state(nm, 0) == states[nm][0]).all(), nm
    # the corrections did happen: a naive saw differs from the output on the samples next to a step
    naive = 2.0 * ((phases[:, None].astype(np.float64) + np.arange(1, 64 * T + 1)[None, :] * istep[:, None].astype(np.float64)) % 2 ** 32) / 2 ** 32 - 1.0
    assert (np.abs(want[0][:, :64 * T].astype(np.float64) - naive) > 1e-3).mean() > 0.005
    g.close()


@pytest.mark.gpu
def test_oscillator_trips_equal_the_per_sample_form_on_free_running_phases(eng, monkeypatch):
    """tools/osc_trip_soak.py in small: the same SawGen / PulseGen graph generated with the corrections once per zone per trip (the default)
    and per sample (MLGPU_GRAPH_OSC_TRIP=0), random phases, frequencies and widths, two launches with carried state: the same bits,
    outputs and final phases. (The per-sample form is what the oracle tests pin; the soak tool runs 6e9 samples per oscillator.)"""
    import madronalib_amd as ml
    V, T = 8192, 16
    rng = np.random.default_rng(5)
    freq = (2e-5 * ((0.0625 / 2e-5) ** rng.random(V))).astype(np.float32)
    width = rng.uniform(0.0, 1.0, V).astype(np.float32)
    phases = rng.integers(0, 2 ** 32, V, dtype=np.uint64).astype(np.uint32)
    desc = [dict(name="f", type="param"), dict(name="w", type="param"),
            dict(name="saw", type="proc", kind=Proc.SAW_GEN, inputs=["f"]),
            dict(name="pw", type="proc", kind=Proc.PULSE_GEN, inputs=["f", "w"])]
    res = {}
    for trip in ("0", None):
        if trip is None:
            monkeypatch.delenv("MLGPU_GRAPH_OSC_TRIP", raising=False)
        else:
            monkeypatch.setenv("MLGPU_GRAPH_OSC_TRIP", trip)
        g = ml.Graph(eng, V, desc, ["saw", "pw"])
        assert (".trip_u<8>(" in g.source) == (trip is None)
        g.set_param("f", freq)
        g.set_param("w", width)
        for nm in ("saw", "pw"):
            g.set_state(nm, 0, phases)
        a = g.process_host(T, {}, Layout.QUAD)
        b = g.process_host(T, {}, Layout.QUAD)
        res[trip] = a + b + [g.get_state("saw", 0).view(np.float32), g.get_state("pw", 0).view(np.float32)]
        g.close()
    for i, (x, y) in enumerate(zip(res[None], res["0"])):
        assert_bits_equal(x, y, False, f"trip form against per-sample form, item {i}")
    assert np.abs(res[None][0]).max() > 0.9 and np.abs(res[None][1]).max() >= 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("V", [320, 1000, 4096 + 77, 65536 + 4096 + 192])      # whole wavefronts and not; one to three row passes
def test_graph_output_mixdown_same_bits(eng, V):
    """mlgpu_graph_set_output_mixdown: config 5's 16-node voice with its output turned into the mixdown of all voices - the tree over
    each wavefront made inside the voice kernel, the voices' signal never written - against the same graph's voices put through
    mlgpu_mixdown (whose order the oracle pins in test_mixdown_vs_oracle): the same bits, three launches, and the same state after."""
    import madronalib_amd as ml
    from madronalib_amd.sharding import cfg5_gate_quad, cfg5_voice_params
    T, L = 2, 3
    desc, outs = patches.synth16()
    params, coeffs, seeds = cfg5_voice_params(0, V, V, ml)
    eng.mixdown_reserve(V, T)
    graphs = []
    for mixed in (False, True):
        g = ml.Graph(eng, V, desc, outs, compile_now=False)
        if mixed:
            g.set_output_mixdown(0)
        g.compile()
        g.clear()
        for k, v in params.items():
            g.set_param(k, v if np.ndim(v) else float(v))
        for k, c in coeffs.items():
            g.set_coeffs(k, [np.ascontiguousarray(r) for r in c])
        g.set_state("noise", 0, seeds)
        graphs.append(g)
    d_gate = eng.to_device(cfg5_gate_quad(0, V, T))
    d_voices, d_two, d_one = eng.alloc(4 * V * T * 64), eng.alloc(4 * T * 64), eng.alloc(4 * T * 64)
    for launch in range(L):
        graphs[0].process(T, [d_gate], [d_voices])
        eng.mixdown(d_voices, Layout.QUAD, V, T, d_two)
        graphs[1].process(T, [d_gate], [d_one])
        two, one = d_two.download(np.float32, 64 * T), d_one.download(np.float32, 64 * T)
        assert_bits_equal(one, two, True, f"graph output mixdown, launch {launch}")
        assert np.isfinite(two).all() and np.abs(two).max() > 1e-4
    for name in ("saw", "pulse", "lfo", "noise", "lp", "hp", "smooth", "dc", "env"):
        i = 0
        while True:
            try:
                a = graphs[0].get_state(name, i)
            except ml.MlgpuError:
                break
            assert (a == graphs[1].get_state(name, i)).all(), (name, i)
            i += 1
        assert i > 0, name
    for g in graphs:
        g.close()


@pytest.mark.gpu
def test_graph_output_mixdown_next_to_a_plain_output_and_delay_rings(eng):
    """Two outputs of one graph, one mixed down and one per voice, on a graph with a delay ring in layout 2 and a voice count that
    fills neither a workgroup nor a wavefront: the mixed one equals mlgpu_mixdown of what the same graph gives per voice, the plain one
    is untouched. And the rules: not with a group sum on the same output, scratch reserved for every mixed output."""
    import madronalib_amd as ml
    V, T = 777, 3
    x = lcg_noise(np.arange(V, dtype=np.uint32) + 5, 64 * T)
    outs = {}
    for mixed in (False, True):
        g = ml.Graph(eng, V, delay_windows=2)
        g.add("x", "input")
        g.add("d", "proc", Proc.INTEGER_DELAY, ["x"], max_delay=200.0)
        g.add("y", "op", Op.ADD, ["x", "d"])
        g.add_output("y")
        g.add_output("d")
        if mixed:
            g.set_output_mixdown(0)
            with pytest.raises(ml.MlgpuError):
                g.set_output_group_sum(0, 4)
        g.compile()
        g.set_state("d", 1, ((np.arange(V) * 13) % 190).astype(np.uint32))
        d_x = eng.to_device(x)
        d_y = eng.alloc(4 * V * T * 64)
        d_d = eng.alloc(4 * V * T * 64)
        if mixed:
            small = ml.Engine(0)
            g2 = ml.Graph(small, 64)
            g2.add("x", "input")
            g2.add_output("x")
            g2.set_output_mixdown(0)
            g2.compile()
            with pytest.raises(ml.MlgpuError) as ei:       # nothing reserved on that engine
                g2.process(1, [small.alloc(4 * 64 * 64)], [small.alloc(4 * 64)])
            assert "reserve_mixdown" in str(ei.value)
            g2.reserve_mixdown(1)
            g2.process(1, [small.alloc(4 * 64 * 64)], [small.alloc(4 * 64)])
            small.close()
            g.reserve_mixdown(T)
        else:
            eng.mixdown_reserve(V, T)
        g.process(T, [d_x], [d_y, d_d], Layout.VOICE_MAJOR, Layout.VOICE_MAJOR)
        if mixed:
            outs["mix"] = d_y.download(np.float32, 64 * T).copy()
            outs["d_mixed_graph"] = d_d.download(np.float32, V * T * 64).copy()
        else:
            d_m = eng.alloc(4 * 64 * T)
            eng.mixdown(d_y, Layout.VOICE_MAJOR, V, T, d_m)
            outs["two"] = d_m.download(np.float32, 64 * T).copy()
            outs["d_plain_graph"] = d_d.download(np.float32, V * T * 64).copy()
        g.close()
    assert_bits_equal(outs["mix"], outs["two"], True, "mixed output")
    assert_bits_equal(outs["d_mixed_graph"], outs["d_plain_graph"], True, "the other output")
    assert np.abs(outs["two"]).max() > 0.1


@pytest.mark.gpu
@pytest.mark.parametrize("V", [256, 777])
def test_early_ring_reads_next_to_a_mixdown_output(eng, V):
    """Ring layout 0 with the reads of a sample issued together (graph.hip: earlyRows - LDS landing slots per wavefront) in one kernel
    with an output that is the mixdown of all voices (its LDS strips) and a plain one: four delay nodes of all kinds; the mixed output
    equals mlgpu_mixdown of what the same graph gives per voice, the plain one is untouched - whole and ragged wavefronts."""
    import madronalib_amd as ml
    T = 3
    x = lcg_noise(np.arange(V, dtype=np.uint32) + 15, 64 * T)
    dt = np.repeat((np.arange(V, dtype=np.float32) * np.float32(1.37) % np.float32(180.0))[:, None], 64 * T, 1).astype(np.float32)
    outs = {}
    for mixed in (False, True):
        g = ml.Graph(eng, V, delay_windows=0)
        g.add("x", "input")
        g.add("dt", "input")
        g.add("half", "const", value=0.5)
        g.add("d0", "proc", Proc.INTEGER_DELAY, ["x"], max_delay=200.0)
        g.add("s0", "op", Op.ADD, ["x", "d0"])
        g.add("d1", "proc", Proc.FRACTIONAL_DELAY, ["s0", "dt"], max_delay=200.0)
        g.add("s1", "op", Op.MULTIPLY, ["d1", "half"])
        g.add("d2", "proc", Proc.PITCHBENDABLE_DELAY, ["s1", "dt"], max_delay=200.0)
        g.add("s2", "op", Op.ADD, ["d2", "x"])
        g.add("d3", "proc", Proc.INTEGER_DELAY, ["s2"], max_delay=40.0)
        g.add("y", "op", Op.MULTIPLY, ["d3", "half"])
        g.add_output("y")
        g.add_output("d2")
        if mixed:
            g.set_output_mixdown(0)
        g.compile()
        assert "ldsEarly" in g.source and ("ldsMix" in g.source) == mixed
        g.set_state("d0", 1, ((np.arange(V) * 13) % 190).astype(np.uint32))
        g.set_state("d3", 1, ((np.arange(V) * 7) % 40).astype(np.uint32))
        d_x, d_t = eng.to_device(x), eng.to_device(dt)
        d_y = eng.alloc(4 * V * T * 64)
        d_d = eng.alloc(4 * V * T * 64)
        if mixed:
            g.reserve_mixdown(T)
        else:
            eng.mixdown_reserve(V, T)
        for launch in range(2):
            g.process(T, [d_x, d_t], [d_y, d_d], Layout.VOICE_MAJOR, Layout.VOICE_MAJOR)
        if mixed:
            outs["mix"] = d_y.download(np.float32, 64 * T).copy()
            outs["d_mixed_graph"] = d_d.download(np.float32, V * T * 64).copy()
        else:
            d_m = eng.alloc(4 * 64 * T)
            eng.mixdown(d_y, Layout.VOICE_MAJOR, V, T, d_m)
            outs["two"] = d_m.download(np.float32, 64 * T).copy()
            outs["d_plain_graph"] = d_d.download(np.float32, V * T * 64).copy()
        g.close()
    assert np.abs(outs["two"]).max() > 0
    assert_bits_equal(outs["mix"], outs["two"], True, "mixed output")
    assert_bits_equal(outs["d_mixed_graph"], outs["d_plain_graph"], True, "the plain output of the graph with a mixed one")
