"""GPU parity of the delay lines (IntegerDelay, Allpass1, FractionalDelay, PitchbendableDelay) and of the composites
built on one-vector feedback (Allpass<>, FDN<>, FeedbackDelayFunction): bit-exact against the CPU oracle and against
golden outputs of the reference's own classes (tests/golden/delays.npz)."""
import os

import numpy as np
import pytest

from graph_oracle import evaluate_stream, new_stream_state
from inputs import DELAY_CASES, assert_bits_equal, delay_case, lcg_noise, stepped
from madronalib_amd import patches
from madronalib_amd.constants import Layout, Op, Proc

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "delays.npz"))


@pytest.fixture(scope="module")
def eng():
    import madronalib_amd as ml
    e = ml.Engine(0)
    yield e
    e.close()


# mlgpu_graph_set_delay_layout 0 / 1 / 2 (layout 2: transposed 64-byte pieces on a wave-uniform clock)
WINDOWS = [pytest.param(False, id="rows"), pytest.param(True, id="windows"), pytest.param(2, id="transposed"), pytest.param(4, id="sectors")]


def _voices(V, windows):
    return V     # (layout 2 took whole wavefronts only until the spare lanes of a bank's last wavefront learned to run its last voice again)


def delay_graph(eng, V, kind, n_inputs, max_delay, windows=False):
    import madronalib_amd as ml
    g = ml.Graph(eng, V, delay_windows=windows)
    names = [f"in{i}" for i in range(n_inputs)]
    for nm in names:
        g.add(nm, "input")
    g.add("d", "proc", kind, names, max_delay=max_delay)
    g.add_output("d")
    g.compile()
    return g, names


def run_delay(g, names, state0, inputs, T, layout):
    for i in range(state0.shape[0]):
        g.set_state("d", i, state0[i])
    outs, states = [], []
    for call in range(2):
        sl = slice(call * 64 * T, (call + 1) * 64 * T)
        (y,) = g.process_host(T, {nm: np.ascontiguousarray(a[:, sl]) for nm, a in zip(names, inputs)}, layout)
        outs.append(y)
        states.append(np.stack([g.get_state("d", i) for i in range(state0.shape[0])]))
    return outs, states


@pytest.mark.gpu
@pytest.mark.parametrize("windows", WINDOWS)
@pytest.mark.parametrize("name", DELAY_CASES)
def test_delay_lines_vs_oracle_and_golden(eng, oracle, name, windows):
    from graph_oracle import ring_len
    # golden (reference objects)
    kind, max_delay = int(GOLD[name + "_kind"]), float(GOLD[name + "_max_delay"])
    ins = []
    while f"{name}_in{len(ins)}" in GOLD.files:
        ins.append(GOLD[f"{name}_in{len(ins)}"])
    V, T = ins[0].shape[0], ins[0].shape[1] // 128
    if True:     # (the golden case has a handful of voices: in layout 2 one wavefront, most of its lanes spare)
        g, names = delay_graph(eng, V, kind, len(ins), max_delay, windows)
        outs, states = run_delay(g, names, GOLD[name + "_state0"], ins, T, Layout.VOICE_MAJOR)
        for call in range(2):
            assert_bits_equal(outs[call], GOLD[f"{name}_out{call + 1}"], True, f"{name} golden out{call + 1}")
            assert_bits_equal(states[call], GOLD[f"{name}_state{call + 1}"], False, f"{name} golden state{call + 1}")
    else:
        # the golden case (a handful of voices) in the first lanes of a whole wavefront
        Vp = _voices(V, windows)
        pad = lambda a: np.concatenate([a, np.repeat(a[-1:], Vp - V, 0)], 0)
        g, names = delay_graph(eng, Vp, kind, len(ins), max_delay, windows)
        outs, states = run_delay(g, names, np.concatenate([GOLD[name + "_state0"], np.repeat(GOLD[name + "_state0"][:, -1:], Vp - V, 1)], 1), [pad(a) for a in ins], T, Layout.VOICE_MAJOR)
        for call in range(2):
            assert_bits_equal(outs[call][:V], GOLD[f"{name}_out{call + 1}"], True, f"{name} golden out{call + 1}")
            assert_bits_equal(states[call][:, :V], GOLD[f"{name}_state{call + 1}"], False, f"{name} golden state{call + 1}")
    # oracle, more voices
    V, T = _voices(300, windows), 9
    c = delay_case(oracle, name, V, 2 * T, seed=8)
    g, names = delay_graph(eng, V, c["kind"], len(c["inputs"]), c["max_delay"], windows)
    outs, states = run_delay(g, names, c["state0"], c["inputs"], T, Layout.QUAD)
    rings = 2 if c["kind"] == Proc.PITCHBENDABLE_DELAY else 1
    st, mem = c["state0"].copy(), np.zeros((V, rings, ring_len(c["max_delay"])), np.float32)
    for call in range(2):
        sl = slice(call * 64 * T, (call + 1) * 64 * T)
        want = oracle.delay_process(c["kind"], T, st, mem, [np.ascontiguousarray(a[:, sl]) for a in c["inputs"]])
        assert_bits_equal(outs[call], want, True, f"{name} call {call}")
        assert_bits_equal(states[call], st, False, f"{name} state after call {call}")
    # clear(): the ring is zeroed, write index and delay time stay
    g.clear()
    (y,) = g.process_host(1, {nm: np.zeros((V, 64), np.float32) for nm in names} | ({names[1]: np.full((V, 64), 70.0, np.float32)} if len(names) > 1 else {}), Layout.QUAD)
    assert (y == 0).all()


def _composite(eng, oracle, desc, outs, V, T, sig, params, coeffs, state_edit, windows=False):
    """Build the graph, run two launches, compare with the streaming oracle; returns the GPU outputs joined."""
    import madronalib_amd as ml
    g = ml.Graph(eng, V, desc, outs, delay_windows=windows)
    for k, v in params.items():
        g.set_param(k, v if np.ndim(v) else float(v))
    for k, c in coeffs.items():
        g.set_coeffs(k, [np.ascontiguousarray(r) for r in c])
    st = new_stream_state(oracle, desc, V)
    state_edit(st)
    for n in desc:
        if n["type"] == "proc":
            for i in range(st[n["name"]].shape[0]):
                g.set_state(n["name"], i, st[n["name"]][i])
    got = [[] for _ in outs]
    want = [[] for _ in outs]
    half = T // 2
    for call in range(2):
        part = {k: np.ascontiguousarray(v[:, call * 64 * half:(call + 1) * 64 * half]) for k, v in sig.items()}
        ys = g.process_host(half, part, Layout.QUAD)
        ws = evaluate_stream(oracle, desc, outs, V, half, part, params, coeffs, st)
        for i in range(len(outs)):
            got[i].append(ys[i]), want[i].append(ws[i])
    got = [np.concatenate(x, 1) for x in got]
    want = [np.concatenate(x, 1) for x in want]
    for i, o in enumerate(outs):
        assert_bits_equal(got[i], want[i], True, f"composite output {o}")
    return got


@pytest.mark.gpu
@pytest.mark.parametrize("windows", WINDOWS)
@pytest.mark.parametrize("which,kind,d", [(0, Proc.INTEGER_DELAY, 101.0), (1, Proc.FRACTIONAL_DELAY, 77.37), (2, Proc.PITCHBENDABLE_DELAY, 0.0)])
def test_allpass_composites(eng, oracle, which, kind, d, windows):
    V, T = _voices(70, windows), 30
    x = np.repeat(GOLD["comp_x"][None, :], V, 0).copy()
    x[1:] = lcg_noise(np.arange(V - 1, dtype=np.uint32) + 40, 64 * T)
    dsig = np.repeat(GOLD["comp_dsig"][None, :], V, 0).copy()
    dsig[1:] += np.linspace(0, 100, V - 1, dtype=np.float32)[:, None]
    desc = [dict(name="x", type="input")] + ([dict(name="dl", type="input")] if which == 2 else [])
    sub, out = patches.allpass("ap_", "x", kind, 400.0, "dl" if which == 2 else None)
    desc += sub
    gains = np.full(V, 0.7, np.float32)
    gains[1:] = np.linspace(-0.9, 0.9, V - 1)

    def edit(st):
        if which == 0:
            st["ap_delay"][1] = np.uint32(int(d - 64))
            st["ap_delay"][1, 1:] = (np.arange(V - 1, dtype=np.uint32) * 4) % 300      # (inside the ring's valid range, 0 .. length - 64)
        if which == 1:
            st["ap_delay"][3:5, :] = oracle.fractional_delay_state(float(np.float32(d) - np.float32(64.0))).view(np.uint32)[:, None]
    sig = {"x": x, "dl": dsig} if which == 2 else {"x": x}
    (got,) = _composite(eng, oracle, desc, [out], V, T, sig, {"ap_gain": gains}, {}, edit, windows)
    assert_bits_equal(got[0], GOLD[f"allpass{which}"], True, f"Allpass<{kind}> voice 0 vs the reference class")


@pytest.mark.gpu
@pytest.mark.parametrize("windows", WINDOWS)
def test_fdn_composite(eng, oracle, windows):
    V, T = _voices(50, windows), 30
    x = np.repeat((GOLD["comp_x"] * np.float32(0.1))[None, :], V, 0).copy()
    x[1:] = lcg_noise(np.arange(V - 1, dtype=np.uint32) + 77, 64 * T) * np.float32(0.1)
    times, omegas, gains = [133.0, 201.0, 307.0, 419.0], [0.2, 0.15, 0.1, 0.05], [0.8, 0.75, 0.7, 0.65]
    desc = [dict(name="x", type="input")]
    sub, outs = patches.fdn(4, "x", 512.0)
    desc += sub
    coeffs = {f"fdn_filter{n}": np.repeat(oracle.make_coeffs("onepole", omegas[n]).reshape(2, 1), V, 1) for n in range(4)}
    params = {f"fdn_gain{n}": np.full(V, gains[n], np.float32) for n in range(4)}

    def edit(st):
        for n in range(4):
            st[f"fdn_delay{n}"][1] = np.uint32(max(1, int(times[n] - 64)))
            st[f"fdn_delay{n}"][1, 1:] += (np.arange(V - 1, dtype=np.uint32) % 37)
    got = _composite(eng, oracle, desc, outs, V, T, {"x": x}, params, coeffs, edit, windows)
    assert_bits_equal(got[0][0], GOLD["fdnL"], True, "FDN<4> sumL voice 0 vs the reference class")
    assert_bits_equal(got[1][0], GOLD["fdnR"], True, "FDN<4> sumR voice 0 vs the reference class")


@pytest.mark.gpu
@pytest.mark.parametrize("windows", WINDOWS)
def test_feedback_delay_function(eng, oracle, windows):
    V, T = _voices(40, windows), 30
    x = np.repeat(GOLD["comp_x"][None, :], V, 0).copy()
    dsig = np.repeat((GOLD["comp_dsig"] + np.float32(150.0))[None, :], V, 0).copy()
    dsig[1:] += np.linspace(0, 300, V - 1, dtype=np.float32)[:, None]
    co = oracle.make_coeffs("lopass", 0.08, 0.9)
    desc = [dict(name="x", type="input"), dict(name="dl", type="input"), dict(name="g", type="const", value=0.6),
            dict(name="c64", type="const", value=64.0), dict(name="vy1", type="feedback", source="delay"),
            dict(name="fb", type="op", kind=Op.MULTIPLY, inputs=["vy1", "g"]), dict(name="sum", type="op", kind=Op.ADD, inputs=["x", "fb"]),
            dict(name="fn", type="proc", kind=Proc.LOPASS, inputs=["sum"]), dict(name="dt", type="op", kind=Op.SUBTRACT, inputs=["dl", "c64"]),
            dict(name="delay", type="proc", kind=Proc.PITCHBENDABLE_DELAY, inputs=["fn", "dt"], max_delay=1000.0)]
    (got,) = _composite(eng, oracle, desc, ["fn"], V, T, {"x": x, "dl": dsig}, {}, {"fn": np.repeat(co.reshape(3, 1), V, 1)}, lambda st: None, windows)
    assert_bits_equal(got[0], GOLD["fbdelay"], True, "FeedbackDelayFunction voice 0 vs the reference class")


@pytest.mark.gpu
def test_delay_rules(eng):
    import madronalib_amd as ml
    with pytest.raises(ml.MlgpuError) as ei:
        eng.bank([Proc.INTEGER_DELAY], 64)                       # delay lines are graph nodes
    assert ei.value.status == ml.Status.ERR_UNSUPPORTED
    g = ml.Graph(eng, 64)
    a = g.add("a", "input")
    d = g.add("d", "proc", Proc.INTEGER_DELAY, [a])
    g.add_output(d)
    with pytest.raises(ml.MlgpuError):
        g.compile()                                              # no memory: set_max_delay missing
    with pytest.raises(ml.MlgpuError):
        g.set_max_delay(a, 100.0)                                # not a delay node
    g.set_max_delay(d, 100.0)
    fb = g.add("fb", "feedback")
    with pytest.raises(ml.MlgpuError):
        g.compile()                                              # feedback without a source
    g.set_feedback(fb, d)
    g.compile()
    g = ml.Graph(eng, 64, delay_windows=True)                    # LDS windows: 8 KiB per ring, 20 rings at most
    a = g.add("a", "input")
    for i in range(21):
        g.add(f"d{i}", "proc", Proc.INTEGER_DELAY, [a], max_delay=100.0)
    g.add_output("d20")
    with pytest.raises(ml.MlgpuError) as ei:
        g.compile()
    assert ei.value.status == ml.Status.ERR_UNSUPPORTED
    g = ml.Graph(eng, 64, delay_windows=2)                       # layout 2: 40 KiB of LDS per ring, four rings at most
    a = g.add("a", "input")
    for i in range(5):
        g.add(f"d{i}", "proc", Proc.INTEGER_DELAY, [a], max_delay=100.0)
    g.add_output("d4")
    with pytest.raises(ml.MlgpuError) as ei:
        g.compile()
    assert ei.value.status == ml.Status.ERR_UNSUPPORTED
    g = ml.Graph(eng, 100, delay_windows=2)                      # a last wavefront that is not full is fine (its spare lanes run the last voice again) ...
    a = g.add("a", "input")
    g.add("d", "proc", Proc.INTEGER_DELAY, [a], max_delay=100.0)
    g.add_output("d")
    g.set_output_group_sum(0, 4)
    with pytest.raises(ml.MlgpuError) as ei:                     # ... unless voices are summed in groups inside the kernel
        g.compile()
    assert ei.value.status == ml.Status.ERR_UNSUPPORTED
    # layout 3 = "per-voice delay times, the best form that applies": decided by compile
    # (round 6: one or two rings - the transposed windows; more - the sector trips, layout 4)
    for V, n_delays, want in ((256, 4, 4), (256, 2, 2), (256, 3, 4), (256, 5, 4), (200, 1, 2)):
        g = ml.Graph(eng, V, delay_windows="best")
        assert g.delay_layout == 3
        a = g.add("a", "input")
        for i in range(n_delays):
            g.add(f"d{i}", "proc", Proc.INTEGER_DELAY, [a], max_delay=100.0)
        g.add_output(f"d{n_delays - 1}")
        g.compile()
        assert g.delay_layout == want, (V, n_delays, g.delay_layout)
        g.close()
    g3 = ml.Graph(eng, 1000, delay_windows=False)
    assert g3.delay_layout == 0
    a = g3.add("a", "input")
    g3.add("d", "proc", Proc.PITCHBENDABLE_DELAY, [a, a], max_delay=1000.0)
    g3.add_output("d")
    g3.compile()
    assert g3.device_bytes == 4 * 1000 * (1 + (10 + 1) + 1 + 2 * 2048)     # coefficient, state and constant slots (+1 spare each), two rings of 2048
    bank = eng.bank([Proc.ALLPASS1], 64)                         # Allpass1 has no ring: fine in a bank
    assert bank.num_coeffs(0) == 1 and bank.num_state(0) == 2


@pytest.mark.gpu
@pytest.mark.parametrize("unequal_w", [False, True, 5, 8, 13])
def test_transposed_rings_every_kind_of_lane(eng, oracle, unequal_w):
    """Layout 2 on the bench's plucked-string voice (noise burst -> FractionalDelay of per-voice length -> OnePole -> one-vector
    feedback), 256 voices whose delay times cover every path of the transposed windows: under 8 samples (served from the write
    window), 8-47 (too close behind the writer to fetch a period ahead: the lane's own history, kept in its read rows), and up to
    the ring's maximum (the prefetched read windows), several launches with carried state; the same with the write indices of one
    wavefront set apart by the host (that wavefront then runs its plain per-sample form); and with every voice's write index at
    5 / 8 / 13 (launches that begin inside a chunk: its first half is in memory and not in the windows, its second half goes out
    alone). Against the streaming oracle and against layout 0."""
    import madronalib_amd as ml
    V, T, launches = 256, 6, 3
    desc = [dict(name="x", type="input"), dict(name="g", type="const", value=0.995),
            dict(name="fb", type="feedback", source="damp"),
            dict(name="fbg", type="op", kind=Op.MULTIPLY, inputs=["fb", "g"]),
            dict(name="sum", type="op", kind=Op.ADD, inputs=["x", "fbg"]),
            dict(name="line", type="proc", kind=Proc.FRACTIONAL_DELAY, inputs=["sum"], max_delay=1024.0),
            dict(name="damp", type="proc", kind=Proc.ONE_POLE, inputs=["line"])]
    rng = np.random.default_rng(5)
    length = np.concatenate([np.linspace(1.3, 15.7, 40), np.linspace(16.2, 47.9, 40), np.linspace(48.1, 959.0, 112), rng.uniform(1.0, 959.0, 64)]).astype(np.float32)
    x = lcg_noise(np.arange(V, dtype=np.uint32) + 9, 64 * T * launches)
    x[:, 640:] = 0
    co = oracle.make_coeffs("onepole", 0.3)
    outs = {}
    for layout in (0, 2, 4):
        g = ml.Graph(eng, V, desc, ["damp"], delay_windows=layout)
        g.set_coeffs("damp", [np.full(V, c, np.float32) for c in co])
        st = new_stream_state(oracle, desc, V)
        fs = np.stack([oracle.fractional_delay_state(float(d)) for d in length], 1)     # [2][V]: delayInt bits, allpass coefficient
        st["line"][3] = fs[0].view(np.uint32)
        st["line"][4] = fs[1].view(np.uint32)
        if unequal_w is not True and unequal_w:
            st["line"][0, :] = unequal_w
        elif unequal_w:
            st["line"][0, 64:128] = (np.arange(64, dtype=np.uint32) * 7) % 1024          # the second wavefront: write indices all over the ring
        for i in range(st["line"].shape[0]):
            g.set_state("line", i, st["line"][i])
        got, want = [], []
        for k in range(launches):
            part = {"x": np.ascontiguousarray(x[:, k * 64 * T:(k + 1) * 64 * T])}
            got.append(g.process_host(T, part, Layout.QUAD)[0])
            want.append(evaluate_stream(oracle, desc, ["damp"], V, T, part, {}, {"damp": np.repeat(np.asarray(co, np.float32).reshape(-1, 1), V, 1)}, st)[0])
        outs[layout] = np.concatenate(got, 1)
        assert_bits_equal(outs[layout], np.concatenate(want, 1), True, f"plucked strings, delay layout {layout}")
        for i in range(st["line"].shape[0]):
            assert (g.get_state("line", i) == st["line"][i]).all(), (layout, i)
        g.close()
    assert_bits_equal(outs[2], outs[0], True, "layout 2 vs layout 0")
    assert_bits_equal(outs[4], outs[0], True, "layout 4 vs layout 0")
    assert np.abs(outs[0]).max() > 0.01


@pytest.mark.gpu
@pytest.mark.parametrize("w0", [0, 3, 8, 12])
@pytest.mark.parametrize("name", ["integer_var", "pitchbend", "frac_ticks"])
def test_transposed_rings_moving_delay_times(eng, oracle, name, w0):
    """Layout 2 with delay times that move while it runs (steps between 0 and the maximum at random moments, slow sweeps): lanes go
    back and forth between the prefetched windows, their own history and - for the samples after a jump that neither holds - memory,
    where the half chunk the wavefront still had parked in registers must have arrived first. Launches beginning at any place of a
    chunk. Bit for bit the oracle, and layout 0."""
    from graph_oracle import ring_len
    V, T = 256, 7
    c = delay_case(oracle, name, V, 2 * T, seed=31 + w0)
    st0 = c["state0"].copy()
    st0[0] = w0
    got = {}
    for layout in (0, 2):
        g, names = delay_graph(eng, V, c["kind"], len(c["inputs"]), c["max_delay"], layout)
        got[layout] = run_delay(g, names, st0, c["inputs"], T, Layout.QUAD)
        g.close()
    rings = 2 if c["kind"] == Proc.PITCHBENDABLE_DELAY else 1
    st, mem = st0.copy(), np.zeros((V, rings, ring_len(c["max_delay"])), np.float32)
    for call in range(2):
        sl = slice(call * 64 * T, (call + 1) * 64 * T)
        want = oracle.delay_process(c["kind"], T, st, mem, [np.ascontiguousarray(a[:, sl]) for a in c["inputs"]])
        for layout in (0, 2):
            assert_bits_equal(got[layout][0][call], want, True, f"{name} w0={w0} layout {layout} call {call}")
            assert_bits_equal(got[layout][1][call], st, False, f"{name} w0={w0} layout {layout} state after call {call}")


@pytest.mark.gpu
def test_transposed_rings_random_graphs(eng):
    """A short run of tools/ring_layout_soak.py: random delay nodes, ring sizes, delay-time signals (constant, stepped, swept through
    the short / long boundary, a new one every sample), launch lengths and write indices - layout 2 against layout 0, outputs and
    state words bit for bit. (1 650 cases of it on the round's closing code: profiles/r05_ring_layout_soak.txt.)"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("ring_layout_soak", os.path.join(os.path.dirname(__file__), "..", "tools", "ring_layout_soak.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(40, 11, eng) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [3, 4, 5, 6])
def test_rows_early_reads_equal_plain_rows(eng, monkeypatch, seed):
    """Ring layout 0 with three or more ring reads per sample issues them ahead of the sample's arithmetic by LDS-DMA (graph.hip:
    earlyRows, RingCore::readEarly / early). Random chains of 3 .. 6 delay nodes of all three kinds - delay times from inputs (read at the
    top of the sample, together), from a constant, none at all (the state's), or made of the previous node's output (read where the node
    stands) - with times of 0, of the ring's whole length and beyond, several launches with carried state, whole and ragged wavefronts:
    outputs and every state word against the same graph with the plain loads (MLGPU_GRAPH_EARLY_READS=0, the form the oracle tests of
    rounds 2-5 pinned), bit for bit."""
    import madronalib_amd as ml
    rng = np.random.default_rng(seed)
    for case in range(6):
        V = int(rng.integers(1, 300)) if case % 2 else 64 * int(rng.integers(1, 5))
        T, launches = int(rng.integers(1, 6)), 3
        S = 64 * T * launches
        n = int(rng.integers(3, 7))
        dmax = float([40.0, 192.0, 700.0][int(rng.integers(0, 3))])
        ring = 1 << int(np.ceil(np.log2(max(64, int(dmax) + 64))))
        kinds = [[Proc.INTEGER_DELAY, Proc.FRACTIONAL_DELAY, Proc.PITCHBENDABLE_DELAY][int(rng.integers(0, 3))] for _ in range(n)]
        tmodes = [int(rng.integers(0, 4)) for _ in range(n)]   # 0 an input, 1 a constant, 2 none, 3 made of the previous node's output
        sig = {"x": lcg_noise(np.arange(V, dtype=np.uint32) + np.uint32(seed * 131 + case), S)}
        desc = [dict(name="x", type="input"), dict(name="half", type="const", value=0.5), dict(name="scale", type="const", value=float(dmax))]
        src = "x"
        for i, (kind, tm) in enumerate(zip(kinds, tmodes)):
            if kind == Proc.PITCHBENDABLE_DELAY and tm == 2:
                tm = 0                                         # (its call takes a delay time)
            ins = [src]
            if tm == 0:
                d = stepped(V, S, seed * 17 + i, 0.0, dmax + 0.9, 1, 150)
                d[:, ::97] = 0.0                               # the read lands on the write ...
                d[:, 5::131] = np.float32(ring)                # ... also by the ring's whole length
                d[:, 7::173] = np.float32(ring + 3)            # ... and beyond it
                sig[f"dt{i}"] = d
                desc.append(dict(name=f"dt{i}", type="input"))
                ins.append(f"dt{i}")
            elif tm == 1:
                desc.append(dict(name=f"dt{i}", type="const", value=float(rng.integers(0, int(dmax)))))
                ins.append(f"dt{i}")
            elif tm == 3:
                desc += [dict(name=f"ab{i}", type="op", kind=Op.ABS, inputs=[src]), dict(name=f"dt{i}", type="op", kind=Op.MULTIPLY, inputs=[f"ab{i}", "scale"])]
                ins.append(f"dt{i}")
            desc.append(dict(name=f"d{i}", type="proc", kind=kind, inputs=ins, max_delay=dmax))
            desc += [dict(name=f"m{i}", type="op", kind=Op.ADD, inputs=[f"d{i}", "x"]), dict(name=f"s{i}", type="op", kind=Op.MULTIPLY, inputs=[f"m{i}", "half"])]
            src = f"s{i}"
        got = {}
        for early in ("1", "0", "round5"):
            # "round5": neither the early reads nor the 32-bit row offsets (VoiceMem::ringPtr, state_row) - the rows as rounds 2-5 had them
            monkeypatch.setenv("MLGPU_GRAPH_EARLY_READS", "1" if early == "1" else "0")
            monkeypatch.setenv("MLGPU_GRAPH_ROW_ADDR32", "0" if early == "round5" else "1")
            g = ml.Graph(eng, V, desc, [src, "d0"], delay_windows=0)
            assert ("ldsEarly" in g.source) == (early == "1")
            assert (", true}" in g.source) == (early != "round5")
            for i, kind in enumerate(kinds):
                g.set_state(f"d{i}", 0, ((np.arange(V, dtype=np.uint32) * (1 if case % 3 == 0 else 0) * 5 + case * 11 + i) % ring).astype(np.uint32))
                if kind == Proc.INTEGER_DELAY:
                    g.set_state(f"d{i}", 1, np.full(V, (7 * i + case) % int(dmax), np.uint32))
            outs = []
            for k in range(launches):
                part = {name: np.ascontiguousarray(a[:, k * 64 * T:(k + 1) * 64 * T]) for name, a in sig.items()}
                outs.append(np.stack(g.process_host(T, part, Layout.QUAD)))
            states = []
            for i in range(n):
                states += [g.get_state(f"d{i}", j) for j in range(g.num_state(f"d{i}"))]
            got[early] = (np.concatenate(outs, 2), np.stack(states))
            g.close()
        assert np.any(got["1"][0] != 0)
        for other in ("0", "round5"):
            assert_bits_equal(got["1"][0], got[other][0], False, f"seed {seed} case {case}: outputs, early ring reads against the plain rows ({other})")
            assert np.array_equal(got["1"][1], got[other][1]), f"seed {seed} case {case}: state words ({other})"


def _soak_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("ring_layout_soak", os.path.join(os.path.dirname(__file__), "..", "tools", "ring_layout_soak.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.gpu
def test_sector_trips_delay_time_rising_through_16(eng):
    """Ring layout 4 tags the sectors it holds; a lane that reads under 16 samples back is served from the history rows and loads
    sectors the writer has not finished (it never looks at them). Such a lane must not leave them tagged: a delay time that then rises to
    16 .. 23 at the start of a trip found "its" two sectors held and read the ring's previous lap (found by tools/ring_layout_soak.py,
    seed 64, case 102 - a FractionalDelay whose delay-time signal goes 15.4 -> 17.5 -> 38.7 - after 1 300 clean cases; RingCore::tripStart).
    Both here: that case replayed, and the pattern by construction for the three kinds, against layout 0, bit for bit."""
    import madronalib_amd as ml
    mod = _soak_module()
    mod.TEST_LAYOUT, mod.LAYOUTS, mod.ONLY = 4, (0, 4), 102
    assert mod.run(103, 64, eng) == 0
    V, T = 69, 7
    S = 64 * T
    x = lcg_noise(np.arange(V, dtype=np.uint32) + 9, S)
    for kind in (Proc.INTEGER_DELAY, Proc.FRACTIONAL_DELAY, Proc.PITCHBENDABLE_DELAY):
        for rise_at, to in ((300, 17.5), (296, 16.2), (304, 23.4), (288, 19.0)):
            d = np.full((V, S), np.float32(14.3))
            d[:, rise_at:] = np.float32(to)
            d[::3, rise_at + 40:] = np.float32(38.7)
            outs = {}
            for layout in (0, 4):
                g = ml.Graph(eng, V, delay_windows=layout)
                g.add("x", "input")
                g.add("dt", "input")
                g.add("d", "proc", kind, ["x", "dt"], max_delay=100.0)
                g.add_output("d")
                g.compile()
                outs[layout] = g.process_host(T, {"x": x, "dt": d}, Layout.QUAD)[0]
                g.close()
            assert_bits_equal(outs[4], outs[0], False, f"kind {int(kind)}: delay time 14.3 -> {to} at sample {rise_at}, ring layout 4 against layout 0")


@pytest.mark.gpu
def test_ring_layouts_random_chains(eng):
    """A short run of tools/ring_graph_soak.py: random chains of 2 .. 6 delay nodes of all three kinds in one kernel, every ring layout
    (and "the best one") against the plain rows of rounds 2-5, outputs and state words bit for bit. (750 chains of it at the round's
    end: profiles/r06_ring_graph_soak.txt.)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ring_graph_soak", os.path.join(os.path.dirname(__file__), "..", "tools", "ring_graph_soak.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(14, 31, eng) == 0
