"""GPU parity of rate regions in a fused graph: Upsample2xFunction / Downsample2xFunction (MLDSPFunctional.h:114-213)
with the wrapped function written out as graph nodes. Bit-exact against the oracle's block restatement and against golden
outputs of the reference's own objects (tests/golden/regions.npz)."""
import os

import numpy as np
import pytest

from inputs import assert_bits_equal, lcg_noise, region_case
from madronalib_amd.constants import Layout, Op, Proc, Region

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "regions.npz"))


@pytest.fixture(scope="module")
def eng():
    import madronalib_amd as ml
    e = ml.Engine(0)
    yield e
    e.close()


def region_graph(eng, V, kind, freq, co, voices_per_lane=0):
    """out = F(fn, {x, m}) * 0.5 with fn(v) = Lopass((clamp(v0 * 3, -1, 1) + SawGen(freq)) * v1), as
    oracle/ref_wrapper.cpp mlref_rate_function_run builds it from the reference's objects."""
    import madronalib_amd as ml
    g = ml.Graph(eng, V, voices_per_lane=voices_per_lane)
    g.add("x", "input")
    g.add("m", "input")
    g.add("three", "const", value=3.0)
    g.add("lo", "const", value=-1.0)
    g.add("hi", "const", value=1.0)
    g.add("half", "const", value=0.5)
    g.add("freq", "param")
    g.begin_region(kind, ["x", "m"], ["rx", "rm"])
    g.add("drive", "op", Op.MULTIPLY, ["rx", "three"])
    g.add("sat", "op", Op.CLAMP, ["drive", "lo", "hi"])
    g.add("saw", "proc", Proc.SAW_GEN, ["freq"])
    g.add("mix", "op", Op.ADD, ["sat", "saw"])
    g.add("am", "op", Op.MULTIPLY, ["mix", "rm"])
    g.add("lp", "proc", Proc.LOPASS, ["am"])
    g.end_region("lp", "y")
    g.add("out", "op", Op.MULTIPLY, ["y", "half"])
    g.add_output("out")
    g.compile()
    g.set_param("freq", freq)
    g.set_coeffs("lp", [np.full(V, c, np.float32) for c in co])
    return g


def run(g, x, m, splits, layout=Layout.QUAD):
    outs, t0 = [], 0
    for n in splits:
        sl = slice(64 * t0, 64 * (t0 + n))
        (y,) = g.process_host(n, {"x": np.ascontiguousarray(x[:, sl]), "m": np.ascontiguousarray(m[:, sl])}, layout)
        outs.append(y)
        t0 += n
    return np.concatenate(outs, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,key", [(Region.UPSAMPLE_2X, "up"), (Region.DOWNSAMPLE_2X, "down")])
def test_rate_regions_golden(eng, kind, key):
    x, m, freq, co = GOLD["x"], GOLD["m"], GOLD["freq"], GOLD["co"]
    g = region_graph(eng, x.shape[0], kind, freq, co)
    assert_bits_equal(run(g, x, m, [10], Layout.VOICE_MAJOR), GOLD[key], True, f"{key} vs the reference objects")


@pytest.mark.gpu
@pytest.mark.parametrize("vpl", [1, 2])
@pytest.mark.parametrize("kind", [Region.UPSAMPLE_2X, Region.DOWNSAMPLE_2X])
def test_rate_regions_vs_oracle_split_launches(eng, oracle, kind, vpl):
    """Launch boundaries are invisible, also odd ones (a Downsample2x region pairs vectors 2k, 2k + 1 across launches);
    clear() restarts the pairing and every object."""
    V, T = 300, 12
    x, m, freq = region_case(V, T, seed=9)
    co = oracle.make_coeffs("lopass", 0.2, 0.8)
    want = oracle.rate_function_run(kind == Region.UPSAMPLE_2X, freq, co, x, m)
    g = region_graph(eng, V, kind, freq, co, voices_per_lane=vpl)
    assert_bits_equal(run(g, x, m, [12]), want, True, "one launch")
    g.clear()
    assert_bits_equal(run(g, x, m, [3, 1, 5, 3]), want, True, "odd split launches")
    g.clear()
    assert_bits_equal(run(g, x, m, [1] * 12), want, True, "vector by vector")


@pytest.mark.gpu
@pytest.mark.parametrize("kind,key", [(Region.UPSAMPLE_2X, "ap_up"), (Region.DOWNSAMPLE_2X, "ap_down")])
def test_rate_region_with_feedback_and_delay(eng, kind, key):
    """fn = OnePole(Allpass<IntegerDelay>(x)): a delay ring and a one-vector feedback (Allpass::vy1) that live at fn's own
    rate, inside the region; golden outputs of the reference objects, split launches included."""
    import madronalib_amd as ml
    from madronalib_amd import patches
    x, opc = GOLD["x"], GOLD["opc"]
    V, T = x.shape[0], x.shape[1] // 64
    g = ml.Graph(eng, V)
    g.add("x", "input")
    g.begin_region(kind, ["x"], ["rx"])
    sub, y = patches.allpass("ap_", "rx", Proc.INTEGER_DELAY, 300.0)
    for n in sub:
        g.add(**{k: v for k, v in n.items() if k != "source"})
    for n in sub:
        if n["type"] == "feedback":
            g.set_feedback(n["name"], n["source"])
    g.add("lp", "proc", Proc.ONE_POLE, [y])
    out = g.end_region("lp", "out")
    g.add_output(out)
    g.compile()
    g.set_param("ap_gain", 0.6)
    g.set_coeffs("lp", [np.full(V, c, np.float32) for c in opc])
    g.set_state("ap_delay", 1, np.full(V, 171 - 64, np.uint32))     # Allpass::setDelayInSamples(d): the inner delay gets d - 64
    got = np.concatenate([g.process_host(n, {"x": np.ascontiguousarray(x[:, 64 * a:64 * (a + n)])}, Layout.QUAD)[0] for a, n in ((0, 3), (3, 1), (4, 6))], 1)
    assert_bits_equal(got, GOLD[key], True, f"{key} vs the reference objects")
    assert np.abs(GOLD[key]).max() > 0.1
    # a feedback node may not cross the region's border
    g2 = ml.Graph(eng, 64)
    g2.add("x", "input")
    fb = g2.add("fb", "feedback")
    (rx,) = g2.begin_region(kind, ["x"])
    inner = g2.add("i", "op", Op.ADD, [rx, rx])
    o = g2.end_region(inner, "o")
    g2.set_feedback(fb, inner)
    g2.add_output(o)
    with pytest.raises(ml.MlgpuError):
        g2.compile()


@pytest.mark.gpu
@pytest.mark.parametrize("outer,inner,key", [(Region.UPSAMPLE_2X, Region.UPSAMPLE_2X, "nested_uu"), (Region.UPSAMPLE_2X, Region.DOWNSAMPLE_2X, "nested_ud"),
                                             (Region.DOWNSAMPLE_2X, Region.UPSAMPLE_2X, "nested_du"), (Region.DOWNSAMPLE_2X, Region.DOWNSAMPLE_2X, "nested_dd")])
def test_nested_rate_regions(eng, outer, inner, key):
    """out = Outer(mid, x), mid(v) = OnePole(Inner(fn, v)) + v / 2, fn(w) = Lopass(clamp(3 w, -1, 1)): a region inside a region
    (4x oversampling, and the mixed cases), stateful processors at both inner levels; golden outputs of the reference's
    function objects nested the same way; launches of 1, 2 and 7 vectors."""
    import madronalib_amd as ml
    x, co, opc = GOLD["x"], GOLD["co"], GOLD["opc"]
    V, T = x.shape[0], x.shape[1] // 64
    g = ml.Graph(eng, V)
    g.add("x", "input")
    g.add("three", "const", value=3.0)
    g.add("lo", "const", value=-1.0)
    g.add("hi", "const", value=1.0)
    g.add("half", "const", value=0.5)
    (v,) = g.begin_region(outer, ["x"], ["v"])
    (w,) = g.begin_region(inner, [v], ["w"])
    g.add("drive", "op", Op.MULTIPLY, [w, "three"])
    g.add("sat", "op", Op.CLAMP, ["drive", "lo", "hi"])
    g.add("lp", "proc", Proc.LOPASS, ["sat"])
    innerOut = g.end_region("lp", "innerOut")
    g.add("op", "proc", Proc.ONE_POLE, [innerOut])
    g.add("dry", "op", Op.MULTIPLY, [v, "half"])
    g.add("mid", "op", Op.ADD, ["op", "dry"])
    out = g.end_region("mid", "out")
    g.add_output(out)
    g.compile()
    g.set_coeffs("lp", [np.full(V, c, np.float32) for c in co])
    g.set_coeffs("op", [np.full(V, c, np.float32) for c in opc])
    got = np.concatenate([g.process_host(n, {"x": np.ascontiguousarray(x[:, 64 * a:64 * (a + n)])}, Layout.QUAD)[0] for a, n in ((0, 1), (1, 2), (3, 7))], 1)
    assert_bits_equal(got, GOLD[key], True, f"{key} vs the reference objects")


@pytest.mark.gpu
def test_region_rules(eng):
    import madronalib_amd as ml
    g = ml.Graph(eng, 64)
    g.add("x", "input")
    g.add("c", "control")
    g.add("p", "param")
    with pytest.raises(ml.MlgpuError):
        g.add("hb", "proc", Proc.HALF_BAND, ["x"])                  # only begin_region / end_region make these
    with pytest.raises(ml.MlgpuError):
        g.end_region("x")                                           # no region open
    (rx,) = g.begin_region(Region.UPSAMPLE_2X, ["x"])
    with pytest.raises(ml.MlgpuError):
        g.begin_region(Region.UPSAMPLE_2X, ["x"])                   # a nested region takes signals of the enclosing one
    with pytest.raises(ml.MlgpuError):
        g.add("bad", "op", Op.ADD, [rx, "x"])                       # an outer audio-rate signal does not exist at 2x
    with pytest.raises(ml.MlgpuError):
        g.add("bad2", "op", Op.ADD, [rx, "c"])                      # nor does a control
    ok = g.add("ok", "op", Op.MULTIPLY, [rx, "p"])                  # per-voice floats are fine
    with pytest.raises(ml.MlgpuError):
        g.end_region("x")                                           # result must be a node of the region
    with pytest.raises(ml.MlgpuError):
        g.add_output(ok) or g.compile()                             # region still open / output inside a region
    y = g.end_region(ok, "y")
    with pytest.raises(ml.MlgpuError):
        g.add("leak", "op", Op.ADD, [ok, "x"])                      # region nodes are invisible outside
    g2 = ml.Graph(eng, 64)
    g2.add("x", "input")
    (r2,) = g2.begin_region(Region.DOWNSAMPLE_2X, ["x"])
    y2 = g2.end_region(r2, "y")                                     # identity fn: down then up, one vector late
    g2.add_output(y2)
    g2.compile()
    assert g2.num_state(y2) == 9 + 64 and g2.num_state(r2) == 9
    x = np.random.default_rng(1).standard_normal((64, 128)).astype(np.float32)
    (out,) = g2.process_host(2, {"x": x}, Layout.QUAD)
    assert (out[:, :64] == 0).all() and np.abs(out[:, 64:]).max() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [Region.UPSAMPLE_2X, Region.DOWNSAMPLE_2X])
def test_rate_region_with_delay_in_every_ring_layout(eng, kind):
    """The same region (a delay ring and a one-vector feedback at fn's own rate: 128 or 32 samples per DSPVector) with the ring in
    layouts 1 and 2 (sectors / transposed pieces behind LDS windows, whose clocks count the REGION's samples) against layout 0,
    which the golden test above pins; per-voice delay times, split launches."""
    import madronalib_amd as ml
    from madronalib_amd import patches
    V, T = 128, 10
    x = lcg_noise(np.arange(V, dtype=np.uint32) + 77, 64 * T)
    opc = GOLD["opc"]
    outs = {}
    for layout in (0, 1, 2):
        g = ml.Graph(eng, V, delay_windows=layout)
        g.add("x", "input")
        g.begin_region(kind, ["x"], ["rx"])
        sub, y = patches.allpass("ap_", "rx", Proc.INTEGER_DELAY, 300.0)
        for n in sub:
            g.add(**{k: v for k, v in n.items() if k != "source"})
        for n in sub:
            if n["type"] == "feedback":
                g.set_feedback(n["name"], n["source"])
        g.add("lp", "proc", Proc.ONE_POLE, [y])
        out = g.end_region("lp", "out")
        g.add_output(out)
        g.compile()
        assert g.delay_layout == layout
        g.set_param("ap_gain", 0.6)
        g.set_coeffs("lp", [np.full(V, c, np.float32) for c in opc])
        g.set_state("ap_delay", 1, ((np.arange(V) * 37) % 237).astype(np.uint32))
        outs[layout] = np.concatenate([g.process_host(n, {"x": np.ascontiguousarray(x[:, 64 * a:64 * (a + n)])}, Layout.QUAD)[0] for a, n in ((0, 3), (3, 1), (4, 6))], 1)
        g.close()
    assert np.abs(outs[0]).max() > 0.1
    assert_bits_equal(outs[1], outs[0], True, "ring layout 1 in a rate region")
    assert_bits_equal(outs[2], outs[0], True, "ring layout 2 in a rate region")
