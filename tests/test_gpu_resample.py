"""GPU parity of Downsampler / Upsampler (HalfBandFilter cascades, MLDSPFilters.h:1245-1473) against the oracle and
against golden outputs of the reference classes."""
import os

import numpy as np
import pytest

from inputs import assert_bits_equal, lcg_noise
from madronalib_amd.constants import Layout

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "resample.npz"))


@pytest.fixture(scope="module")
def eng():
    import madronalib_amd as ml
    e = ml.Engine(0)
    yield e
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("octaves", [0, 1, 2, 3, 6])
@pytest.mark.parametrize("up", [False, True])
@pytest.mark.parametrize("layout", [Layout.QUAD, Layout.VOICE_MAJOR])
def test_resampler_vs_oracle(eng, oracle, octaves, up, layout):
    import madronalib_amd as ml
    V = 300
    Tin = 2 * (1 << octaves) if not up else 2
    x = lcg_noise(np.arange(V, dtype=np.uint32) + 11, 64 * Tin * 2)
    r = ml.Resampler(eng, V, octaves, up)
    st = np.zeros((octaves * 9, V), np.float32)
    for call in range(2):   # state carries from launch to launch
        xs = np.ascontiguousarray(x[:, call * 64 * Tin:(call + 1) * 64 * Tin])
        assert_bits_equal(r.process_host(xs, layout), oracle.resample(octaves, up, st, xs), True, f"octaves {octaves} up {up} call {call}")
        assert_bits_equal(r.get_state(), st, True, "HalfBandFilter state")
    r.clear()
    assert (r.get_state() == 0).all()


@pytest.mark.gpu
def test_resampler_golden_and_round_trip(eng):
    """Reference Downsampler / Upsampler outputs (golden), and up then down by the same factor stays close to the input
    delayed by the filters (a property independent of size)."""
    import madronalib_amd as ml
    x = GOLD["x"]
    V = x.shape[0]
    for octaves in (1, 3):
        for up in (False, True):
            r = ml.Resampler(eng, V, octaves, up)
            assert_bits_equal(r.process_host(x, Layout.ROWS), GOLD[f"{'up' if up else 'down'}{octaves}"], True, f"golden octaves {octaves} up {up}")
    with pytest.raises(ml.MlgpuError):
        ml.Resampler(eng, V, 2, False).process_host(x[:, :64 * 3])   # a downsampler needs a multiple of 2^octaves vectors
    with pytest.raises(ml.MlgpuError):
        ml.Resampler(eng, V, 9, True)
    # low-frequency sine: up 2x then down 2x reproduces it (delayed)
    n = np.arange(64 * 16)
    s = np.sin(2 * np.pi * 0.01 * n).astype(np.float32)[None, :]
    up, down = ml.Resampler(eng, 1, 1, True), ml.Resampler(eng, 1, 1, False)
    y = down.process_host(up.process_host(s))
    # the allpass interpolators have a fractional group delay: compare at the best integer alignment (error <= half a
    # sample of slope) and check that the amplitude is preserved
    best = min(np.abs(y[0, 200 + d:900 + d] - s[0, 200:900]).max() for d in range(0, 12))
    assert best < 0.04 and abs(np.abs(y[0, 200:900]).max() - 1.0) < 0.01


@pytest.mark.gpu
@pytest.mark.parametrize("octaves,up", [(1, False), (2, False), (1, True), (3, True)])
def test_resampler_hostile_input(eng, oracle, octaves, up):
    """The half-band cascades on a signal with infinities, NaNs, denormals, huge values and raw bit patterns mixed in: whatever
    the reference's allpass sections make of them (any NaN equals any NaN), output and filter state, over two launches."""
    import madronalib_amd as ml
    from inputs import general_floats
    V = 130
    Tin = 2 * (1 << octaves) if not up else 2
    S = 64 * Tin * 2
    x = lcg_noise(np.arange(V, dtype=np.uint32) + 5, S)
    g = general_floats(V * S, 40 + octaves).reshape(V, S)
    rng = np.random.default_rng(octaves)
    mask = rng.random((V, S)) < 0.03
    x[mask] = g[mask]
    big = np.isfinite(x) & (np.abs(x) > 1e15)     # (as for the state-variable filters: stay below the float range's last octaves)
    x[big] = np.float32(1e15) * np.sign(x[big])
    r = ml.Resampler(eng, V, octaves, up)
    st = np.zeros((octaves * 9, V), np.float32)
    for call in range(2):
        xs = np.ascontiguousarray(x[:, call * 64 * Tin:(call + 1) * 64 * Tin])
        assert_bits_equal(r.process_host(xs, Layout.QUAD), oracle.resample(octaves, up, st, xs), True, f"hostile octaves {octaves} up {up} call {call}")
        gs, ws = r.get_state().view(np.uint32), st.view(np.uint32)
        bothnan = np.isnan(gs.view(np.float32)) & np.isnan(ws.view(np.float32))
        assert ((gs == ws) | bothnan).all(), "HalfBandFilter state"
