"""The widened rows' bench workloads (bench.py --workload synth / strings / events / resample / reverb) at 512 voices, bit for bit
against the CPU checkers: what the driver's line reports as `<leg>_crc_match` (tests/widened_parity.py holds the cases)."""
import numpy as np
import pytest

import widened_parity as wp
from inputs import assert_bits_equal


@pytest.fixture(scope="module")
def eng():
    import madronalib_amd as ml
    e = ml.Engine(0)
    yield e
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["resample", "strings", "events", "synth", "reverb"])
def test_widened_leg_matches_cpu(eng, oracle, name):
    cases = wp.all_cases(eng, oracle)
    if name not in cases:
        pytest.skip(f"{name}: needs the compiled reference (oracle/_ref) / tests/cpp/libexamples_gpu.so, absent here")
    got, want, checker = cases[name]()
    assert got.shape == want.shape
    assert_bits_equal(got, want, True, f"bench workload {name} at {wp.VOICES} voices vs {checker}")
    assert np.abs(want).max() > 0


@pytest.mark.gpu
def test_events_reserve_for_graph_refuses_longer_blocks(eng):
    """mlgpu_events_reserve_for_graph: after a setup-time reserve, a longer block is refused (MLGPU_ERR_RANGE) before the router consumes
    its events - and nothing is allocated in the process call; without a reserve the first block sizes the buffers itself."""
    import madronalib_amd as ml
    from madronalib_amd import patches
    P, N = 4, 16
    V = N * P
    ev = ml.Events(eng, N, P, 48000.0)
    ev.set_wanted_rows([0, 1])
    assert ev.graph_reserve_bytes(8) == (16 + 2 * 256) * 8 * V
    desc, outn = patches.synth16(pitch_input=True, event_rows=True)
    g = ml.Graph(eng, V, desc, outn)
    g.bind_events(ev)
    g.clear()
    d_out = eng.alloc(4 * V * 16 * 64)
    g.process_events(4, 0, [], [d_out])          # no reserve yet: the call sizes the buffers (a convenience)
    g.process_events(8, 0, [], [d_out])          # ... and grows them
    ev.reserve_for_graph(8)
    g.process_events(8, 0, [], [d_out])
    ev.add_event(0, ml.Event(1, 1, 60, 5, 0.0, 0.8))
    with pytest.raises(ml.MlgpuError) as ex:
        g.process_events(16, 0, [], [d_out])
    assert ex.value.status == ml.Status.ERR_RANGE
    g.process_events(8, 0, [], [d_out])          # the refused block's event is still there and is consumed now
    ev.clear_events()
    g.close()
