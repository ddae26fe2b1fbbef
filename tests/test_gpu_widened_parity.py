"""The widened rows' bench workloads (bench.py --workload synth / strings / events / resample / reverb) at 512 voices, bit for bit
against the CPU checkers: what the driver's line reports as `<leg>_crc_match` (tests/widened_parity.py holds the cases)."""
import os

import numpy as np
import pytest

import widened_parity as wp
from inputs import assert_bits_equal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


@pytest.mark.gpu
def test_reverb_example_full_bank(monkeypatch):
    """The reference's reverb.cpp through the shim at the BENCH's size - 65 536 stereo reverbs x 16 DSPVectors, every CU busy, one wavefront per
    SIMD - with the ring reads of a sample issued together by LDS-DMA and waited for by count (graph.hip: earlyRows; the bench's form) against
    the plain loads of rounds 2-5 (MLGPU_GRAPH_EARLY_READS=0): every word of every voice; and its first 2 048 voices against the reference
    compiled on the CPU. (A counted wait that were one too lax would show under load, not in a 512-voice case.)"""
    import ctypes
    from inputs import lcg_noise
    gso, rso = os.path.join(ROOT, "tests", "cpp", "libexamples_gpu.so"), os.path.join(ROOT, "oracle", "_ref", "libexamples_ref.so")
    if not (os.path.exists(gso) and os.path.exists(rso)):
        pytest.skip("tests/cpp/libexamples_gpu.so or oracle/_ref/libexamples_ref.so not built here")
    V, T, n = 65536, 16, 2048
    fp = ctypes.POINTER(ctypes.c_float)
    G, R = ctypes.CDLL(gso), ctypes.CDLL(rso)
    G.example_reverb_gpu_run.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, fp, fp, fp, fp, ctypes.c_char_p, ctypes.c_size_t]
    R.example_reverb_ref_run.argtypes = [ctypes.c_size_t, ctypes.c_size_t, fp, fp, fp, fp]
    x0 = (lcg_noise(np.arange(V, dtype=np.uint32), 64 * T) * np.float32(0.05)).astype(np.float32)
    x1 = (lcg_noise(np.arange(V, dtype=np.uint32) + (1 << 20), 64 * T) * np.float32(0.05)).astype(np.float32)
    p = lambda a: a.ctypes.data_as(fp)  # noqa: E731
    outs = {}
    for early in ("1", "0"):
        monkeypatch.setenv("MLGPU_GRAPH_EARLY_READS", early)
        g0, g1 = np.zeros((V, 64 * T), np.float32), np.zeros((V, 64 * T), np.float32)
        err = ctypes.create_string_buffer(2048)
        assert G.example_reverb_gpu_run(V, T, 2, p(x0), p(x1), p(g0), p(g1), err, 2048) == 0, err.value.decode()
        outs[early] = (g0, g1)
    assert np.abs(outs["1"][0]).max() > 0.01
    for c in (0, 1):
        assert_bits_equal(outs["1"][c], outs["0"][c], False, f"channel {c}: early ring reads against the plain rows, 65 536 voices")
    w0, w1 = np.zeros((n, 64 * T), np.float32), np.zeros((n, 64 * T), np.float32)
    assert R.example_reverb_ref_run(n, T, p(np.ascontiguousarray(x0[:n])), p(np.ascontiguousarray(x1[:n])), p(w0), p(w1)) == 0
    assert_bits_equal(outs["1"][0][:n], w0, False, "left, the first 2 048 voices of the full bank against the reference")
    assert_bits_equal(outs["1"][1][:n], w1, False, "right, the first 2 048 voices of the full bank against the reference")
