"""The C oracle's restatements of the two per-instrument context signals - AudioContext::ProcessTime (the transport phasor) and
EventsToSignals::SmoothedController (what getInputController returns) - pinned against the reference's own classes
(oracle/_ref/libdropin_ref.so, built from /root/reference) on scripted host sessions and controller movements, and against
fixtures of those runs committed under tests/golden/ (context.npz, made by tests/golden/make_golden_context.py). CPU only."""
import ctypes
import os

import numpy as np
import pytest

from inputs import assert_bits_equal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "context.npz")


class Step(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("vectors", ctypes.c_int), ("playing", ctypes.c_int), ("pad", ctypes.c_int), ("ppq", ctypes.c_double),
                ("bpm", ctypes.c_double), ("sr", ctypes.c_double)]


class RefEvent(ctypes.Structure):
    _fields_ = [("type", ctypes.c_uint8), ("channel", ctypes.c_uint8), ("sourceIdx", ctypes.c_uint16), ("time", ctypes.c_int32),
                ("value1", ctypes.c_float), ("value2", ctypes.c_float)]


def host_session(seed, blocks=60, sr=48000.0):
    """[kind, vectors, playing, ppq, bpm, sr] rows: what a host application reports before each block (see tests/test_gpu_transport.py)."""
    rng = np.random.default_rng(seed)
    rows, ppq, bpm, playing = [], float(rng.uniform(-2, 8)), float(rng.uniform(60, 180)), False
    for b in range(blocks):
        r = rng.random()
        if r < 0.15:
            playing = not playing
        elif r < 0.25:
            bpm = float(rng.uniform(40, 220))
        elif r < 0.32:
            ppq = float(rng.uniform(-1, 16))
        elif r < 0.36:
            rows.append([0, 0, playing, float("nan") if rng.random() < 0.5 else float("inf"), bpm, sr])
        elif r < 0.40:
            rows.append([0, 0, playing, ppq, bpm, sr])
        elif r < 0.43:
            rows.append([2, 0, 0, 0.0, 0.0, 0.0])
        rows.append([0, 0, playing, ppq, bpm, sr])
        vectors = int(rng.choice([1, 2, 4, 8]))
        rows.append([1, vectors, 0, 0.0, 0.0, 0.0])
        if playing:
            ppq += vectors * 64 * bpm / 60.0 / sr
    return np.array(rows, np.float64)


def run_transport(lib, fn, rows):
    steps = (Step * len(rows))(*[Step(int(r[0]), int(r[1]), int(r[2]), 0, r[3], r[4], r[5]) for r in rows])
    frames = 64 * int(sum(r[1] for r in rows if int(r[0]) == 1))
    out, since = np.zeros(frames, np.float32), np.zeros(len(rows), np.uint64)
    f = getattr(lib, fn)
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.POINTER(Step), ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_uint64)]
    assert f(steps, len(rows), out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), since.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))) == 0
    return out, since


def controller_script(seed, n_vectors):
    """(values per vector [n_vectors], awake_from, events): a controller that moves now and then, several times inside one vector
    (the last value counts), by tiny and by huge amounts; the instrument's first event arrives at vector awake_from."""
    rng = np.random.default_rng(seed)
    awake_from = int(rng.integers(0, 6))
    values, events, cur = np.zeros(n_vectors, np.float32), [], np.float32(0.0)
    for t in range(n_vectors):
        if t == awake_from or (t > awake_from and rng.random() < 0.2):
            for when in sorted(rng.choice(64, int(rng.integers(1, 4)), replace=False)):     # distinct frames: the order is then the time order
                cur = np.float32(rng.choice([rng.random(), rng.random() * 1e-3, rng.uniform(-100, 100), 0.0, cur]))
                events.append((6, 1, 74, t * 64 + int(when), float(cur), 0.0))
        values[t] = cur
    events.sort(key=lambda e: e[3])
    # the value during vector t is that of the last event with time < 64 (t + 1)
    cur = np.float32(0.0)
    k = 0
    for t in range(n_vectors):
        while k < len(events) and events[k][3] < 64 * (t + 1):
            cur = np.float32(events[k][4])
            k += 1
        values[t] = cur
    return values, awake_from, events


def ref_controller(Lr, events, n_vectors, sr):
    fp = ctypes.POINTER(ctypes.c_float)
    Lr.e2s_ref_run_controllers.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                           ctypes.c_float, ctypes.c_int, ctypes.POINTER(RefEvent), ctypes.c_int, ctypes.c_int, ctypes.c_int, fp,
                                           ctypes.POINTER(ctypes.c_int), ctypes.c_int, fp]
    rows = np.zeros((8, 1, 64 * n_vectors), np.float32)
    ctl = np.zeros((1, 64 * n_vectors), np.float32)
    arr = (RefEvent * max(1, len(events)))(*[RefEvent(*e) for e in events])
    nums = (ctypes.c_int * 1)(74)
    assert Lr.e2s_ref_run_controllers(1, 0, 0, sr, 0.0, 0.0, 7.0, 24.0, 16, arr, len(events), 64 * n_vectors, 1, rows.ctypes.data_as(fp), nums, 1,
                                      ctl.ctypes.data_as(fp)) == 0
    return ctl[0]


def oracle_controller(oracle, values, awake_from, sr):
    f = oracle.lib.mlorc_smoothed_controller_run
    f.restype = ctypes.c_int
    fp = ctypes.POINTER(ctypes.c_float)
    f.argtypes = [ctypes.c_double, fp, ctypes.c_size_t, ctypes.c_size_t, fp]
    out = np.zeros(64 * len(values), np.float32)
    v = np.ascontiguousarray(values, np.float32)
    assert f(sr, v.ctypes.data_as(fp), len(v), awake_from, out.ctypes.data_as(fp)) == 0
    return out


def _ref_lib():
    so = os.path.join(ROOT, "oracle", "_ref", "libdropin_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libdropin_ref.so not available here")
    return ctypes.CDLL(so)


@pytest.mark.parametrize("sr", [48000.0, 44100.0, 96000.0])
def test_oracle_transport_vs_reference(oracle, sr):
    Lr = _ref_lib()
    for seed in range(12):
        rows = host_session(1000 * seed + int(sr) % 13, sr=sr)
        want, wsince = run_transport(Lr, "transport_ref_run", rows)
        got, gsince = run_transport(oracle.lib, "mlorc_transport_run", rows)
        assert_bits_equal(got, want, True, f"transport oracle, session {seed}, sr {sr}")
        assert np.array_equal(gsince, wsince)
    assert (np.diff(want) != 0).any()


@pytest.mark.parametrize("sr", [48000.0, 44100.0, 8000.0, 192000.0])
def test_oracle_smoothed_controller_vs_reference(oracle, sr):
    Lr = _ref_lib()
    for seed in range(10):
        values, awake_from, events = controller_script(seed, 80)
        want = ref_controller(Lr, events, 80, sr)
        got = oracle_controller(oracle, values, awake_from, sr)
        assert_bits_equal(got, want, True, f"smoothed controller oracle, script {seed}, sr {sr}")


def test_oracle_context_signals_vs_golden(oracle):
    """The committed outputs of the compiled reference (tests/golden/context.npz): what a box without /root/reference checks."""
    g = np.load(GOLDEN)
    for k in range(int(g["n_sessions"])):
        got, since = run_transport(oracle.lib, "mlorc_transport_run", g[f"session{k}"])
        assert_bits_equal(got, g[f"phase{k}"], True, f"golden transport session {k}")
        assert np.array_equal(since, g[f"since{k}"])
    for k in range(int(g["n_scripts"])):
        got = oracle_controller(oracle, g[f"values{k}"], int(g[f"awake{k}"]), float(g[f"sr{k}"]))
        assert_bits_equal(got, g[f"ctl{k}"], True, f"golden controller script {k}")
