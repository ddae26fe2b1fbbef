"""mlgpu_transport (AudioContext::ProcessTime: updateTime / processVector / getBeatPhase) against the reference's own AudioContext
(oracle/_ref/libdropin_ref.so: transport_ref_run) on scripted host behaviour: bit-exact."""
import ctypes
import os

import numpy as np
import pytest

from inputs import assert_bits_equal
from madronalib_amd.constants import Proc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


class Step(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("vectors", ctypes.c_int), ("playing", ctypes.c_int), ("pad", ctypes.c_int), ("ppq", ctypes.c_double),
                ("bpm", ctypes.c_double), ("sr", ctypes.c_double)]


def update(ppq, bpm, playing, sr=48000.0):
    return ("update", ppq, bpm, playing, sr)


def ref_run(script):
    so = os.path.join(ROOT, "oracle", "_ref", "libdropin_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libdropin_ref.so not available here")
    L = ctypes.CDLL(so)
    steps = []
    for s in script:
        if s[0] == "update":
            steps.append(Step(0, 0, int(s[3]), 0, s[1], s[2], s[4]))
        elif s[0] == "process":
            steps.append(Step(1, s[1], 0, 0, 0, 0, 0))
        else:
            steps.append(Step(2, 0, 0, 0, 0, 0, 0))
    arr = (Step * len(steps))(*steps)
    frames = 64 * sum(s[1] for s in script if s[0] == "process")
    out = np.zeros(frames, np.float32)
    since = np.zeros(len(steps), np.uint64)
    L.transport_ref_run.argtypes = [ctypes.POINTER(Step), ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_uint64)]
    assert L.transport_ref_run(arr, len(steps), out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), since.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))) == 0
    return out, since


def gpu_run(tr, script, index):
    outs, since = [], []
    for s in script:
        if s[0] == "update":
            tr.update_time(s[1], s[2], s[3], s[4], index)
        elif s[0] == "process":
            outs.append(tr.process_host(s[1]))
        else:
            tr.clear(index)
        since.append(tr.samples_since_start(0 if index is None else index))
    return np.concatenate(outs, 1), np.array(since, np.uint64)


def host_session(seed, blocks=40, sr=48000.0):
    """A host application: reports position and tempo before every block (block sizes vary), starts, stops, loops back, changes
    tempo, relocates; now and then reports rubbish (NaN, infinity), twice in a row, or a negative position (count-in)."""
    rng = np.random.default_rng(seed)
    script, ppq, bpm, playing = [], float(rng.uniform(-2, 8)), float(rng.uniform(60, 180)), False
    for b in range(blocks):
        r = rng.random()
        if r < 0.15:
            playing = not playing
        elif r < 0.25:
            bpm = float(rng.uniform(40, 220))
        elif r < 0.32:
            ppq = float(rng.uniform(-1, 16))              # the user moved the playhead / the loop jumped back
        elif r < 0.36:
            script.append(update(float("nan") if rng.random() < 0.5 else float("inf"), bpm, playing, sr))
        elif r < 0.40:
            script.append(update(ppq, bpm, playing, sr))   # the same report twice: no samples in between
        elif r < 0.43:
            script.append(("clear",))
        script.append(update(ppq, bpm, playing, sr))
        vectors = int(rng.choice([1, 2, 4, 8]))
        script.append(("process", vectors))
        if playing:
            ppq += vectors * 64 * bpm / 60.0 / sr
    return script


@pytest.fixture(scope="module")
def eng():
    import madronalib_amd as ml
    e = ml.Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("sr", [48000.0, 44100.0, 96000.0])
def test_transport_matches_reference(eng, sr):
    """Five contexts driven by five different host sessions (each context addressed by index), one launch per host block."""
    import madronalib_amd as ml
    N = 5
    scripts = [host_session(11 * k + int(sr) % 7, sr=sr) for k in range(N)]
    # the same block structure for all (one device launch covers every context): take the process steps of script 0
    procs = [s for s in scripts[0] if s[0] == "process"]
    aligned = []
    for sc in scripts:
        it, out = iter(procs), []
        for s in sc:
            out.append(next(it) if s[0] == "process" else s)
        aligned.append(out)
    tr = ml.Transport(eng, N, 8)
    want = [ref_run(sc) for sc in aligned]
    # interleave: all contexts' reports of a block, then one process call
    pos = [0] * N
    outs = []
    since = [[] for _ in range(N)]
    for p in procs:
        for k in range(N):
            while aligned[k][pos[k]][0] != "process":
                s = aligned[k][pos[k]]
                if s[0] == "update":
                    tr.update_time(s[1], s[2], s[3], s[4], k)
                else:
                    tr.clear(k)
                pos[k] += 1
            pos[k] += 1
        outs.append(tr.process_host(p[1]))
        for k in range(N):
            since[k].append(tr.samples_since_start(k))
    got = np.concatenate(outs, 1)
    moving = 0
    for k in range(N):
        assert_bits_equal(got[k], want[k][0], True, f"sr {sr}: beat phase of context {k}")
        ref_since = want[k][1][[i for i, s in enumerate(aligned[k]) if s[0] == "process"]]
        assert np.array_equal(np.array(since[k], np.uint64), ref_since), f"samplesSinceStart of context {k}"
        moving += int((np.diff(want[k][0]) > 0).any())
    assert moving >= 3
    tr.close()


def test_transport_all_contexts_and_long_run(eng):
    """MLGPU_TRANSPORT_ALL: one host application behind every context; a steady 30 s run at 120 bpm in 8-vector launches (the
    phasor wraps 60 times; the float accumulation drifts exactly as the reference's does)."""
    import madronalib_amd as ml
    sr, bpm, T = 48000.0, 120.0, 8
    script, ppq = [], 0.0
    for b in range(2813):
        script.append(update(ppq, bpm, True, sr))
        script.append(("process", T))
        ppq += T * 64 * bpm / 60.0 / sr
    want, since = ref_run(script)
    tr = ml.Transport(eng, 300, T)
    got, gsince = gpu_run(tr, script, None)
    for k in (0, 17, 299):
        assert_bits_equal(got[k], want, True, f"context {k}")
    assert np.array_equal(gsince, since)
    assert (np.diff(want) < -0.5).sum() >= 59
    tr.close()


def test_transport_error_paths(eng):
    import madronalib_amd as ml
    with pytest.raises(ml.MlgpuError):
        ml.Transport(eng, 0, 4)
    with pytest.raises(ml.MlgpuError):
        ml.Transport(eng, 4, 0)
    tr = ml.Transport(eng, 4, 2)
    with pytest.raises(ml.MlgpuError):
        tr.update_time(0.0, 120.0, True, 48000.0, 4)
    with pytest.raises(ml.MlgpuError):
        tr.clear(9)
    with pytest.raises(ml.MlgpuError) as ei:
        tr.process(3)
    assert ei.value.status == ml.Status.ERR_RANGE
    assert not tr.process_host(2).any()          # never reported: omega_{0}, dpdt_{0}
    tr.close()


def test_transport_survives_a_longer_reservation(eng):
    """mlgpu_transport_reserve changes the launch length; the phasors go on."""
    import madronalib_amd as ml
    script = [update(0.25, 133.0, True), ("process", 2), ("process", 2), ("process", 16), update(3.9, 90.0, True), ("process", 16), ("process", 1)]
    want, _ = ref_run(script)
    tr = ml.Transport(eng, 3, 2)
    outs = []
    for s in script:
        if s[0] == "update":
            tr.update_time(s[1], s[2], s[3], s[4])
        else:
            if s[1] > 2:
                tr.reserve(16)
            outs.append(tr.process_host(s[1]))
    got = np.concatenate(outs, 1)
    for k in range(3):
        assert_bits_equal(got[k], want, True, f"context {k} across a re-reservation")
    tr.close()


def test_signal_buffers_do_not_move_under_recorded_sequences(eng):
    """The beat-phase and controller signals are buffers whose device pointers callers hold and recorded sequences replay:
    a shorter reservation keeps the allocation (same pointer), a longer one is refused while a sequence of the engine is alive
    (afterwards it goes through); an events object destroyed meanwhile is gone for its owner and freed with the last sequence."""
    import madronalib_amd as ml
    tr = ml.Transport(eng, 4, 8)
    p0 = tr.beat_phase
    tr.reserve(2)
    assert tr.beat_phase == p0                      # shrinking never reallocates
    tr.reserve(8)
    assert tr.beat_phase == p0
    ev = ml.Events(eng, 4, 2, 48000.0)
    ev.watch_controllers([7, 11], 4)
    c0 = ev.controller_signal(0)
    ev.watch_controllers([7, 11], 2)
    assert ev.controller_signal(0) == c0
    bank = eng.bank([Proc.SINE_GEN], 64)
    bank.set_input_const(np.full(64, 0.01, np.float32))
    out = eng.alloc(4 * 64 * 64)
    with eng.record() as seq:
        bank.process(1, out)
    try:
        with pytest.raises(ml.MlgpuError) as ei:
            tr.reserve(32)
        assert ei.value.status == ml.Status.ERR_INVALID and "sequences" in str(ei.value)
        with pytest.raises(ml.MlgpuError):
            ev.watch_controllers([7, 11], 64)
        with pytest.raises(ml.MlgpuError):
            ev.watch_controllers([7, 11, 12], 4)     # another set of controllers would free the old signals
        # destroying an events object under a live sequence goes through for the caller (round 4: it used to be refused, and an
        # owner that ignores the status - a destructor - leaked the object); its memory lives until the last sequence is gone
        doomed = ml.Events(eng, 2, 2, 48000.0)
        doomed.watch_controllers([7], 2)
        assert doomed.L.mlgpu_events_destroy(doomed.h) == ml.Status.OK
        doomed.h = None
        seq.launch()                                  # (the sequence still replays)
        eng.sync()
    finally:
        seq.close()                                   # ... and here the deferred object is freed
    tr.reserve(32)
    assert tr.process_host(32).shape[1] == 32 * 64
    ev.watch_controllers([7, 11], 64)
    ev.close()
    tr.close()
    bank.close()
