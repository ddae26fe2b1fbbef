"""Immediate mode, the app layer: oracle/dropin_ref.cpp's host loops for the reference's AudioContext / EventsToSignals / Synth /
SignalProcessBuffer - the way a plug-in wrapper or a test steps those objects by hand - compiled against the shim
(tests/cpp/libdropin_imm.so; MLSynth.h, MLSignalProcessBuffer.h ... are forwarded by include/mlgpu/compat) and run call by call on
the device, against the same loops compiled against the reference (oracle/_ref/libdropin_ref.so). Same exported names, same
arguments, the reference's bits. One instrument each: the banks of instruments are gpu::SynthProgram's (tests/test_gpu_dropin.py)."""
import ctypes
import os

import numpy as np
import pytest

from inputs import assert_bits_equal

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
c_f32p = ctypes.POINTER(ctypes.c_float)
c_ip = ctypes.POINTER(ctypes.c_int)


class _Ev(ctypes.Structure):
    _fields_ = [("type", ctypes.c_uint8), ("channel", ctypes.c_uint8), ("sourceIdx", ctypes.c_uint16), ("time", ctypes.c_int32),
                ("value1", ctypes.c_float), ("value2", ctypes.c_float)]


def _libs():
    import madronalib_amd as ml
    if ml.device_count() == 0:
        pytest.skip("no GPU")
    ref = os.path.join(ROOT, "oracle", "_ref", "libdropin_ref.so")
    imm = os.path.join(ROOT, "tests", "cpp", "libdropin_imm.so")
    if not os.path.exists(ref) or not os.path.exists(imm):
        pytest.skip("oracle/_ref/libdropin_ref.so (the compiled reference) or tests/cpp/libdropin_imm.so not built here")
    return ctypes.CDLL(ref), ctypes.CDLL(imm)


def _p(a):
    return a.ctypes.data_as(c_f32p)


def _events(kind, seed, frames, polyphony):
    from test_gpu_events import performance
    evs = performance(kind, seed, frames, polyphony)
    return (_Ev * max(1, len(evs)))(*[_Ev(*e) for e in evs]), len(evs)


@pytest.mark.parametrize("which", ["controller", "tempo", "lean"])
def test_synth_subclasses_stepped_by_a_host_loop(which):
    """tests/cpp/dropin_synth.h: Synth subclasses whose processVoice reads the voice rows (lean: pitch and gate only), a smoothed
    controller, the beat phase; Synth::processVector's voice loop and sum, AudioContext::processVector per DSPVector, updateTime before
    every block, the envelope knobs turned half way through."""
    Lr, Li = _libs()
    block, n_blocks = 256, 12
    S = block * n_blocks
    arr, n = _events("midi", 4100 + len(which), S, 6)
    out = {}
    for tag, L in (("reference", Lr), ("immediate", Li)):
        f = getattr(L, which + "_synth_ref_run")
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.POINTER(_Ev), ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int, c_f32p, c_f32p]
        l, r = np.zeros(S, np.float32), np.zeros(S, np.float32)
        assert f(arr, n, 0.012, 0.6, block, n_blocks, _p(l), _p(r)) == 0
        out[tag] = (l, r)
    assert_bits_equal(out["immediate"][0], out["reference"][0], True, f"{which}: left")
    assert_bits_equal(out["immediate"][1], out["reference"][1], True, f"{which}: right")
    assert np.abs(out["reference"][0]).max() > 1e-3


def test_plugin_flow_through_signal_process_buffer():
    """Host blocks of 1 ... 512 frames through SignalProcessBuffer: the DSPVectors fall due when the output ring runs short, the events
    of a block carry block-relative times (and are dropped unprocessed when no DSPVector falls due in their block), the host's time
    report before every block; PluginSynth reads voice rows, a controller and the beat phase."""
    Lr, Li = _libs()
    rng = np.random.default_rng(5)
    sizes = [64, 100, 37, 512, 1, 200, 64, 333, 17, 480, 31, 33, 128, 7, 250] + [int(x) for x in rng.integers(1, 513, 12)]
    S = int(np.sum(sizes))
    blocks = (ctypes.c_int * len(sizes))(*sizes)
    arr, n = _events("midi", 1201, S, 6)
    out = {}
    for tag, L in (("reference", Lr), ("immediate", Li)):
        L.plugin_ref_run.restype = ctypes.c_int
        L.plugin_ref_run.argtypes = [ctypes.POINTER(_Ev), ctypes.c_int, ctypes.c_float, ctypes.c_float, c_ip, ctypes.c_int, ctypes.c_int, c_f32p, c_f32p]
        l, r = np.zeros(S, np.float32), np.zeros(S, np.float32)
        assert L.plugin_ref_run(arr, n, 0.01, 0.3, blocks, len(sizes), 512, _p(l), _p(r)) == 0
        out[tag] = np.stack([l, r])
    assert_bits_equal(out["immediate"], out["reference"], True, "plug-in flow")
    assert np.abs(out["reference"]).max() > 1e-3


def test_signal_process_buffer_with_audio_inputs():
    """spb_ref_run: one audio input, a stateful Lopass and a gain, host blocks of assorted sizes: latency, zero fill and ring
    behaviour of the adaptor."""
    Lr, Li = _libs()
    sizes = [64, 1, 63, 128, 100, 7, 512, 300, 64, 64, 5, 250]
    S = int(np.sum(sizes))
    blocks = (ctypes.c_int * len(sizes))(*sizes)
    x = np.random.default_rng(3).uniform(-1, 1, S).astype(np.float32)
    out = {}
    for tag, L in (("reference", Lr), ("immediate", Li)):
        L.spb_ref_run.restype = ctypes.c_int
        L.spb_ref_run.argtypes = [ctypes.c_int, c_ip, ctypes.c_int, c_f32p, c_f32p, c_f32p]
        o0, o1 = np.zeros(S, np.float32), np.zeros(S, np.float32)
        assert L.spb_ref_run(512, blocks, len(sizes), _p(x), _p(o0), _p(o1)) == 0
        out[tag] = np.stack([o0, o1])
    assert_bits_equal(out["immediate"], out["reference"], True, "SignalProcessBuffer")
    assert np.abs(out["reference"]).max() > 1e-2


def test_controllers_to_audio_process_function_stepped_by_hand():
    """tests/cpp/dropin_controllers.h (the reference's controllers-to-audio example in vector form): ctx->getInputController(n) for a
    handful of controllers, read sample 0 of each into a host float, drive sine generators."""
    Lr, Li = _libs()
    block, n_blocks = 512, 8
    S = block * n_blocks
    rng = np.random.default_rng(3003)
    numbers = [19, 23, 27, 31, 49, 53, 57, 61, 62]   # eight tunings and the volume
    evs, t = [(6, 1, 62, int(rng.integers(0, 100)), float(np.float32(rng.uniform(0.3, 1.0))), 0.0)], int(rng.integers(0, 300))
    while t < S:
        evs.append((6, int(rng.integers(1, 17)), int(rng.choice(numbers)), t, float(np.float32(rng.random())), 0.0))
        t += int(rng.integers(1, 700))
    evs.sort(key=lambda e: e[3])
    arr, n = (_Ev * len(evs))(*[_Ev(*e) for e in evs]), len(evs)
    out = {}
    for tag, L in (("reference", Lr), ("immediate", Li)):
        L.ctl_audio_ref_run.restype = ctypes.c_int
        L.ctl_audio_ref_run.argtypes = [ctypes.POINTER(_Ev), ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p]
        o = np.zeros(S, np.float32)
        assert L.ctl_audio_ref_run(arr, n, block, n_blocks, _p(o)) == 0
        out[tag] = o
    assert_bits_equal(out["immediate"], out["reference"], True, "controllers to audio")
    assert np.abs(out["reference"]).max() > 1e-3


@pytest.mark.parametrize("seed", [1, 2])
def test_transport_of_an_immediate_context(seed):
    """AudioContext::updateTime / processVector / getBeatPhase / getTimeInfo().samplesSinceStart / clear on the scripted host sessions
    of tests/test_gpu_transport.py (start, stop, tempo changes, relocation, repeated and NaN reports)."""
    Lr, Li = _libs()
    from test_gpu_transport import Step, host_session
    script = host_session(seed)
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
    out = {}
    for tag, L in (("reference", Lr), ("immediate", Li)):
        L.transport_ref_run.restype = ctypes.c_int
        L.transport_ref_run.argtypes = [ctypes.POINTER(Step), ctypes.c_int, c_f32p, ctypes.POINTER(ctypes.c_uint64)]
        o, since = np.zeros(frames, np.float32), np.zeros(len(steps), np.uint64)
        assert L.transport_ref_run(arr, len(steps), _p(o), since.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))) == 0
        out[tag] = (o, since)
    assert_bits_equal(out["immediate"][0], out["reference"][0], True, "beat phase")
    assert (out["immediate"][1] == out["reference"][1]).all()
    assert np.abs(out["reference"][0]).max() > 0


def test_synth_with_a_published_signal_read_by_the_ui_side():
    """SmallSynth (tests/cpp/dropin_synth.h) stores a two-channel "scope" signal per voice with storePublishedSignal; the host reads up to
    700 frames after every block (PublishedSignal::read: sometimes all there is, sometimes not). Outside a capture the published signal
    owns its ring (mlgpu_published_signal) and every voice's DSPVectorArray goes in as the reference's writeQuick puts it: audio, the
    scope's floats and the counts of every read against the reference."""
    Lr, Li = _libs()
    block, n_blocks, scope_read = 512, 10, 700
    S = block * n_blocks
    arr, n = _events("midi", 917, S, 6)
    c_szp = ctypes.POINTER(ctypes.c_size_t)
    out = {}
    for tag, L in (("reference", Lr), ("immediate", Li)):
        L.synth_ref_run.restype = ctypes.c_int
        L.synth_ref_run.argtypes = [ctypes.POINTER(_Ev), ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int, c_f32p, c_f32p,
                                    c_f32p, c_szp, ctypes.c_int]
        l, r = np.zeros(S, np.float32), np.zeros(S, np.float32)
        scope, counts = np.zeros(n_blocks * scope_read * 2, np.float32), np.zeros(n_blocks, np.uint64)
        assert L.synth_ref_run(arr, n, 0.012, 0.6, block, n_blocks, _p(l), _p(r), _p(scope), counts.ctypes.data_as(c_szp), scope_read) == 0
        out[tag] = (np.stack([l, r]), scope, counts)
    assert_bits_equal(out["immediate"][0], out["reference"][0], True, "audio")
    assert (out["immediate"][2] == out["reference"][2]).all(), (out["immediate"][2], out["reference"][2])
    assert_bits_equal(out["immediate"][1], out["reference"][1], True, "scope floats")
    assert out["reference"][2].sum() > 0 and np.abs(out["reference"][1]).max() > 0


@pytest.mark.parametrize("octaves,max_frames", [(0, 64), (2, 64), (3, 16), (6, 4)])
def test_published_signal_written_and_read_by_hand(octaves, max_frames):
    """SignalProcessor::PublishedSignal on its own: writeQuick of three voices in rotation interleaved with read / readLatest /
    peekLatest of assorted sizes (tests/test_gpu_published.py's scripts), floats and counts against the reference's object."""
    Lr, Li = _libs()
    from test_gpu_published import script, reference
    from inputs import lcg_noise
    nV, T = 3, 12
    ch0 = lcg_noise(np.arange(nV, dtype=np.uint32) + 9, 64 * T)
    ch1 = lcg_noise(np.arange(nV, dtype=np.uint32) + 4009, 64 * T)
    ops, args = script(T, seed=octaves)
    res = {}
    for tag, L in (("reference", Lr), ("immediate", Li)):
        L.published_ref_run.restype = ctypes.c_int
        L.published_ref_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, c_f32p, c_f32p, c_ip, c_ip, ctypes.c_int,
                                        c_f32p, ctypes.POINTER(ctypes.c_size_t)]
        res[tag] = reference(L, max_frames, nV, octaves, ch0, ch1, ops, args)
    assert (res["immediate"][1] == res["reference"][1]).all()
    assert_bits_equal(res["immediate"][0], res["reference"][0], True, "published floats")
    assert res["reference"][1].sum() > 0
