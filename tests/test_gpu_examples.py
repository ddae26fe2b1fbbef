"""The reference's OWN example programs (examples/audio-and-midi/reverb.cpp — the Aaltoverb algorithm — and sine.cpp), included
unchanged from the reference checkout, compiled once against the reference (oracle/example_ref_*.cpp) and once against the
MI355X shim (tests/cpp/example_gpu_*.cpp): the same source gives the same bits. The two libraries are built where the
reference checkout exists and travel to the GPU box; nothing here reads the checkout at run time."""
import ctypes
import os

import numpy as np
import pytest

from inputs import assert_bits_equal, lcg_noise

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
c_f32p = ctypes.POINTER(ctypes.c_float)


def _libs():
    gpu_so = os.path.join(ROOT, "tests", "cpp", "libexamples_gpu.so")
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libexamples_ref.so")
    if not (os.path.exists(gpu_so) and os.path.exists(ref_so)):
        pytest.skip("the example libraries are built only where the reference checkout exists")
    Lg, Lr = ctypes.CDLL(gpu_so), ctypes.CDLL(ref_so)
    Lr.example_reverb_ref_run.argtypes = [ctypes.c_size_t, ctypes.c_size_t, c_f32p, c_f32p, c_f32p, c_f32p]
    Lg.example_reverb_gpu_run.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_char_p, ctypes.c_size_t]
    Lr.example_sine_ref_run.argtypes = [ctypes.c_size_t, c_f32p, c_f32p]
    Lg.example_sine_gpu_run.argtypes = [ctypes.c_size_t, ctypes.c_size_t, c_f32p, c_f32p, ctypes.c_char_p, ctypes.c_size_t]
    Lr.example_params_ref_run.argtypes = [ctypes.c_size_t, ctypes.c_size_t, c_f32p, c_f32p, c_f32p, c_f32p]
    Lg.example_params_gpu_run.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_char_p, ctypes.c_size_t]
    return Lg, Lr


def test_example_libraries_load_where_built():
    gpu_so = os.path.join(ROOT, "tests", "cpp", "libexamples_gpu.so")
    if not os.path.exists(gpu_so):
        pytest.skip("not built here")
    L = ctypes.CDLL(gpu_so)
    assert hasattr(L, "example_reverb_gpu_run") and hasattr(L, "example_sine_gpu_run") and hasattr(L, "example_params_gpu_run")


@pytest.mark.gpu
@pytest.mark.parametrize("launches", [1, 4])
def test_reference_reverb_example_same_source_same_bits(launches):
    """reverb.cpp: two LinearGlide-smoothed parameters from host Projections and powf, ten Allpass<PitchbendableDelay>, two
    PitchbendableDelays, stereo feedback kept in DSPVector members — 48 reverbs per launch on the GPU."""
    Lg, Lr = _libs()
    V, T = 48, 80
    in0 = lcg_noise(np.arange(V, dtype=np.uint32) + 3, 64 * T) * np.float32(0.25)
    in1 = lcg_noise(np.arange(V, dtype=np.uint32) + 2003, 64 * T) * np.float32(0.25)
    in0[:, 64 * 30:] = 0
    in1[:, 64 * 30:] = 0
    want0, want1 = np.zeros_like(in0), np.zeros_like(in0)
    assert Lr.example_reverb_ref_run(V, T, in0.ctypes.data_as(c_f32p), in1.ctypes.data_as(c_f32p), want0.ctypes.data_as(c_f32p), want1.ctypes.data_as(c_f32p)) == 0
    got0, got1 = np.zeros_like(in0), np.zeros_like(in0)
    err = ctypes.create_string_buffer(4096)
    st = Lg.example_reverb_gpu_run(V, T, launches, in0.ctypes.data_as(c_f32p), in1.ctypes.data_as(c_f32p), got0.ctypes.data_as(c_f32p), got1.ctypes.data_as(c_f32p), err, 4096)
    assert st == 0, err.value.decode()
    assert_bits_equal(got0, want0, True, "Aaltoverb left")
    assert_bits_equal(got1, want1, True, "Aaltoverb right")
    assert np.abs(want0[:, 64 * 60:]).max() > 1e-4      # the tail still rings 30 vectors after the input stopped


@pytest.mark.gpu
def test_reference_sine_example_same_source_same_bits():
    Lg, Lr = _libs()
    V, T = 70, 12
    want0, want1 = np.zeros(64 * T, np.float32), np.zeros(64 * T, np.float32)
    assert Lr.example_sine_ref_run(T, want0.ctypes.data_as(c_f32p), want1.ctypes.data_as(c_f32p)) == 0
    got0, got1 = np.zeros((V, 64 * T), np.float32), np.zeros((V, 64 * T), np.float32)
    err = ctypes.create_string_buffer(4096)
    st = Lg.example_sine_gpu_run(V, T, got0.ctypes.data_as(c_f32p), got1.ctypes.data_as(c_f32p), err, 4096)
    assert st == 0, err.value.decode()
    for v in (0, 1, V - 1):
        assert_bits_equal(got0[v], want0, True, f"sine example left, voice {v}")
        assert_bits_equal(got1[v], want1, True, f"sine example right, voice {v}")


@pytest.mark.gpu
def test_reference_params_example_same_source_same_bits():
    """params.cpp: a SignalProcessor subclass whose process function reads freq1 / freq2 (log ranges) / gain from its
    ParameterTree every vector (a run-time Path and two hashed ones). The shim defers to madronalib's own host-side
    parameter layer; the captured kernel takes parameter changes through VoiceProgram::update(). Four settings in a row, the
    oscillators' phases carried across the changes."""
    Lg, Lr = _libs()
    V, S, T = 70, 4, 9
    steps = np.array([[0, 0, 0], [0.3, 0.9, 1.0], [1.0, 0.0, 0.5], [0.51, 0.49, 0.25]], np.float32)
    want0, want1, wr = np.zeros(64 * S * T, np.float32), np.zeros(64 * S * T, np.float32), np.zeros((S, 3), np.float32)
    assert Lr.example_params_ref_run(S, T, steps.ctypes.data_as(c_f32p), want0.ctypes.data_as(c_f32p), want1.ctypes.data_as(c_f32p), wr.ctypes.data_as(c_f32p)) == 0
    got0, got1, gr = np.zeros((V, 64 * S * T), np.float32), np.zeros((V, 64 * S * T), np.float32), np.zeros((S, 3), np.float32)
    err = ctypes.create_string_buffer(4096)
    st = Lg.example_params_gpu_run(V, S, T, steps.ctypes.data_as(c_f32p), got0.ctypes.data_as(c_f32p), got1.ctypes.data_as(c_f32p), gr.ctypes.data_as(c_f32p), err, 4096)
    assert st == 0, err.value.decode()
    assert_bits_equal(gr, wr, True, "real parameter values (the log and linear projections are madronalib's own on both sides)")
    assert wr[0, 1] == np.float32(633.95734) or abs(wr[0, 1] - 633.957) < 1e-2      # freq2 at normalized 0.6 of 40..4000 Hz, log
    for v in (0, 1, V - 1):
        assert_bits_equal(got0[v], want0, True, f"params example left, voice {v}")
        assert_bits_equal(got1[v], want1, True, f"params example right, voice {v}")
    assert np.abs(want0).max() > 0.05


class _CtlEv(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int), ("channel", ctypes.c_int), ("sourceIdx", ctypes.c_int), ("time", ctypes.c_int),
                ("value1", ctypes.c_float), ("value2", ctypes.c_float)]


@pytest.mark.gpu
def test_reference_controllers_to_audio_example_unchanged():
    """controllers-to-audio.cpp, the example that reads a SAMPLE of a signal into a host float: `float freqInHz =
    ctrlToFreq(ctrlSig[0])` (a std::function projection, 110 * 4^x) per DSPVector, then SineGen(freqInHz / sr). Compiled
    unchanged against the shim and run through gpu::VoiceProgramOptions::hostContextSamples - controller signals made on the
    device, their samples fetched, the process function re-run on the host per DSPVector, the captured kernel launched - it gives
    the reference's floats: 300 vectors of eight sines following eight moving MIDI controllers and a volume controller."""
    Lg, Lr = _libs()
    Lr.example_controllers_ref_run.argtypes = [ctypes.POINTER(_CtlEv), ctypes.c_int, ctypes.c_int, c_f32p, c_f32p]
    Lg.example_controllers_gpu_run.argtypes = [ctypes.POINTER(_CtlEv), ctypes.c_int, ctypes.c_int, c_f32p, c_f32p, ctypes.c_char_p, ctypes.c_size_t]
    CTRL, T = 6, 300
    numbers = [19, 23, 27, 31, 49, 53, 57, 61]
    rng = np.random.default_rng(77)
    evs = [(CTRL, 1, 62, 5, 0.8, 0.0)]
    t = 40
    while t < 64 * T:
        evs.append((CTRL, int(rng.integers(1, 17)), int(rng.choice(numbers + [62])), t, float(np.float32(rng.random())), 0.0))
        t += int(rng.integers(1, 900))
    arr = (_CtlEv * len(evs))(*[_CtlEv(*e) for e in evs])
    want = [np.zeros(64 * T, np.float32) for _ in range(2)]
    assert Lr.example_controllers_ref_run(arr, len(evs), T, want[0].ctypes.data_as(c_f32p), want[1].ctypes.data_as(c_f32p)) == 0
    got = [np.zeros(64 * T, np.float32) for _ in range(2)]
    err = ctypes.create_string_buffer(4096)
    st = Lg.example_controllers_gpu_run(arr, len(evs), T, got[0].ctypes.data_as(c_f32p), got[1].ctypes.data_as(c_f32p), err, 4096)
    assert st == 0, err.value.decode()
    assert_bits_equal(got[0], want[0], True, "controllers-to-audio left")
    assert_bits_equal(got[1], want[1], True, "controllers-to-audio right")
    assert np.abs(want[0]).max() > 0.2
    # the oscillators really do move with their controllers: the spectrum centroid of the first and last second differ
    assert not np.allclose(want[0][:4096], want[0][-4096:])


@pytest.mark.gpu
def test_reference_fdtd_example_unchanged():
    """fdtd.cpp - the last of the reference's DSP examples. Its process function is not a DSPVector graph: per SAMPLE it reads
    `freq[i]` and `inputVec[i]` into floats, steps a 16 x 16 finite-difference mesh in host code and writes `outLVec[i]`, inside a
    flush-denormals scope. Compiled unchanged against the shim it runs as the reference runs it, called once per DSPVector, in
    immediate mode: ImpulseGen, SineGen and the vector arithmetic on the device, the mesh the program's own loop. 1600 DSPVectors
    (two ticks of the 2 Hz impulse train with the mesh ringing in between): the reference's floats."""
    Lg, Lr = _libs()
    Lr.example_fdtd_ref_run.argtypes = [ctypes.c_size_t, c_f32p, c_f32p]
    Lg.example_fdtd_gpu_run.argtypes = [ctypes.c_size_t, c_f32p, c_f32p, ctypes.c_char_p, ctypes.c_size_t]
    T = 1600
    want = [np.zeros(64 * T, np.float32) for _ in range(2)]
    assert Lr.example_fdtd_ref_run(T, want[0].ctypes.data_as(c_f32p), want[1].ctypes.data_as(c_f32p)) == 0
    got = [np.zeros(64 * T, np.float32) for _ in range(2)]
    err = ctypes.create_string_buffer(4096)
    st = Lg.example_fdtd_gpu_run(T, got[0].ctypes.data_as(c_f32p), got[1].ctypes.data_as(c_f32p), err, 4096)
    assert st == 0, err.value.decode()
    assert_bits_equal(got[0], want[0], True, "fdtd left pickup")
    assert_bits_equal(got[1], want[1], True, "fdtd right pickup")
    assert np.abs(want[0]).max() > 1e-3 and np.isfinite(want[0]).all()
    assert np.abs(want[0][64 * 1000:]).max() > 0        # still ringing (or struck again) late in the run
