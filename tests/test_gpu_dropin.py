"""Source-level drop-in: tests/cpp/dropin_patch.h is user code written against madronalib's public API. It is
compiled UNCHANGED twice — against the reference's own headers + AudioContext (oracle/_ref/libdropin_ref.so, one
voice per state object, process function called once per 64 frames) and against include/mlgpu/compat (captured
once into a fused gfx950 kernel, V voices per launch). Outputs must be identical bit for bit."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from inputs import assert_bits_equal, gate_signal, lcg_noise

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
c_f32p = ctypes.POINTER(ctypes.c_float)


def _gpu_lib():
    so = os.path.join(ROOT, "tests", "cpp", "libdropin_gpu.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "include", "mlgpu")], stdout=subprocess.DEVNULL)
    L = ctypes.CDLL(so)
    L.dropin_gpu_run.restype = ctypes.c_int
    L.dropin_gpu_run.argtypes = [ctypes.c_size_t, ctypes.c_size_t, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_char_p, ctypes.c_size_t]
    L.dropin_gpu_run_mixed.restype = ctypes.c_int
    L.dropin_gpu_run_mixed.argtypes = L.dropin_gpu_run.argtypes
    L.plate_gpu_run.restype = ctypes.c_int
    L.plate_gpu_run.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_size_t, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_char_p,
                                ctypes.c_size_t]
    L.decay_gpu_run.restype = ctypes.c_int
    L.decay_gpu_run.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, c_f32p, c_f32p, c_f32p, ctypes.POINTER(ctypes.c_int), ctypes.c_char_p, ctypes.c_size_t]
    L.oversample_gpu_run.restype = ctypes.c_int
    L.oversample_gpu_run.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_char_p, ctypes.c_size_t]
    return L


def _ref_lib():
    so = os.path.join(ROOT, "oracle", "_ref", "libdropin_ref.so")
    if os.path.isdir("/root/reference/source/DSP"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "_ref/libdropin_ref.so"], stdout=subprocess.DEVNULL)
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libdropin_ref.so not available here")
    L = ctypes.CDLL(so)
    L.dropin_ref_run.restype = ctypes.c_int
    L.dropin_ref_run.argtypes = [ctypes.c_size_t, ctypes.c_size_t, c_f32p, c_f32p, c_f32p, c_f32p]
    L.plate_ref_run.restype = ctypes.c_int
    L.plate_ref_run.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, c_f32p, c_f32p, c_f32p, c_f32p]
    L.decay_ref_run.restype = ctypes.c_int
    L.decay_ref_run.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, c_f32p, c_f32p, c_f32p]
    L.oversample_ref_run.restype = ctypes.c_int
    L.oversample_ref_run.argtypes = [ctypes.c_size_t, ctypes.c_size_t, c_f32p, c_f32p, c_f32p, c_f32p]
    return L


def _inputs(V, T):
    gate = gate_signal(V, 64 * T, seed=11)
    rng = np.random.default_rng(5)
    base = rng.uniform(-1.0, 3.0, V).astype(np.float32)
    vib = (0.02 * np.sin(np.arange(64 * T)[None, :] * 0.002 * (1 + np.arange(V)[:, None] % 4))).astype(np.float32)
    return gate, np.ascontiguousarray(base[:, None] + vib, np.float32)


def test_user_code_compiles_against_both_and_fails_loudly_without_gpu():
    import madronalib_amd as ml
    L = _gpu_lib()
    _ref_lib()
    if ml.device_count() > 0:
        pytest.skip("a GPU is visible here")
    V, T = 4, 2
    gate, pitch = _inputs(V, T)
    o0, o1 = np.zeros_like(gate), np.zeros_like(gate)
    err = ctypes.create_string_buffer(512)
    st = L.dropin_gpu_run(V, T, gate.ctypes.data_as(c_f32p), pitch.ctypes.data_as(c_f32p), o0.ctypes.data_as(c_f32p),
                          o1.ctypes.data_as(c_f32p), err, 512)
    assert st == ml.Status.ERR_NO_DEVICE, (st, err.value)


@pytest.mark.gpu
def test_same_source_same_bits():
    Lg, Lr = _gpu_lib(), _ref_lib()
    V, T = 384, 40
    gate, pitch = _inputs(V, T)
    want0, want1 = np.zeros_like(gate), np.zeros_like(gate)
    assert Lr.dropin_ref_run(V, T, gate.ctypes.data_as(c_f32p), pitch.ctypes.data_as(c_f32p), want0.ctypes.data_as(c_f32p),
                             want1.ctypes.data_as(c_f32p)) == 0
    got0, got1 = np.zeros_like(gate), np.zeros_like(gate)
    err = ctypes.create_string_buffer(2048)
    st = Lg.dropin_gpu_run(V, T, gate.ctypes.data_as(c_f32p), pitch.ctypes.data_as(c_f32p), got0.ctypes.data_as(c_f32p),
                           got1.ctypes.data_as(c_f32p), err, 2048)
    assert st == 0, err.value.decode()
    assert_bits_equal(got0, want0, True, "drop-in patch output 0")
    assert_bits_equal(got1, want1, True, "drop-in patch output 1")
    assert np.abs(want0).max() > 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("V", [384, 1000])
def test_same_source_all_voices_mixed(oracle, V):
    """VoiceProgramOptions::mixOutputs: the same unchanged process function, its two outputs as the sum of all voices made inside the
    voice kernel - against the per-voice outputs of the same program added up in mlgpu_mixdown's order (the oracle's restatement)."""
    Lg = _gpu_lib()
    T = 12
    gate, pitch = _inputs(V, T)
    per0, per1 = np.zeros_like(gate), np.zeros_like(gate)
    err = ctypes.create_string_buffer(2048)
    assert Lg.dropin_gpu_run(V, T, gate.ctypes.data_as(c_f32p), pitch.ctypes.data_as(c_f32p), per0.ctypes.data_as(c_f32p),
                             per1.ctypes.data_as(c_f32p), err, 2048) == 0, err.value.decode()
    mix0, mix1 = np.zeros(64 * T, np.float32), np.zeros(64 * T, np.float32)
    st = Lg.dropin_gpu_run_mixed(V, T, gate.ctypes.data_as(c_f32p), pitch.ctypes.data_as(c_f32p), mix0.ctypes.data_as(c_f32p),
                                 mix1.ctypes.data_as(c_f32p), err, 2048)
    assert st == 0, err.value.decode()
    assert_bits_equal(mix0, oracle.mixdown(per0, None), True, "mixed output 0")
    assert_bits_equal(mix1, oracle.mixdown(per1, None), True, "mixed output 1")
    assert np.abs(mix0).max() > 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("launches,knobs_at", [(1, 60), (5, 60), (5, 24), (10, 6)])
def test_reverb_with_feedback_state_same_source_same_bits(launches, knobs_at):
    """tests/cpp/dropin_reverb.h: smoothed float parameters (LinearGlide), FractionalDelay, Allpass<IntegerDelay>, seven
    Allpass<PitchbendableDelay>, two PitchbendableDelays and a stereo feedback path kept in DSPVector members of the user's
    state struct — compiled unchanged against the reference and against the shim; 64 reverbs per launch on the GPU.
    knobs_at < 60: before that vector the host changes three plain floats of the state (size, feedback, damping). The reference
    just reads them on its next call; the captured program takes them in with VoiceProgram::update() — live constants and
    coefficients, no recompilation — and the glides, the pitch-bending delays and the tail follow bit for bit."""
    from inputs import lcg_noise
    Lg, Lr = _gpu_lib(), _ref_lib()
    V, T = 64, 60
    inL = lcg_noise(np.arange(V, dtype=np.uint32) + 1, 64 * T) * np.float32(0.3)
    inR = lcg_noise(np.arange(V, dtype=np.uint32) + 1001, 64 * T) * np.float32(0.3)
    inL[:, 64 * 20:] = 0   # let the tail ring
    inR[:, 64 * 20:] = 0
    wantL, wantR = np.zeros_like(inL), np.zeros_like(inL)
    assert Lr.plate_ref_run(V, T, knobs_at, inL.ctypes.data_as(c_f32p), inR.ctypes.data_as(c_f32p), wantL.ctypes.data_as(c_f32p),
                            wantR.ctypes.data_as(c_f32p)) == 0
    gotL, gotR = np.zeros_like(inL), np.zeros_like(inL)
    err = ctypes.create_string_buffer(4096)
    st = Lg.plate_gpu_run(V, T, launches, knobs_at, inL.ctypes.data_as(c_f32p), inR.ctypes.data_as(c_f32p), gotL.ctypes.data_as(c_f32p),
                          gotR.ctypes.data_as(c_f32p), err, 4096)
    assert st == 0, err.value.decode()
    assert_bits_equal(gotL, wantL, True, "plate reverb left")
    assert_bits_equal(gotR, wantR, True, "plate reverb right")
    assert np.abs(wantL[:, 64 * 40:]).max() > 1e-4   # the tail is still sounding 20 vectors after the input stopped


@pytest.mark.gpu
@pytest.mark.parametrize("launches", [1, 3, 9])
def test_oversampled_functions_same_source_same_bits(launches):
    """tests/cpp/dropin_oversample.h: Upsample2xFunction<1> around a stateful waveshaper and Downsample2xFunction<2> around a
    half-rate modulator — compiled unchanged against the reference and against the shim (rate regions of the fused graph);
    process calls of 9, 3 and 1 vectors (odd counts: the half-rate function pairs vectors across launches)."""
    from inputs import lcg_noise
    Lg, Lr = _gpu_lib(), _ref_lib()
    V, T = 70, 9
    in0 = lcg_noise(np.arange(V, dtype=np.uint32) + 5, 64 * T) * np.float32(0.7)
    in1 = (0.5 * np.sin(np.arange(64 * T)[None, :] * 0.004 * (1 + np.arange(V)[:, None] % 5))).astype(np.float32)
    want0, want1 = np.zeros_like(in0), np.zeros_like(in0)
    assert Lr.oversample_ref_run(V, T, in0.ctypes.data_as(c_f32p), in1.ctypes.data_as(c_f32p), want0.ctypes.data_as(c_f32p),
                                 want1.ctypes.data_as(c_f32p)) == 0
    got0, got1 = np.zeros_like(in0), np.zeros_like(in0)
    err = ctypes.create_string_buffer(4096)
    st = Lg.oversample_gpu_run(V, T, launches, in0.ctypes.data_as(c_f32p), in1.ctypes.data_as(c_f32p), got0.ctypes.data_as(c_f32p),
                               got1.ctypes.data_as(c_f32p), err, 4096)
    assert st == 0, err.value.decode()
    assert_bits_equal(got0, want0, True, "oversampled shaper")
    assert_bits_equal(got1, want1, True, "half-rate branch + mix")
    assert np.abs(want0).max() > 0.1 and np.abs(want1[:, 64:]).max() > 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("flush", [0, 1])
def test_flush_denormals_scope_same_source_same_bits(flush):
    """tests/cpp/dropin_decay.h: a process function that opens with `UsingFlushDenormalsToZero f;` (MLDSPUtils.h:51-96, as
    examples/audio-and-midi/fdtd.cpp:161) around recurrences that ring out for 1500 DSPVectors after a short burst —
    compiled unchanged against the reference (MXCSR FZ | DAZ inside the scope) and against the shim (the captured program
    runs in the engine's flush mode); and the same body without the scope, whose tails cross the denormal range."""
    from inputs import lcg_noise
    Lg, Lr = _gpu_lib(), _ref_lib()
    V, T = 48, 1500
    in0 = np.zeros((V, 64 * T), np.float32)
    in0[:, :128] = lcg_noise(np.arange(V, dtype=np.uint32) + 3, 128) * np.linspace(0.05, 1.0, V, dtype=np.float32)[:, None]
    want0, want1 = np.zeros_like(in0), np.zeros_like(in0)
    assert Lr.decay_ref_run(V, T, flush, in0.ctypes.data_as(c_f32p), want0.ctypes.data_as(c_f32p), want1.ctypes.data_as(c_f32p)) == 0
    got0, got1 = np.zeros_like(in0), np.zeros_like(in0)
    err = ctypes.create_string_buffer(4096)
    used = ctypes.c_int(-1)
    st = Lg.decay_gpu_run(V, T, flush, in0.ctypes.data_as(c_f32p), got0.ctypes.data_as(c_f32p), got1.ctypes.data_as(c_f32p), ctypes.byref(used), err, 4096)
    assert st == 0, err.value.decode()
    assert used.value == flush                      # the capture saw (or did not see) the scope object
    assert_bits_equal(got0, want0, True, f"low copy of the ringing filter, flush={flush}")
    assert_bits_equal(got1, want1, True, f"feedback path, flush={flush}")
    tiny = lambda a: int(((np.abs(a) > 0) & (np.abs(a) < np.float32(1.17549435e-38))).sum())   # noqa: E731
    if flush:
        assert tiny(want0) == 0 and tiny(want1) == 0
    else:
        assert tiny(want0) > 1000                   # the unflushed tail really lives in the denormal range


class _Ev(ctypes.Structure):
    _fields_ = [("type", ctypes.c_uint8), ("channel", ctypes.c_uint8), ("sourceIdx", ctypes.c_uint16), ("time", ctypes.c_int32),
                ("value1", ctypes.c_float), ("value2", ctypes.c_float)]


@pytest.mark.gpu
@pytest.mark.parametrize("own_stream", [False, True])
def test_synth_subclass_events_to_audio_same_source_same_bits(own_stream, monkeypatch):
    """own_stream: VoiceProgramOptions::eventsOnOwnStream - EventsToSignals on a second engine, its kernel for block k + 1 beside the
    voice kernel of block k, two sets of row signals, fences: the same bits.
    tests/cpp/dropin_synth.h: a Synth subclass (processVoice reading the EventsToSignals voice rows). Reference side: its own
    Synth::processVector + AudioContext + EventsToSignals, one instrument at a time. GPU side: mlgpu_events -> the captured
    processVoice for all voices of 40 instruments -> mlgpu_mixdown_groups. MIDI events in, stereo audio out, bit for bit.
    Half way through the host changes the envelope times of every voice (SmallSynth::setEnvelope, as a parameter callback would);
    the GPU side calls SynthProgram::update() and the sounding notes continue with the new coefficients, like the reference's."""
    from test_gpu_events import performance
    if own_stream:
        monkeypatch.setenv("MLGPU_TEST_EVENTS_OWN_STREAM", "1")
    Lg, Lr = _gpu_lib(), _ref_lib()
    Lr.synth_ref_run.restype = ctypes.c_int
    c_szp = ctypes.POINTER(ctypes.c_size_t)
    Lr.synth_ref_run.argtypes = [ctypes.POINTER(_Ev), ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int, c_f32p, c_f32p,
                                 c_f32p, c_szp, ctypes.c_int]
    Lg.synth_gpu_run.restype = ctypes.c_int
    Lg.synth_gpu_run.argtypes = [ctypes.c_size_t, ctypes.POINTER(_Ev), ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                 ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p, c_f32p, ctypes.c_size_t, c_f32p, c_szp, ctypes.c_int,
                                 ctypes.c_char_p, ctypes.c_size_t]
    N, block, n_blocks = 40, 512, 10
    scope_inst, scope_read = 17, 700     # the UI reads up to 700 frames after each block: sometimes all there is, sometimes not
    scope_cap = n_blocks * scope_read * 2
    S = block * n_blocks
    glide, drift = 0.012, 0.6
    per_inst = [performance("midi", 900 + k, S, 6) for k in range(N)]
    wantL, wantR = np.zeros((N, S), np.float32), np.zeros((N, S), np.float32)
    for k, evs in enumerate(per_inst):
        arr = (_Ev * max(1, len(evs)))(*[_Ev(*e) for e in evs])
        if k == scope_inst:
            want_scope, want_counts = np.zeros(scope_cap, np.float32), np.zeros(n_blocks, np.uint64)
            sc = (want_scope.ctypes.data_as(c_f32p), want_counts.ctypes.data_as(c_szp), scope_read)
        else:
            sc = (None, None, 0)
        assert Lr.synth_ref_run(arr, len(evs), glide, drift, block, n_blocks, wantL[k].ctypes.data_as(c_f32p), wantR[k].ctypes.data_as(c_f32p), *sc) == 0
    flat = [(e, k) for k, evs in enumerate(per_inst) for e in evs]
    arr = (_Ev * len(flat))(*[_Ev(*e) for e, _ in flat])
    inst = (ctypes.c_int * len(flat))(*[k for _, k in flat])
    gotL, gotR = np.zeros((N, S), np.float32), np.zeros((N, S), np.float32)
    err = ctypes.create_string_buffer(4096)
    got_scope, got_counts = np.zeros(scope_cap, np.float32), np.zeros(n_blocks, np.uint64)
    st = Lg.synth_gpu_run(N, arr, inst, len(flat), glide, drift, block, n_blocks, 3, gotL.ctypes.data_as(c_f32p), gotR.ctypes.data_as(c_f32p),
                          scope_inst, got_scope.ctypes.data_as(c_f32p), got_counts.ctypes.data_as(c_szp), scope_read, err, 4096)
    assert st == 0, err.value.decode()
    assert_bits_equal(gotL, wantL, True, "synth left")
    assert_bits_equal(gotR, wantR, True, "synth right")
    assert np.abs(wantL).max() > 0.05
    # storePublishedSignal("scope", ...) inside processVoice: the published ring of instrument 17, read as a UI would
    assert np.array_equal(got_counts, want_counts), (got_counts, want_counts)
    assert want_counts.sum() > 0 and np.abs(want_scope).max() > 0.01
    assert_bits_equal(got_scope, want_scope, True, "published scope")


@pytest.mark.gpu
def test_lean_synth_rows_in_kernel_same_bits():
    """tests/cpp/dropin_synth.h: LeanSynth reads nothing but pitch and gate. With gpu::VoiceProgramOptions::eventRowsInKernel the
    captured voice kernel computes the two rows itself from the events' records (mlgpu_graph_add_event_row) - no events kernel,
    no row buffers. Both GPU forms against the reference's own Synth / AudioContext / EventsToSignals, bit for bit, with the
    envelope change half way through."""
    from test_gpu_events import performance
    Lg, Lr = _gpu_lib(), _ref_lib()
    Lr.lean_synth_ref_run.restype = ctypes.c_int
    Lr.lean_synth_ref_run.argtypes = [ctypes.POINTER(_Ev), ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int, c_f32p, c_f32p]
    Lg.lean_synth_gpu_run.restype = ctypes.c_int
    Lg.lean_synth_gpu_run.argtypes = [ctypes.c_size_t, ctypes.POINTER(_Ev), ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p, c_f32p, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                      ctypes.c_char_p, ctypes.c_size_t]
    N, block, n_blocks = 24, 512, 10
    S = block * n_blocks
    glide, drift = 0.012, 0.6
    per_inst = [performance("midi", 700 + k, S, 6) for k in range(N)]
    wantL, wantR = np.zeros((N, S), np.float32), np.zeros((N, S), np.float32)
    for k, evs in enumerate(per_inst):
        arr = (_Ev * max(1, len(evs)))(*[_Ev(*e) for e in evs])
        assert Lr.lean_synth_ref_run(arr, len(evs), glide, drift, block, n_blocks, wantL[k].ctypes.data_as(c_f32p), wantR[k].ctypes.data_as(c_f32p)) == 0
    flat = [(e, k) for k, evs in enumerate(per_inst) for e in evs]
    arr = (_Ev * len(flat))(*[_Ev(*e) for e, _ in flat])
    inst = (ctypes.c_int * len(flat))(*[k for _, k in flat])
    for in_kernel in (0, 1):
        gotL, gotR = np.zeros((N, S), np.float32), np.zeros((N, S), np.float32)
        err = ctypes.create_string_buffer(4096)
        did = ctypes.c_int(-1)
        st = Lg.lean_synth_gpu_run(N, arr, inst, len(flat), glide, drift, block, n_blocks, 3, gotL.ctypes.data_as(c_f32p), gotR.ctypes.data_as(c_f32p),
                                   in_kernel, ctypes.byref(did), err, 4096)
        assert st == 0, err.value.decode()
        assert did.value == in_kernel
        assert_bits_equal(gotL, wantL, True, f"lean synth left (rows in kernel: {in_kernel})")
        assert_bits_equal(gotR, wantR, True, f"lean synth right (rows in kernel: {in_kernel})")
    assert np.abs(wantL).max() > 0.05


@pytest.mark.gpu
def test_every_free_function_by_name_same_source_same_bits():
    """tests/cpp/dropin_ops.h: every free function of MLDSPOps.h that takes whole DSPVectors, called by name with the reference's
    argument order, compiled unchanged against the reference and against the shim, on general floats (infinities, NaNs, denormals,
    huge and tiny values). Pins which device operation each name stands for and which argument goes where."""
    from inputs import general_floats, assert_rel_close
    Lg, Lr = _gpu_lib(), _ref_lib()
    K = 8
    V, T = 96, 3
    S = 64 * T
    a = general_floats(V * S, seed=71).reshape(V, S).copy()
    b = general_floats(V * S, seed=72)[::-1].reshape(V, S).copy()
    # half of the voices get ordinary audio-range values, so that the transcendental paths see their usual domain as well
    rng = np.random.default_rng(7)
    a[::2] = rng.uniform(-2.0, 2.0, (V // 2, S)).astype(np.float32)
    b[::2] = rng.uniform(-2.0, 2.0, (V // 2, S)).astype(np.float32)
    want = np.zeros((K, V, S), np.float32)
    got = np.zeros((K, V, S), np.float32)
    Lr.ops_ref_run.restype = ctypes.c_int
    Lr.ops_ref_run.argtypes = [ctypes.c_size_t, ctypes.c_size_t, c_f32p, c_f32p, c_f32p]
    Lg.ops_gpu_run.restype = ctypes.c_int
    Lg.ops_gpu_run.argtypes = [ctypes.c_size_t, ctypes.c_size_t, c_f32p, c_f32p, c_f32p, ctypes.c_char_p, ctypes.c_size_t]
    assert Lr.ops_ref_run(V, T, a.ctypes.data_as(c_f32p), b.ctypes.data_as(c_f32p), want.ctypes.data_as(c_f32p)) == 0
    err = ctypes.create_string_buffer(4096)
    assert Lg.ops_gpu_run(V, T, a.ctypes.data_as(c_f32p), b.ctypes.data_as(c_f32p), got.ctypes.data_as(c_f32p), err, 4096) == 0, err.value.decode()
    names = ["arithmetic, min / max / clamp", "sin cos exp log", "exp2 log2 pow", "approximations", "sqrt abs sign signBit fractionalPart within lerp inverseLerp",
             "comparisons + selects", "conversions, int arithmetic, index vectors, rows", "hardware approximations"]
    for k in range(K - 1):
        assert_bits_equal(got[k], want[k], True, f"ops drop-in output {k} ({names[k]})")
    # the last output is a sum of two hardware-approximate terms: each is within 1.5 * 2^-11 of the reference's (rsqrtps / rcpps
    # tables against v_rsq_f32 / v_rcp_f32), so the sum is within that of the terms' magnitudes
    with np.errstate(all="ignore"):
        a64, b64 = np.abs(a.astype(np.float64)), np.abs(b.astype(np.float64))
        scale = np.sqrt(a64 + 0.01) + 0.5 * b64 / (a64 + 1.0)
        fin = np.isfinite(want[K - 1]) & np.isfinite(got[K - 1]) & np.isfinite(scale)
        err = np.abs(got[K - 1].astype(np.float64) - want[K - 1].astype(np.float64))
    assert (err[fin] <= 2.0 * 1.5 * 2.0 ** -11 * scale[fin] + 1e-30).all(), names[K - 1]
    assert (np.isfinite(want[K - 1]) == np.isfinite(got[K - 1])).mean() > 0.99


@pytest.mark.gpu
@pytest.mark.parametrize("launches", [1, 3])
def test_every_stateful_object_by_name_same_source_same_bits(launches):
    """tests/cpp/dropin_objects.h: the generators, filters and delay lines no other drop-in source uses - TickGen, ImpulseGen,
    PhasorGen, OneShotGen (trigger()), TestSineGen, TempoLock, Bandpass, HiShelf, Integrator (mLeak), Differentiator, Peak
    (peakHoldSamples), RMS, LinearGlide, Interpolator1, IntegerDelay (fixed and modulated), FractionalDelay, PitchbendableDelay -
    constructed, configured and called the way user code does, compiled unchanged against the reference and against the shim.
    Pins the shim's class surface: names, setters, coefficient makers, operator() forms and the state each starts from."""
    from inputs import assert_rel_close
    Lg, Lr = _gpu_lib(), _ref_lib()
    K = 8
    V, T = 80, 12
    S = 64 * T
    rng = np.random.default_rng(23)
    x = rng.uniform(-1.0, 1.0, (V, S)).astype(np.float32)
    x[:, 64 * 5:64 * 7] = 0.0                                    # a silence: Peak's hold and RMS's decay
    phase = rng.uniform(0.0, 1.0, (V, 1))
    slow = np.mod(phase + np.arange(S)[None, :] / 700.0, 1.0).astype(np.float32)      # a transport phasor, wrapping once
    want = np.zeros((K, V, S), np.float32)
    got = np.zeros((K, V, S), np.float32)
    Lr.objects_ref_run.restype = ctypes.c_int
    Lr.objects_ref_run.argtypes = [ctypes.c_size_t, ctypes.c_size_t, c_f32p, c_f32p, c_f32p]
    Lg.objects_gpu_run.restype = ctypes.c_int
    Lg.objects_gpu_run.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, c_f32p, c_f32p, c_f32p, ctypes.c_char_p, ctypes.c_size_t]
    assert Lr.objects_ref_run(V, T, x.ctypes.data_as(c_f32p), slow.ctypes.data_as(c_f32p), want.ctypes.data_as(c_f32p)) == 0
    err = ctypes.create_string_buffer(4096)
    assert Lg.objects_gpu_run(V, T, launches, x.ctypes.data_as(c_f32p), slow.ctypes.data_as(c_f32p), got.ctypes.data_as(c_f32p), err, 4096) == 0, err.value.decode()
    names = ["TickGen + ImpulseGen + OneShotGen", "PhasorGen + TestSineGen + TempoLock", "Bandpass + HiShelf", "Integrator + Differentiator", "Peak", "RMS",
             "LinearGlide + Interpolator1", "IntegerDelay (fixed, modulated) + FractionalDelay + PitchbendableDelay"]
    for k in range(K):
        assert np.abs(want[k]).max() > 1e-3, names[k]
        if k in (4, 5):        # sqrtApprox inside: 2^-11 relative (test_gpu_parity.HW_REL)
            assert_rel_close(got[k], want[k], 2.0 ** -11, f"objects drop-in output {k} ({names[k]})")
        else:
            assert_bits_equal(got[k], want[k], True, f"objects drop-in output {k} ({names[k]})")


@pytest.mark.gpu
@pytest.mark.parametrize("which,own_stream", [("controller", None), ("tempo", None), ("controller", "0"), ("tempo", "1")])
def test_synth_that_reads_context_signals_same_bits(which, own_stream, monkeypatch):
    """tests/cpp/dropin_synth.h. TempoSynth::processVoice locks a tremolo LFO to ctx->getBeatPhase() with a TempoLock per voice,
    while the host (HostTransport, both sides) starts, changes tempo, stops and restarts elsewhere: the beat phase is one device
    signal per instrument (mlgpu_transport) behind SynthProgram::updateTime.
    ControllerSynth::processVoice calls ctx->getInputController(74 / 1 / 128) - the instrument's smoothed
    MIDI controllers (brightness into a per-sample filter cutoff, mod wheel into the pitch, channel pressure into the level). The
    shim turns each into one device signal per instrument (mlgpu_events_watch_controllers) read by that instrument's voices
    (mlgpu_graph_set_input_group). Against the reference's own Synth / AudioContext / EventsToSignals, bit for bit, with the
    voice rows read from memory and computed in the kernel.
    own_stream: the run asks for VoiceProgramOptions::eventsOnOwnStream. "1": it must take effect (TempoSynth reads the beat phase, which
    the transport makes on the voice stream) - with the rows read from memory; "0": it must be declined (ControllerSynth's controller
    signals are made by the events call into ONE buffer the voice kernel of the block before may still be reading)."""
    from test_gpu_events import performance
    Lg, Lr = _gpu_lib(), _ref_lib()
    ref_run, gpu_run = getattr(Lr, which + "_synth_ref_run"), getattr(Lg, which + "_synth_gpu_run")
    ref_run.restype = ctypes.c_int
    ref_run.argtypes = [ctypes.POINTER(_Ev), ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int, c_f32p, c_f32p]
    gpu_run.restype = ctypes.c_int
    gpu_run.argtypes = [ctypes.c_size_t, ctypes.POINTER(_Ev), ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p, c_f32p, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                            ctypes.c_char_p, ctypes.c_size_t]
    N, block, n_blocks = 20, 512, 10
    S = block * n_blocks
    glide, drift = 0.01, 0.4
    per_inst = [performance("midi", 900 + k, S, 6) for k in range(N - 1)] + [[]]
    wantL, wantR = np.zeros((N, S), np.float32), np.zeros((N, S), np.float32)
    for k, evs in enumerate(per_inst):
        arr = (_Ev * max(1, len(evs)))(*[_Ev(*e) for e in evs])
        assert ref_run(arr, len(evs), glide, drift, block, n_blocks, wantL[k].ctypes.data_as(c_f32p), wantR[k].ctypes.data_as(c_f32p)) == 0
    flat = [(e, k) for k, evs in enumerate(per_inst) for e in evs]
    arr = (_Ev * len(flat))(*[_Ev(*e) for e, _ in flat])
    inst = (ctypes.c_int * len(flat))(*[k for _, k in flat])
    for in_kernel in (0, 1):
        if own_stream is not None:
            if in_kernel:
                continue     # rows computed inside the voice kernel: there is no events kernel to move
            monkeypatch.setenv("MLGPU_TEST_EVENTS_OWN_STREAM", own_stream)
        gotL, gotR = np.zeros((N, S), np.float32), np.zeros((N, S), np.float32)
        err = ctypes.create_string_buffer(4096)
        did = ctypes.c_int(-1)
        st = gpu_run(N, arr, inst, len(flat), glide, drift, block, n_blocks, 3, gotL.ctypes.data_as(c_f32p), gotR.ctypes.data_as(c_f32p),
                                         in_kernel, ctypes.byref(did), err, 4096)
        assert st == 0, err.value.decode()
        assert did.value == in_kernel
        assert_bits_equal(gotL, wantL, True, f"{which} synth left (rows in kernel: {in_kernel})")
        assert_bits_equal(gotR, wantR, True, f"{which} synth right (rows in kernel: {in_kernel})")
    assert np.abs(wantL).max() > 0.05 and np.abs(wantR).max() > 0.01


@pytest.mark.gpu
@pytest.mark.parametrize("in_kernel", [0, 1])
def test_plugin_flow_arbitrary_host_blocks_same_bits(in_kernel):
    """The whole boundary at once, as a plug-in wrapper drives it: host blocks of arbitrary sizes (1 ... 512 frames) through
    SignalProcessBuffer, note and controller events with block-relative times, the host's time report before every block,
    clearInputEvents() after it - tests/cpp/dropin_synth.h: PluginSynth reads voice rows, a controller and the beat phase.
    Reference: one SignalProcessBuffer + AudioContext + PluginSynth per instrument (oracle/dropin_ref.cpp: plugin_ref_run), summed
    in instrument order. GPU: every instrument behind one mlgpu_process_buffer and one SynthProgram. Bit for bit - including what
    the reference does with events of a block in which no DSPVector falls due (they are cleared unprocessed)."""
    from test_gpu_events import performance
    Lg, Lr = _gpu_lib(), _ref_lib()
    c_ip = ctypes.POINTER(ctypes.c_int)
    Lr.plugin_ref_run.restype = ctypes.c_int
    Lr.plugin_ref_run.argtypes = [ctypes.POINTER(_Ev), ctypes.c_int, ctypes.c_float, ctypes.c_float, c_ip, ctypes.c_int, ctypes.c_int, c_f32p, c_f32p]
    Lg.plugin_gpu_run.restype = ctypes.c_int
    Lg.plugin_gpu_run.argtypes = [ctypes.c_size_t, ctypes.POINTER(_Ev), c_ip, ctypes.c_int, ctypes.c_float, ctypes.c_float, c_ip, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_int, c_f32p, c_f32p, ctypes.c_char_p, ctypes.c_size_t]
    rng = np.random.default_rng(5)
    sizes = [64, 100, 37, 512, 1, 200, 64, 333, 17, 480, 31, 33, 128, 7, 250] + [int(x) for x in rng.integers(1, 513, 25)]
    S, N, glide, drift = int(np.sum(sizes)), 6, 0.01, 0.3
    blocks = (ctypes.c_int * len(sizes))(*sizes)
    per_inst = [performance("midi", 1200 + k, S, 6) for k in range(N)]
    want = np.zeros((2, S), np.float32)
    for k, evs in enumerate(per_inst):
        l, r = np.zeros(S, np.float32), np.zeros(S, np.float32)
        arr = (_Ev * max(1, len(evs)))(*[_Ev(*e) for e in evs])
        assert Lr.plugin_ref_run(arr, len(evs), glide, drift, blocks, len(sizes), 512, l.ctypes.data_as(c_f32p), r.ctypes.data_as(c_f32p)) == 0
        want[0], want[1] = want[0] + l, want[1] + r          # ((0 + i0) + i1) + ...: mlgpu_mixdown_groups' order
    flat = [(e, k) for k, evs in enumerate(per_inst) for e in evs]
    arr = (_Ev * len(flat))(*[_Ev(*e) for e, _ in flat])
    inst = (ctypes.c_int * len(flat))(*[k for _, k in flat])
    gotL, gotR = np.zeros(S, np.float32), np.zeros(S, np.float32)
    err = ctypes.create_string_buffer(4096)
    st = Lg.plugin_gpu_run(N, arr, inst, len(flat), glide, drift, blocks, len(sizes), 512, in_kernel, gotL.ctypes.data_as(c_f32p), gotR.ctypes.data_as(c_f32p), err, 4096)
    assert st == 0, err.value.decode()
    assert_bits_equal(gotL, want[0], True, "plug-in bank left")
    assert_bits_equal(gotR, want[1], True, "plug-in bank right")
    assert np.abs(want[0]).max() > 0.05 and np.abs(want[1]).max() > 0.001


@pytest.mark.gpu
@pytest.mark.parametrize("launches", [1, 4])
def test_routing_and_function_wrappers_by_name_same_source_same_bits(launches):
    """tests/cpp/dropin_routing.h: mix, multiplex, multiplexLinear, demultiplex, demultiplexLinear (selectors inside and beyond
    [0, 1), exactly on the boundaries), Bank<SineGen, 3> / Bank<Lopass, 2> with row arguments, operator[] and clear(), map() over rows
    with and without the row index, the routing functions on two-row arrays - compiled unchanged against the reference and against
    the shim. (FeedbackDelayFunction cannot run in the reference: see the header.)"""
    Lg, Lr = _gpu_lib(), _ref_lib()
    K = 8
    V, T = 72, 16
    S = 64 * T
    rng = np.random.default_rng(41)
    a = rng.uniform(-1.0, 1.0, (V, S)).astype(np.float32)
    b = rng.uniform(-1.0, 1.0, (V, S)).astype(np.float32)
    sel = np.mod(rng.uniform(0, 1, (V, 1)) + np.arange(S)[None, :] / 300.0, 1.0).astype(np.float32)
    sel[::7] = (sel[::7] * 3.0).astype(np.float32)       # beyond [0, 1): the fractional part counts (a negative selector indexes
    #                                                      out of bounds in the reference: undefined there, index 0 here, mlgpu.h)
    sel[:, ::50] = np.float32(1.0 / 3.0)                                  # exactly on a boundary
    sel[:, 25::50] = np.float32(0.0)
    want = np.zeros((K, V, S), np.float32)
    got = np.zeros((K, V, S), np.float32)
    Lr.routing_ref_run.restype = ctypes.c_int
    Lr.routing_ref_run.argtypes = [ctypes.c_size_t, ctypes.c_size_t, c_f32p, c_f32p, c_f32p, c_f32p]
    Lg.routing_gpu_run.restype = ctypes.c_int
    Lg.routing_gpu_run.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_char_p, ctypes.c_size_t]
    p = lambda x: x.ctypes.data_as(c_f32p)  # noqa: E731
    assert Lr.routing_ref_run(V, T, p(a), p(b), p(sel), p(want)) == 0
    err = ctypes.create_string_buffer(4096)
    assert Lg.routing_gpu_run(V, T, launches, p(a), p(b), p(sel), p(got), err, 4096) == 0, err.value.decode()
    names = ["mix", "multiplex", "multiplexLinear", "demultiplex", "demultiplexLinear", "Bank<SineGen,3> + Bank<Lopass,2>", "map (two forms)",
             "multiplex / mix on two-row arrays"]
    for k in range(K):
        assert np.abs(want[k]).max() > 1e-3, names[k]
        assert_bits_equal(got[k], want[k], True, f"routing drop-in output {k} ({names[k]})")


@pytest.mark.gpu
def test_controllers_to_audio_vector_form_same_source_same_bits():
    """tests/cpp/dropin_controllers.h: the process function of the reference's controllers-to-audio.cpp example (eight sines tuned by
    eight MIDI controllers, a ninth for the volume) with its per-vector float mapping written on whole vectors. A plain
    SignalProcessFn that reads ctx->getInputController(n): through gpu::VoiceProgram the nine controllers become context inputs
    (VoiceProgram::contextInputs) fed from mlgpu_events_controller_signal - here one context per voice, 40 instances with their own
    controller movements, against one reference AudioContext each."""
    Lg, Lr = _gpu_lib(), _ref_lib()
    Lr.ctl_audio_ref_run.restype = ctypes.c_int
    Lr.ctl_audio_ref_run.argtypes = [ctypes.POINTER(_Ev), ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p]
    Lg.ctl_audio_gpu_run.restype = ctypes.c_int
    Lg.ctl_audio_gpu_run.argtypes = [ctypes.c_size_t, ctypes.POINTER(_Ev), ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     c_f32p, ctypes.c_char_p, ctypes.c_size_t]
    N, block, n_blocks = 40, 512, 12
    S = block * n_blocks
    CTRL = 6
    numbers = [19, 23, 27, 31, 49, 53, 57, 61, 62]
    per_inst = []
    for k in range(N):
        rng = np.random.default_rng(3000 + k)
        evs, t = [(CTRL, 1, 62, int(rng.integers(0, 100)), float(np.float32(rng.uniform(0.3, 1.0))), 0.0)], int(rng.integers(0, 300))
        while t < S and k != N - 1:             # the last instance only ever gets its volume set
            evs.append((CTRL, int(rng.integers(1, 17)), int(rng.choice(numbers)), t, float(np.float32(rng.random())), 0.0))
            t += int(rng.integers(1, 700))
        per_inst.append(sorted(evs, key=lambda e: e[3]))
    want = np.zeros((N, S), np.float32)
    for k, evs in enumerate(per_inst):
        arr = (_Ev * len(evs))(*[_Ev(*e) for e in evs])
        assert Lr.ctl_audio_ref_run(arr, len(evs), block, n_blocks, want[k].ctypes.data_as(c_f32p)) == 0
    flat = [(e, k) for k, evs in enumerate(per_inst) for e in evs]
    arr = (_Ev * len(flat))(*[_Ev(*e) for e, _ in flat])
    inst = (ctypes.c_int * len(flat))(*[k for _, k in flat])
    got = np.zeros((N, S), np.float32)
    err = ctypes.create_string_buffer(4096)
    st = Lg.ctl_audio_gpu_run(N, arr, inst, len(flat), block, n_blocks, 3, got.ctypes.data_as(c_f32p), err, 4096)
    assert st == 0, err.value.decode()
    assert_bits_equal(got, want, True, "controllers-to-audio")
    assert np.abs(want).max() > 0.2 and np.abs(want[-1]).max() > 0.01


@pytest.mark.gpu
@pytest.mark.parametrize("every_other", [0, 1])
def test_one_shot_retriggered_between_launches(every_other):
    """OneShotGen::trigger() called again while the program runs - gpu::VoiceProgram::trigger(shot) for every voice,
    trigger(shot, flags) for some - against the reference's objects retriggered before the same DSPVector."""
    Lg, Lr = _gpu_lib(), _ref_lib()
    V, T, launches = 40, 12, 3
    S = 64 * T
    rng = np.random.default_rng(3)
    x = rng.uniform(-1.0, 1.0, (V, S)).astype(np.float32)
    slow = np.mod(rng.uniform(0, 1, (V, 1)) + np.arange(S)[None, :] / 700.0, 1.0).astype(np.float32)
    want, got = np.zeros((8, V, S), np.float32), np.zeros((8, V, S), np.float32)
    Lr.objects_ref_run_retrigger.restype = ctypes.c_int
    Lr.objects_ref_run_retrigger.argtypes = [ctypes.c_size_t, ctypes.c_size_t, c_f32p, c_f32p, c_f32p, ctypes.c_int, ctypes.c_int]
    Lg.objects_gpu_run_retrigger.restype = ctypes.c_int
    Lg.objects_gpu_run_retrigger.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, c_f32p, c_f32p, c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p,
                                             ctypes.c_size_t]
    p = lambda a: a.ctypes.data_as(c_f32p)  # noqa: E731
    assert Lr.objects_ref_run_retrigger(V, T, p(x), p(slow), p(want), 2 * (T // launches), every_other) == 0
    err = ctypes.create_string_buffer(4096)
    assert Lg.objects_gpu_run_retrigger(V, T, launches, p(x), p(slow), p(got), 2, every_other, err, 4096) == 0, err.value.decode()
    assert_bits_equal(got[0], want[0], True, "TickGen + ImpulseGen + OneShotGen, retriggered")
    plain = np.zeros((8, V, S), np.float32)
    Lr.objects_ref_run.argtypes = [ctypes.c_size_t, ctypes.c_size_t, c_f32p, c_f32p, c_f32p]
    assert Lr.objects_ref_run(V, T, p(x), p(slow), p(plain)) == 0
    changed = np.abs(plain[0] - want[0]).max(axis=1) > 0
    assert changed[0::2].all() and (changed[1::2].all() if not every_other else not changed[1::2].any())


@pytest.mark.gpu
def test_host_data_forms_same_source_same_bits():
    """tests/cpp/dropin_hostdata.h: where DSPVector code touches single floats on the host - map(float()) with a stateful function,
    map(float(int)) over columnIndexInt(), window tables written through getBuffer() in a setup function, the overlap-add use of
    DSPBuffer (Tests/dspBufferTest.cpp "overlap") read back into a DSPVector, v[n] on host-made vectors. Same source against the
    reference and against the shim: in the shim these are host-evaluated tables carried into the kernel as constant vectors."""
    Lg, Lr = _gpu_lib(), _ref_lib()
    K, V, T = 4, 70, 3
    S = 64 * T
    x = lcg_noise(np.arange(V, dtype=np.uint32) + 11, S)
    want = np.zeros((K, V, S), np.float32)
    got = np.zeros((K, V, S), np.float32)
    Lr.hostdata_ref_run.restype = ctypes.c_int
    Lr.hostdata_ref_run.argtypes = [ctypes.c_size_t, ctypes.c_size_t, c_f32p, c_f32p]
    Lg.hostdata_gpu_run.restype = ctypes.c_int
    Lg.hostdata_gpu_run.argtypes = [ctypes.c_size_t, ctypes.c_size_t, c_f32p, c_f32p, ctypes.c_char_p, ctypes.c_size_t]
    assert Lr.hostdata_ref_run(V, T, x.ctypes.data_as(c_f32p), want.ctypes.data_as(c_f32p)) == 0
    err = ctypes.create_string_buffer(4096)
    assert Lg.hostdata_gpu_run(V, T, x.ctypes.data_as(c_f32p), got.ctypes.data_as(c_f32p), err, 4096) == 0, err.value.decode()
    for k, name in enumerate(["window through getBuffer + Lopass", "map(float())", "map(float(int))", "v[n] of host vectors, DSPBuffer overlap-add"]):
        assert_bits_equal(got[k], want[k], True, f"host-data drop-in output {k} ({name})")
    assert np.abs(want).max() > 0.1
