"""The source shim's immediate mode: DSPVector code OUTSIDE a VoiceProgram capture runs call by call on the device.

1. tests/cpp/dropin_eager.h - imperative user code (objects made, called and read on the spot) compiled unchanged against the
   reference (oracle/_ref/libdropin_ref.so: CPU) and against include/mlgpu/compat (tests/cpp/libdropin_gpu.so: one launch per call):
   every recorded float bit for bit.
2. The reference's OWN unit tests - Tests/tests.cpp, dspOpsTest.cpp, dspGensTest.cpp, dspFiltersTest.cpp, dspBufferTest.cpp, compiled
   UNCHANGED against the shim where the reference checkout exists (include/mlgpu/Makefile: reftests; the binary travels to the GPU
   box) - must report what they report on the CPU: all 42 assertions in 10 test cases pass. And with the device hidden the same
   binary must FAIL: there is no CPU arithmetic behind the shim."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from test_gpu_dropin import _gpu_lib, _ref_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
c_f32p = ctypes.POINTER(ctypes.c_float)
pytestmark = pytest.mark.gpu


def _blocks(names):
    out = []
    for item in names.decode().split(";"):
        if item:
            name, start = item.rsplit("@", 1)
            out.append((name, int(start)))
    return out


def test_immediate_mode_same_source_same_bits():
    Lg, Lr = _gpu_lib(), _ref_lib()
    cap = 1 << 20
    Lr.immediate_ref_run.restype = ctypes.c_long
    Lr.immediate_ref_run.argtypes = [c_f32p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    Lg.immediate_gpu_run.restype = ctypes.c_long
    Lg.immediate_gpu_run.argtypes = [c_f32p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    want, got = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
    nw, ng, err = ctypes.create_string_buffer(4096), ctypes.create_string_buffer(4096), ctypes.create_string_buffer(4096)
    n_want = Lr.immediate_ref_run(want.ctypes.data_as(c_f32p), cap, nw, 4096)
    n_got = Lg.immediate_gpu_run(got.ctypes.data_as(c_f32p), cap, ng, 4096, err, 4096)
    assert n_got >= 0, err.value.decode()
    assert n_got == n_want and 15000 < n_want <= cap
    assert nw.value == ng.value
    blocks = _blocks(nw.value) + [("end", n_want)]
    assert len(blocks) >= 11
    bad = []
    for (name, a), (_, b) in zip(blocks, blocks[1:]):
        w, g = want[a:b].view(np.uint32), got[a:b].view(np.uint32)
        assert b > a and np.abs(want[a:b]).max() > 0, name
        assert np.isfinite(want[a:b]).all(), name        # the suite stays away from NaN payload questions
        if not (w == g).all():
            i = int(np.flatnonzero(w != g)[0])
            bad.append(f"{name}: {int((w != g).sum())} of {b - a} floats differ, first at +{i} (DSPVector {i // 64} of the block, sample {i % 64}): "
                       f"reference {want[a + i]!r} ({w[i]:#010x}), device {got[a + i]!r} ({g[i]:#010x})")
    assert not bad, "\n".join(bad)


def _reftests():
    exe = os.path.join(ROOT, "tests", "cpp", "reftests_gpu")
    if os.path.isdir("/root/reference/Tests"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "include", "mlgpu"), "reftests"], stdout=subprocess.DEVNULL)
    if not os.path.exists(exe):
        pytest.skip("tests/cpp/reftests_gpu not built (it is made from the reference's Tests/ where that checkout exists)")
    return exe


def test_reference_unit_tests_pass_unchanged_on_the_device():
    """(The reference's two-thread DSPBuffer case shares an unsynchronised random source and a 1 ms / 2 ms sleep schedule between its
    threads - a host-only test that can lose its race on a loaded machine with the reference's own DSPBuffer as well. It is run on
    its own, with up to three attempts; the nine other cases, every DSP one among them, must pass at once.)"""
    exe = _reftests()
    r = subprocess.run([exe, "~[threads]"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "All tests passed (39 assertions in 9 test cases)" in r.stdout, r.stdout[-2000:]
    for attempt in range(3):
        t = subprocess.run([exe, "[threads]"], capture_output=True, text=True, timeout=600)
        if t.returncode == 0:
            break
    assert t.returncode == 0 and "All tests passed (3 assertions in 1 test case)" in t.stdout, t.stdout[-2000:]


def test_reference_unit_tests_need_the_device():
    """No CPU arithmetic behind the shim: the same binary with no GPU visible fails in the DSP test cases (the DSPBuffer ones are
    host-only by nature and still pass)."""
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    r = subprocess.run([_reftests()], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    assert "All tests passed" not in r.stdout
    assert "dsp_ops" in r.stdout and "dsp_gens" in r.stdout


# ---- the captured programs' user code, run imperatively -----------------------------------------------------------------------
# oracle/dropin_ref.cpp drives tests/cpp/dropin_*.h against the reference: one state object per voice, the process function called
# once per DSPVector with host data in ctx.inputs. Its DSP half compiled against the shim (tests/cpp/libdropin_imm.so) is the same
# loop in immediate mode. Same exported names, same arguments: outputs must be the reference's bits.

def _imm_lib():
    so = os.path.join(ROOT, "tests", "cpp", "libdropin_imm.so")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "libdropin_imm.so"], stdout=subprocess.DEVNULL)
    return ctypes.CDLL(so)


def _both(name, argtypes):
    fr, fi = getattr(_ref_lib(), name), getattr(_imm_lib(), name)
    for f in (fr, fi):
        f.restype, f.argtypes = ctypes.c_int, argtypes
    return fr, fi


def _p(a):
    return a.ctypes.data_as(c_f32p)


def _noise(V, S, seed):
    from inputs import lcg_noise
    return lcg_noise(np.arange(V, dtype=np.uint32) + seed, S)


def _same(got, want, what):
    from inputs import assert_bits_equal
    assert np.abs(want).max() > 1e-3, what
    assert_bits_equal(got, want, True, what)


def test_patch_process_function_run_imperatively():
    from test_gpu_dropin import _inputs
    sz = ctypes.c_size_t
    fr, fi = _both("dropin_ref_run", [sz, sz, c_f32p, c_f32p, c_f32p, c_f32p])
    V, T = 3, 12
    gate, pitch = _inputs(V, T)
    w0, w1, g0, g1 = (np.zeros_like(gate) for _ in range(4))
    assert fr(V, T, _p(gate), _p(pitch), _p(w0), _p(w1)) == 0
    assert fi(V, T, _p(gate), _p(pitch), _p(g0), _p(g1)) == 0
    _same(g0, w0, "patch, output 0")
    _same(g1, w1, "patch, output 1")


def test_reverb_process_function_run_imperatively():
    """dropin_reverb.h: LinearGlide-smoothed floats, FractionalDelay, Allpass<IntegerDelay>, seven Allpass<PitchbendableDelay>, a stereo
    feedback path kept in DSPVector members of the user's state - the loops live in the objects in immediate mode."""
    sz = ctypes.c_size_t
    fr, fi = _both("plate_ref_run", [sz, sz, sz, c_f32p, c_f32p, c_f32p, c_f32p])
    V, T = 2, 30
    inL, inR = _noise(V, 64 * T, 3), _noise(V, 64 * T, 9)
    inL[:, 64 * 4:] = 0
    inR[:, 64 * 4:] = 0
    wL, wR, gL, gR = (np.zeros_like(inL) for _ in range(4))
    assert fr(V, T, 12, _p(inL), _p(inR), _p(wL), _p(wR)) == 0
    assert fi(V, T, 12, _p(inL), _p(inR), _p(gL), _p(gR)) == 0
    _same(gL, wL, "reverb, left")
    _same(gR, wR, "reverb, right")


@pytest.mark.parametrize("flush", [0, 1])
def test_decay_process_function_run_imperatively(flush):
    sz = ctypes.c_size_t
    fr, fi = _both("decay_ref_run", [sz, sz, ctypes.c_int, c_f32p, c_f32p, c_f32p])
    V, T = 2, 400
    x = _noise(V, 64 * T, 21)
    x[:, 128:] = 0
    w0, w1, g0, g1 = (np.zeros_like(x) for _ in range(4))
    assert fr(V, T, flush, _p(x), _p(w0), _p(w1)) == 0
    assert fi(V, T, flush, _p(x), _p(g0), _p(g1)) == 0
    from inputs import assert_bits_equal
    assert_bits_equal(g0, w0, True, f"decay (flush {flush}), output 0")   # tiny on purpose: the tails cross the denormal range
    assert_bits_equal(g1, w1, True, f"decay (flush {flush}), output 1")
    tiny = int(((np.abs(w0) > 0) & (np.abs(w0) < np.float32(1.17549435e-38))).sum())
    assert (tiny == 0) if flush else (tiny > 100)


def test_ops_routing_objects_hostdata_run_imperatively():
    sz = ctypes.c_size_t
    V, T = 2, 6
    S = 64 * T
    a, b, sel = _noise(V, S, 1), _noise(V, S, 2), np.abs(_noise(V, S, 3)) * 0.999
    # outputs made of hardware-approximate terms (sqrtApprox / divideApprox; Peak, RMS): 1.5 * 2^-11 relative, as in test_gpu_dropin.py
    approx = {"ops_ref_run": {7}, "objects_ref_run": {4, 5}}
    for name, args, K in (("ops_ref_run", (a, b), None), ("routing_ref_run", (a, b, sel), None), ("objects_ref_run", (a, np.abs(b)), None),
                          ("hostdata_ref_run", (a,), 4)):
        fr, fi = _both(name, [sz, sz] + [c_f32p] * (len(args) + 1))
        if K is None:   # the number of outputs is a constant of the header: find it by letting the reference fill a large array
            probe = np.full(64 * V * S, np.float32(12345.0))
            assert fr(V, T, *[_p(x) for x in args], _p(probe)) == 0
            K = int(np.flatnonzero(probe != np.float32(12345.0)).max()) // (V * S) + 1
        want, got = np.zeros((K, V, S), np.float32), np.zeros((K, V, S), np.float32)
        assert fr(V, T, *[_p(x) for x in args], _p(want)) == 0
        assert fi(V, T, *[_p(x) for x in args], _p(got)) == 0
        w, g = want.view(np.uint32), got.view(np.uint32)
        # NaN results (the ops header divides and takes logs of noise): any NaN == any NaN, the contract of include/mlgpu.h
        same = (w == g) | (np.isnan(want) & np.isnan(got))
        bad = [k for k in range(K) if k not in approx.get(name, ()) and not same[k].all()]
        assert not bad, f"{name}: outputs {bad} differ"
        for k in approx.get(name, ()):
            fin = np.isfinite(want[k]) & np.isfinite(got[k])
            assert (np.abs(got[k][fin] - want[k][fin]) <= 4 * 1.5 * 2.0 ** -11 * (np.abs(want[k][fin]) + np.abs(a).max())).all(), (name, k)


def test_oversampled_process_function_run_imperatively():
    """dropin_oversample.h: Upsample2xFunction / Downsample2xFunction around stateful lambdas. Captured, they are rate regions of one
    kernel; imperatively, the reference's own schedule over HalfBandFilters on the device."""
    sz = ctypes.c_size_t
    fr, fi = _both("oversample_ref_run", [sz, sz, c_f32p, c_f32p, c_f32p, c_f32p])
    V, T = 2, 9
    in0, in1 = _noise(V, 64 * T, 5), _noise(V, 64 * T, 6)
    w0, w1, g0, g1 = (np.zeros_like(in0) for _ in range(4))
    assert fr(V, T, _p(in0), _p(in1), _p(w0), _p(w1)) == 0
    assert fi(V, T, _p(in0), _p(in1), _p(g0), _p(g1)) == 0
    _same(g0, w0, "oversampled, output 0")
    _same(g1, w1, "oversampled, output 1")


def test_feedback_composites_captured_and_immediate_agree():
    """FDN<4>, FeedbackDelayFunction, FeedbackDelayFunctionWithTap (the reference's own classes cannot size their delay lines, so there
    is no CPU run to compare with) and Allpass<PitchbendableDelay> with a modulated time: one process function, captured into a kernel
    (whose arithmetic the other drop-in tests pin against the reference) and run imperatively - same bits."""
    Lg = _gpu_lib()
    Lg.loops_captured_and_immediate_run.restype = ctypes.c_int
    Lg.loops_captured_and_immediate_run.argtypes = [ctypes.c_size_t, c_f32p, c_f32p, c_f32p, ctypes.c_char_p, ctypes.c_size_t]
    T = 24
    x = _noise(1, 64 * T, 41)[0].copy()
    x[64 * 3:] = 0
    cap, imm = np.zeros((4, 64 * T), np.float32), np.zeros((4, 64 * T), np.float32)
    err = ctypes.create_string_buffer(4096)
    assert Lg.loops_captured_and_immediate_run(T, _p(x), _p(cap), _p(imm), err, 4096) == 0, err.value.decode()
    for k in range(4):
        assert np.abs(cap[k][64 * 6:]).max() > 1e-4, k          # the loops ring on after the burst
        assert (cap[k].view(np.uint32) == imm[k].view(np.uint32)).all(), f"output {k}: captured and immediate differ"


@pytest.mark.parametrize("name", ["midi_poly4", "midi_steal2", "unison3", "sustain", "mpe5", "sr44k", "poly1"])
def test_events_to_signals_object_stepped_imperatively(name):
    """ml::EventsToSignals as an OBJECT - configured, fed events, processVector() per DSPVector, getVoice(v).outputs and
    getController(n).output read after every step: oracle/dropin_ref.cpp's driver for the reference's class, compiled against the shim.
    There it is a one-instrument mlgpu_events per object (and one helper per controller number asked for, given the object's whole
    history when it is first asked - here at the first vector, in the test below late). All eight rows and five controller signals, the
    scripted performances of tests/test_gpu_events.py, bit for bit."""
    from test_gpu_events import SCENARIOS, RefEvent, performance
    cfg = SCENARIOS[name]
    Li, Lr = _imm_lib(), _ref_lib()
    block, n_blocks = 512, 8
    P = cfg["polyphony"]
    kind = "mpe" if cfg.get("mpe") else ("sustain" if name == "sustain" else "midi")
    evs = performance(kind, 4242 + len(name), block * n_blocks, P)
    arr = (RefEvent * max(1, len(evs)))(*[RefEvent(*e) for e in evs])
    numbers = [16, 73, 74, 1, 128]
    nums = (ctypes.c_int * len(numbers))(*numbers)
    out = {}
    for tag, L in (("reference", Lr), ("immediate", Li)):
        f = L.e2s_ref_run_controllers
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                      ctypes.c_int, ctypes.POINTER(RefEvent), ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p, ctypes.POINTER(ctypes.c_int),
                      ctypes.c_int, c_f32p]
        rows = np.zeros((8, P, block * n_blocks), np.float32)
        ctl = np.zeros((len(numbers), block * n_blocks), np.float32)
        assert f(P, int(cfg.get("mpe", 0)), int(cfg.get("unison", 0)), cfg.get("sr", 48000.0), cfg.get("glide", 0.0), cfg.get("drift", 0.0),
                 cfg.get("bend", 7.0), cfg.get("mpe_bend", 24.0), cfg.get("mod_cc", 16), arr, len(evs), block, n_blocks, _p(rows), nums,
                 len(numbers), _p(ctl)) == 0
        out[tag] = (rows, ctl)
    from inputs import assert_bits_equal
    from test_gpu_events import ROW_NAMES
    for r in range(8):
        assert_bits_equal(out["immediate"][0][r], out["reference"][0][r], True, f"{name}: row {ROW_NAMES[r]}")
    for c, n in enumerate(numbers):
        assert_bits_equal(out["immediate"][1][c], out["reference"][1][c], True, f"{name}: controller {n}")
    assert np.abs(out["reference"][0][1]).max() > 0 and np.abs(out["reference"][1]).max() > 0


@pytest.mark.parametrize("name,from_vector,history_ops", [("midi_poly4", 17, None), ("mpe5", 40, None), ("midi_poly4", 40, 50),
                                                          ("sr44k", -25, None), ("midi_steal2", -9, 20), ("unison3", -1, 1)])
def test_events_to_signals_object_controllers_first_read_late(name, from_vector, history_ops, monkeypatch):
    """A process function that looks at getController(n) for the first time in the middle of a performance: the reference keeps all 129
    controller glides running from the start; the immediate object makes the signal of a controller number when it is first asked for
    and gives that helper everything the object has been told and has processed so far - the signal from there on is the reference's.
    history_ops: the bound on that history (MLGPU_E2S_HISTORY_OPS; when it is reached the controller helpers start by themselves and the
    history is dropped). from_vector < 0: the host settles on its polyphony in two steps with an event in between (setPolyphony = clear()
    at set-up time), and reads the controllers from vector -from_vector."""
    if history_ops is not None:
        monkeypatch.setenv("MLGPU_E2S_HISTORY_OPS", str(history_ops))
    from test_gpu_events import SCENARIOS, RefEvent, performance
    from inputs import assert_bits_equal
    cfg = SCENARIOS[name]
    block, n_blocks = 512, 8
    P = cfg["polyphony"]
    evs = performance("mpe" if cfg.get("mpe") else "midi", 977 + abs(from_vector), block * n_blocks, P)
    arr = (RefEvent * max(1, len(evs)))(*[RefEvent(*e) for e in evs])
    numbers = [74, 1, 16, 128, 11]
    nums = (ctypes.c_int * len(numbers))(*numbers)
    out = {}
    for tag, L in (("reference", _ref_lib()), ("immediate", _imm_lib())):
        f = L.e2s_ref_run_controllers_from
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                      ctypes.c_int, ctypes.POINTER(RefEvent), ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p, ctypes.POINTER(ctypes.c_int),
                      ctypes.c_int, c_f32p, ctypes.c_int]
        rows = np.zeros((8, P, block * n_blocks), np.float32)
        ctl = np.zeros((len(numbers), block * n_blocks), np.float32)
        assert f(P, int(cfg.get("mpe", 0)), int(cfg.get("unison", 0)), cfg.get("sr", 48000.0), cfg.get("glide", 0.0), cfg.get("drift", 0.0),
                 cfg.get("bend", 7.0), cfg.get("mpe_bend", 24.0), cfg.get("mod_cc", 16), arr, len(evs), block, n_blocks, _p(rows), nums,
                 len(numbers), _p(ctl), from_vector) == 0
        out[tag] = (rows, ctl)
    assert_bits_equal(out["immediate"][0], out["reference"][0], True, f"{name}: voice rows")
    assert_bits_equal(out["immediate"][1], out["reference"][1], True, f"{name}: controllers from vector {from_vector}")
    first = abs(from_vector)
    assert np.abs(out["reference"][1][:, first * 64:]).max() > 0 and not out["reference"][1][:, :first * 64].any()
