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
    r = subprocess.run([_reftests()], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "All tests passed (42 assertions in 10 test cases)" in r.stdout, r.stdout[-2000:]


def test_reference_unit_tests_need_the_device():
    """No CPU arithmetic behind the shim: the same binary with no GPU visible fails in the DSP test cases (the DSPBuffer ones are
    host-only by nature and still pass)."""
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    r = subprocess.run([_reftests()], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    assert "All tests passed" not in r.stdout
    assert "dsp_ops" in r.stdout and "dsp_gens" in r.stdout
