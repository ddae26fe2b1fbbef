"""The source shim's immediate mode without a GPU: the reference's own unit tests compile unchanged against the shim
(include/mlgpu/Makefile: reftests - madronalib's header names are forwarded by include/mlgpu/compat/dsp) and, with no device, every
test case that does DSPVector arithmetic fails with the engine's error - there is no CPU arithmetic behind the shim. The GPU half is
tests/test_gpu_immediate.py."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "reftests_gpu")


def test_reference_unit_tests_compile_unchanged_and_fail_loudly_without_a_device():
    import madronalib_amd as ml
    if os.path.isdir("/root/reference/Tests"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "include", "mlgpu"), "reftests"], stdout=subprocess.DEVNULL)
    if not os.path.exists(EXE):
        pytest.skip("tests/cpp/reftests_gpu is built from the reference's Tests/ where that checkout exists")
    if ml.device_count() > 0:
        pytest.skip("a GPU is visible here (tests/test_gpu_immediate.py runs the binary on it)")
    # (without the reference's two-thread DSPBuffer case: host-only, and a race by construction - see tests/test_gpu_immediate.py)
    r = subprocess.run([EXE, "~[threads]"], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "no gfx950 (MI355X) HIP device" in r.stdout
    assert "All tests passed" not in r.stdout
    # the host-only cases (DSPBuffer sizes, overlap-add, std::vector<DSPVector>) still pass
    assert "test cases:  9 |  4 passed | 5 failed" in r.stdout, r.stdout[-600:]


def test_forwarding_headers_cover_the_dsp_headers_user_code_includes():
    have = set(os.listdir(os.path.join(ROOT, "include", "mlgpu", "compat", "dsp")))
    for h in ("MLDSPOps.h", "MLDSPFilters.h", "MLDSPGens.h", "MLDSPBuffer.h", "MLDSPFunctional.h", "MLDSPUtils.h", "MLDSPRouting.h", "MLDSPSample.h",
              "MLDSPMath.h"):
        assert h in have
        assert '#include "../mldsp.h"' in open(os.path.join(ROOT, "include", "mlgpu", "compat", "dsp", h)).read()


def test_events_to_signals_object_needs_the_device_too():
    """oracle/dropin_ref.cpp's EventsToSignals driver compiled against the shim (tests/cpp/libdropin_imm.so): the symbols are there, and
    without a device stepping the object ends in the engine's error (a C++ exception out of an extern "C" call: the process dies) - no
    CPU restatement behind the class."""
    import sys
    import madronalib_amd as ml
    lib = os.path.join(ROOT, "tests", "cpp", "libdropin_imm.so")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")], stdout=subprocess.DEVNULL)
    syms = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    for name in ("e2s_ref_run", "e2s_ref_run_controllers", "e2s_ref_run_controllers_from", "immediate_ref_run", "plugin_ref_run", "spb_ref_run",
                 "controller_synth_ref_run", "tempo_synth_ref_run", "lean_synth_ref_run", "ctl_audio_ref_run", "transport_ref_run"):
        assert f" T {name}\n" in syms
    if ml.device_count() > 0:
        pytest.skip("a GPU is visible here (tests/test_gpu_immediate.py steps the object on it)")
    code = ("import ctypes, numpy as np\n"
            f"L = ctypes.CDLL({lib!r})\n"
            "out = np.zeros((8, 2, 64), np.float32)\n"
            "L.e2s_ref_run.argtypes = [ctypes.c_int] * 3 + [ctypes.c_double] + [ctypes.c_float] * 4 + [ctypes.c_int, ctypes.c_void_p] + [ctypes.c_int] * 3 + [ctypes.c_void_p]\n"
            "print('rc', L.e2s_ref_run(2, 0, 0, 48000.0, 0.0, 0.0, 7.0, 24.0, 16, None, 0, 64, 1, out.ctypes.data))\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "rc 0" not in r.stdout
    assert "no gfx950 (MI355X) HIP device" in r.stderr


def test_a_context_that_is_only_configured_never_touches_a_device(tmp_path):
    """An AudioContext handed to a capture is constructed, configured and fed events on the host (gpu::SynthProgram takes it from there);
    only a context that is STEPPED outside a capture (processVector) becomes an immediate one with device objects of its own. The first
    must work on a machine without a GPU, the second must fail there with the engine's error."""
    import madronalib_amd as ml
    if ml.device_count() > 0:
        pytest.skip("a GPU is visible here")
    src = tmp_path / "ctx.cpp"
    src.write_text(r'''
#include "MLAudioContext.h"
#include <cstdio>
#include <cstring>
using namespace ml;
int main(int argc, char** argv)
{
  AudioContext ctx(0, 2, 48000);
  ctx.setInputPolyphony(4);
  ctx.setInputGlideTimeInSeconds(0.01f);
  ctx.setInputDriftAmount(0.5f);
  ctx.setInputProtocol(Symbol("MIDI"));
  ctx.updateTime(0.0, 120.0, true, 48000.0);
  Event e;
  e.type = kNoteOn; e.channel = 1; e.sourceIdx = 60; e.time = 0; e.value1 = 60.f; e.value2 = 0.5f;
  ctx.addInputEvent(e);
  ctx.clearInputEvents();
  SignalProcessBuffer buffer(0, 2, 512);
  std::printf("configured %zu\n", ctx.getInputPolyphony());
  if (argc > 1 && !std::strcmp(argv[1], "step"))
  {
    try { ctx.processVector(0); }
    catch (const std::exception& ex) { std::printf("stepping failed: %s\n", ex.what()); return 3; }
  }
  return 0;
}
''')
    exe = tmp_path / "ctx"
    lib = os.path.join(ROOT, "madronalib_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-w", "-I" + os.path.join(ROOT, "include", "mlgpu", "compat"), str(src), "-o", str(exe),
                           "-L" + lib, "-lmlgpu", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "configured 4" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([str(exe), "step"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "no gfx950 (MI355X) HIP device" in r.stdout, r.stdout + r.stderr
