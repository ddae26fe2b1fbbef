"""The C++ host mirror (include/mlgpu/mldsp_gpu.hpp): compiles on CPU against the C-ABI; on the GPU
box the reference-style test program tests/cpp/host_mirror_test.cpp must pass."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "host_mirror_test")
MULTI = os.path.join(ROOT, "tests", "cpp", "multi_engine_test")


def _build():
    from madronalib_amd import _lib
    _lib.load()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "include", "mlgpu")], stdout=subprocess.DEVNULL)


def test_host_mirror_compiles():
    if not os.path.exists("/usr/bin/g++"):
        pytest.skip("no g++")
    _build()
    assert os.path.exists(EXE)
    assert os.path.exists(MULTI)


def test_multi_engine_program_fails_loudly_without_a_gpu():
    """No CPU fallback in the multi-device host either: without a GPU the program says so and exits non-zero."""
    if not os.path.exists(MULTI):
        _build()
    from madronalib_amd import _lib
    if _lib.load().mlgpu_device_count() > 0:
        pytest.skip("a GPU is visible")
    r = subprocess.run([MULTI], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "no GPU" in r.stdout


@pytest.mark.gpu
def test_host_mirror_program_passes():
    if not os.path.exists(EXE):
        _build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "All tests passed" in r.stdout


@pytest.mark.gpu
def test_multi_engine_program_passes():
    """ml::gpu::DeviceGroup (one host thread + engine + stream per device) through the C-ABI: union of the shards ==
    unsharded, outputs and state; two devices when the box has them, else one device and two engines on two threads."""
    if not os.path.exists(MULTI):
        _build()
    r = subprocess.run([MULTI], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "All tests passed" in r.stdout


@pytest.mark.parametrize("sanitizer,mode", [("address,undefined", ""), ("thread", "threads")])
def test_host_code_under_sanitizers(sanitizer, mode, tmp_path):
    """SURVEY §5 (sanitizer builds): the pure-host translation units (the DSPBuffer ring, the coefficient makers) built with
    g++ -fsanitize=address,undefined and -fsanitize=thread and driven by random operation sequences and a two-thread
    producer / consumer run (tests/cpp/sanitize_host_test.cpp). Any report aborts the program."""
    if not os.path.exists("/usr/bin/g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "sanitize_host_test")
    src = [os.path.join(ROOT, "tests", "cpp", "sanitize_host_test.cpp"), os.path.join(ROOT, "madronalib_amd", "csrc", "dspbuffer.cpp"),
           os.path.join(ROOT, "madronalib_amd", "csrc", "coeffs.cpp")]
    cmd = ["g++", "-std=c++17", "-O1", "-g", f"-fsanitize={sanitizer}", "-fno-sanitize-recover=all", "-D__HIP_PLATFORM_AMD__",
           "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include")] + src + ["-o", exe, "-pthread"]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if b.returncode != 0 and ("cannot find" in b.stderr or "unrecognized" in b.stderr):
        pytest.skip("sanitizer runtime not installed: " + b.stderr[-200:])
    assert b.returncode == 0, b.stderr[-2000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=1")
    r = subprocess.run([exe] + ([mode] if mode else []), capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "All tests passed" in r.stdout, (r.stdout + r.stderr)[-3000:]
