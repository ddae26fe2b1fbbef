"""The C++ host mirror (include/mlgpu/mldsp_gpu.hpp): compiles on CPU against the C-ABI; on the GPU
box the reference-style test program tests/cpp/host_mirror_test.cpp must pass."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "host_mirror_test")


def _build():
    from madronalib_amd import _lib
    _lib.load()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "include", "mlgpu")], stdout=subprocess.DEVNULL)


def test_host_mirror_compiles():
    if not os.path.exists("/usr/bin/g++"):
        pytest.skip("no g++")
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_host_mirror_program_passes():
    if not os.path.exists(EXE):
        _build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "All tests passed" in r.stdout
