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


REF = "/root/reference"
COMPAT = os.path.join(ROOT, "include", "mlgpu", "compat")


@pytest.mark.parametrize("example", ["sine", "reverb", "fdtd", "controllers-to-audio"])
def test_reference_examples_compile_against_the_shim_alone(example, tmp_path):
    """The reference's example programs compile against include/mlgpu/compat with NO madronalib directory on the include path: the
    shim brings its own scalar helpers, intervals and projections (mlscalar.h). Only the example's source file comes from the
    reference checkout (params.cpp keeps madronalib's parameter layer and is not in this list)."""
    src = os.path.join(REF, "examples", "audio-and-midi", example + ".cpp")
    if not os.path.exists(src):
        pytest.skip("no reference checkout here")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-c", "-w", "-I" + COMPAT, "-Dmain=example_main", src, "-o", str(tmp_path / "x.o")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    used = subprocess.run(["g++", "-std=c++17", "-M", "-w", "-I" + COMPAT, src], capture_output=True, text=True, timeout=300).stdout
    assert "/root/reference/source" not in used and "mlscalar.h" in used


@pytest.mark.parametrize("test_file", ["dspOpsTest", "dspGensTest", "dspFiltersTest", "dspBufferTest"])
def test_reference_unit_tests_compile_against_the_shim_alone(test_file, tmp_path):
    """madronalib's own DSP unit tests include its headers by name (MLDSPOps.h, MLDSPScalarMath.h, MLDSPProjections.h ...): with
    include/mlgpu/compat/dsp and include/mlgpu/compat as the only header directories (plus the tests' own directory for Catch) they
    compile; tests/test_gpu_immediate.py runs the linked program on the device."""
    src = os.path.join(REF, "Tests", test_file + ".cpp")
    if not os.path.exists(src):
        pytest.skip("no reference checkout here")
    inc = ["-I" + os.path.join(COMPAT, "dsp"), "-I" + COMPAT, "-I" + os.path.join(REF, "Tests")]
    r = subprocess.run(["g++", "-std=c++17", "-O0", "-fsyntax-only", "-w"] + inc + [src], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    used = subprocess.run(["g++", "-std=c++17", "-M", "-w"] + inc + [src], capture_output=True, text=True, timeout=300).stdout
    assert "/root/reference/source" not in used


SCALAR_PROBE = r"""
#include <cstdio>
#include <cstring>
#include <cstdint>
#include HEADER
using namespace ml;
static void p(const char* what, double v) { uint64_t u; std::memcpy(&u, &v, 8); std::printf("%s %016llx\n", what, (unsigned long long)u); }
static void pf(const char* what, float v) { uint32_t u; std::memcpy(&u, &v, 4); std::printf("%s %08x\n", what, u); }
int main()
{
  for (int i = 0; i <= 40; ++i)
  {
    const double x = 0.013 + 0.37 * i;
    p("sqrt", const_math::sqrt(x)); p("sin", const_math::sin(x - 5.0)); p("cos", const_math::cos(x - 5.0)); p("sinh", const_math::sinh(0.2 * x - 1.0));
    p("cosh", const_math::cosh(0.2 * x - 1.0)); p("exp", const_math::exp(x - 6.0)); p("log", const_math::log(x * 37.0)); p("atan", const_math::atan(x - 7.0));
    p("atan2", const_math::atan2(x - 7.0, 3.0 - x)); p("pow", const_math::pow(x, i % 9 - 3)); p("nearest", const_math::nearest(x)); p("fraction", const_math::fraction(x));
    p("mantissa", const_math::mantissa(x * 1e-3)); p("exponent", (double)const_math::exponent(x * 1e5));
    const float f = (float)x - 6.f;
    pf("clamp", clamp(f, -1.f, 2.f)); pf("lerp", lerp(f, 3.f, 0.3f)); pf("min", (min)(f, 1.f)); pf("max", (max)(f, 1.f)); pf("modf", modulo(f, 1.7f));
    pf("smoothstep", smoothstep(-2.f, 3.f, f)); pf("fSignBit", fSignBit(f)); pf("lerpBipolar", lerpBipolar(-1.f, 0.5f, 2.f, f * 0.2f)); pf("amp", dBToAmp(f)); pf("dB", ampTodB(0.1f + (float)x));
    std::printf("ints %d %d %d %d %d %d\n", (int)bitsToContain(i * 37), chunkSizeToContain(4, i * 7), modulo(i - 20, 7), ilog2(1 + i * i * i), sign(f), (int)within(f, -1.f, 1.f));
    pf("log01", projections::log({110.f, 440.f})(0.025f * i)); pf("exp01", projections::exp({110.f, 440.f})(0.025f * i)); pf("u2l", projections::unityToLogParam({0.8f, 20.f})(0.025f * i));
    pf("l2u", projections::logParamToUnity({0.8f, 20.f})(0.8f + 0.48f * i)); pf("lin", projections::linear({0.f, 3.f}, {-1.f, 7.f})(f)); pf("pw", projections::piecewiseLinear({0.f, 2.f, -1.f, 5.f})(0.025f * i));
    pf("pws", projections::piecewise({0.f, 2.f, -1.f}, {projections::easeIn, projections::easeOutCubic})(0.025f * i)); pf("bell", projections::bell(0.025f * i)); pf("eio", projections::easeInOutQuartic(0.025f * i));
    pf("invbi", projections::invBisquared(f)); pf("flat", projections::flatcenter(0.025f * i)); pf("mid", midpoint(Interval{f, 3.f}));
  }
  RandomScalarSource r; r.seed_ = 12345u;
  for (int i = 0; i < 8; ++i) { pf("rand", r.getFloat()); std::printf("bits %08x\n", r.getUInt32()); }
  float t[4] = {0.1f, -0.4f, 0.9f, 0.3f};
  pf("herp", herp(t, 0.37f));
  return 0;
}
"""


def test_scalar_helpers_match_the_reference(tmp_path):
    """mlscalar.h against the reference's own MLDSPScalarMath.h / MLDSPProjections.h: one probe program, built once with each, must
    print the same bits for a sweep of every function (const_math, the scalar templates, RandomScalarSource, the projections)."""
    if not os.path.exists(os.path.join(REF, "source", "DSP", "MLDSPProjections.h")):
        pytest.skip("no reference checkout here")
    outs = []
    for name, header, inc in (("own", '"mlscalar.h"', ["-I" + COMPAT]), ("ref", '"MLDSPProjections.h"', ["-I" + os.path.join(REF, "source", "DSP"), "-include", "iostream", "-include", "string"])):
        src = tmp_path / f"probe_{name}.cpp"
        src.write_text(SCALAR_PROBE.replace("HEADER", header))
        exe = tmp_path / f"probe_{name}"
        r = subprocess.run(["g++", "-std=c++17", "-O2", "-w"] + inc + [str(src), "-o", str(exe)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(subprocess.run([str(exe)], capture_output=True, text=True, timeout=60).stdout.splitlines())
    assert len(outs[0]) == len(outs[1]) > 1500
    bad = [(a, b) for a, b in zip(*outs) if a != b]
    assert not bad, bad[:10]
