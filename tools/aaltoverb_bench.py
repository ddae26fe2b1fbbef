"""Developer tool: throughput of the reference's own reverb example (examples/audio-and-midi/reverb.cpp, compiled unchanged
against the shim: tests/cpp/libexamples_gpu.so) for V independent reverbs.   python tools/aaltoverb_bench.py [V] [T] [options]"""
import ctypes
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
L = ctypes.CDLL(os.path.join(ROOT, "tests", "cpp", "libexamples_gpu.so"))
L.example_reverb_gpu_bench.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.c_char_p, ctypes.c_size_t]
V = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
options = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # bit 0: windowed rings, bit 1: online tuning, bit 2: live constants
ms = ctypes.c_float()
err = ctypes.create_string_buffer(2048)
st = L.example_reverb_gpu_bench(V, T, 20, options, ctypes.byref(ms), err, 2048)
if st:
    raise SystemExit(f"failed: {st} {err.value.decode()}")
rate = V * T * 64 / (ms.value * 1e-3)
print(f"Aaltoverb example: {V} reverbs x {T} vectors: {ms.value:.3f} ms per launch = {rate:.3e} reverb-samples/s = {rate / 48000:.0f} stereo reverbs in real time at 48 kHz")
