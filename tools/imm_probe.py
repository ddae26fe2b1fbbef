import ctypes, numpy as np, sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from inputs import lcg_noise
c_f32p = ctypes.POINTER(ctypes.c_float)
L = ctypes.CDLL("tests/cpp/libdropin_imm.so")
V, T = 2, 6
S = 64 * T
n = lambda s: lcg_noise(np.arange(V, dtype=np.uint32) + s, S)
a, b, sel = n(1), n(2), np.abs(n(3)) * 0.999
p = lambda x: x.ctypes.data_as(c_f32p)
name = sys.argv[1]
args = {"ops_ref_run": (a, b), "routing_ref_run": (a, b, sel), "objects_ref_run": (a, np.abs(b)), "hostdata_ref_run": (a,)}[name]
f = getattr(L, name)
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_size_t, ctypes.c_size_t] + [c_f32p] * (len(args) + 1)
out = np.zeros(64 * V * S, np.float32)
print(name, f(V, T, *[p(x) for x in args], p(out)))
