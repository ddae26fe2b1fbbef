import ctypes, sys, os
sys.path.insert(0, os.getcwd())
import madronalib_amd as ml
eng = ml.Engine(0)
X = ctypes.CDLL('tests/cpp/libexamples_gpu.so')
X.example_reverb_gpu_open.restype = ctypes.c_void_p
X.example_reverb_gpu_open.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
X.example_reverb_gpu_graph.restype = ctypes.c_void_p
X.example_reverb_gpu_graph.argtypes = [ctypes.c_void_p]
err = ctypes.create_string_buffer(2048)
p = X.example_reverb_gpu_open(eng.h, 65536, 0, err, 2048)
gh = ctypes.c_void_p(X.example_reverb_gpu_graph(p))
L = eng.L
L.mlgpu_graph_source.restype = ctypes.c_char_p
L.mlgpu_graph_source.argtypes = [ctypes.c_void_p]
open('gpurun_out/reverb_rows_early.hip', 'wb').write(L.mlgpu_graph_source(gh))
