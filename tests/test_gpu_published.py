"""GPU parity of mlgpu_published_signal against the reference's SignalProcessor::PublishedSignal
(source/app/MLSignalProcessor.h:26-105) driven by the same script of writes and reads (oracle/dropin_ref.cpp)."""
import ctypes
import os

import numpy as np
import pytest

from inputs import lcg_noise
from madronalib_amd.constants import Layout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
c_f32p = ctypes.POINTER(ctypes.c_float)


def _ref_lib():
    so = os.path.join(ROOT, "oracle", "_ref", "libdropin_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libdropin_ref.so not available here")
    L = ctypes.CDLL(so)
    L.published_ref_run.restype = ctypes.c_int
    L.published_ref_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, c_f32p, c_f32p,
                                    ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_int, c_f32p, ctypes.POINTER(ctypes.c_size_t)]
    return L


def script(T, seed):
    """writes interleaved with read / readLatest / peekLatest of assorted sizes; all T vectors get written"""
    rng = np.random.default_rng(seed)
    ops, args, t = [], [], 0
    while t < T:
        ops.append(0), args.append(0)
        t += 1
        if rng.random() < 0.6:
            ops.append(int(rng.integers(1, 4))), args.append(int(rng.integers(1, 40)))
    ops += [3, 2, 1]
    args += [16, 8, 100]
    return np.array(ops, np.int32), np.array(args, np.int32)


def reference(L, max_frames, max_voices, octaves, ch0, ch1, ops, args):
    nV, T = ch0.shape[0], ch0.shape[1] // 64
    out = np.zeros(int(args.sum()) * 2 + 8, np.float32)
    counts = np.zeros(len(ops), np.uint64)
    assert L.published_ref_run(max_frames, max_voices, octaves, nV, T, ch0.ctypes.data_as(c_f32p), ch1.ctypes.data_as(c_f32p),
                               ops.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), args.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), len(ops),
                               out.ctypes.data_as(c_f32p), counts.ctypes.data_as(ctypes.POINTER(ctypes.c_size_t))) == 0
    return out, counts


@pytest.fixture(scope="module")
def eng():
    import madronalib_amd as ml
    e = ml.Engine(0)
    yield e
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("octaves,max_frames,layout", [(0, 64, Layout.QUAD), (2, 64, Layout.QUAD), (3, 16, Layout.VOICE_MAJOR), (6, 4, Layout.ROWS)])
def test_published_signal_same_script_same_floats(eng, octaves, max_frames, layout):
    import madronalib_amd as ml
    L = _ref_lib()
    Vtotal, first, nV, T = 300, 37, 3, 12          # publish 3 of 300 voices
    sig0 = lcg_noise(np.arange(Vtotal, dtype=np.uint32) + 9, 64 * T)
    sig1 = lcg_noise(np.arange(Vtotal, dtype=np.uint32) + 4009, 64 * T)
    ops, args = script(T, seed=octaves)
    want, counts = reference(L, max_frames, nV, octaves, np.ascontiguousarray(sig0[first:first + nV]), np.ascontiguousarray(sig1[first:first + nV]), ops, args)
    ps = ml.PublishedSignal(eng, max_frames, nV, 2, octaves)
    got, t, pos = np.zeros_like(want), 0, 0
    one = [eng.alloc(4 * Vtotal * 64), eng.alloc(4 * Vtotal * 64)]
    stage = eng.alloc(4 * Vtotal * 64)
    for i, (op, arg) in enumerate(zip(ops, args)):
        if op == 0:
            for d, s in zip(one, (sig0, sig1)):
                stage.upload(np.ascontiguousarray(s[:, 64 * t:64 * (t + 1)]))
                eng.layout_convert(stage, Layout.VOICE_MAJOR, d, layout, Vtotal, 1)
            ps.write(1, one, Vtotal, first, nV, layout)
            t += 1
            continue
        if op == 3:
            out, n = ps.peek_latest(arg), arg * 2
        else:
            out, n = (ps.read if op == 1 else ps.read_latest)(arg)
        assert n == counts[i], (i, op, arg, n, counts[i])
        got[pos:pos + arg * 2] = out
        pos += arg * 2
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
    assert ps.read_available() == 0


@pytest.mark.gpu
def test_published_signal_many_vectors_per_write(eng):
    """One write of T vectors == T writes of one vector (the gather keeps the reference's [vector][voice][frame][channel])."""
    import madronalib_amd as ml
    V, T, octaves = 5, 6, 1
    sig = [lcg_noise(np.arange(V, dtype=np.uint32) + k, 64 * T) for k in (1, 2, 3)]
    d = []
    for s in sig:
        b, q = eng.alloc(4 * V * 64 * T), eng.alloc(4 * V * 64 * T)
        b.upload(s)
        eng.layout_convert(b, Layout.VOICE_MAJOR, q, Layout.QUAD, V, T)
        d.append(q)
    ps = ml.PublishedSignal(eng, 32 * T, V, 3, octaves)
    ps.write(T, d, V)
    got, n = ps.read(32 * T * V)
    assert n == 32 * T * V * 3
    want = np.stack([s.reshape(V, T, 64)[:, :, 1::2] for s in sig], -1)      # [V][T][32][3]: every second frame, from frame 1
    want = np.ascontiguousarray(want.transpose(1, 0, 2, 3)).ravel()           # [T][V][32][3]
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
    with pytest.raises(ml.MlgpuError):
        ps.write(1, d, V, first_voice=3, n_voices=4)                          # outside the signals
    with pytest.raises(ml.MlgpuError):
        ml.PublishedSignal(eng, 8, 1, 17, 0)                                  # too many channels
