"""The host ring (mlgpu_dspbuffer) against the reference's own DSPBuffer (compiled, oracle/_ref) on random operation
sequences, and against the reference's test assertions (Tests/dspBufferTest.cpp). Host-only: runs without a GPU."""
import numpy as np
import pytest

import madronalib_amd as ml


def test_resize_rounds_up_to_pow2_with_a_floor_of_64():
    # MLDSPBuffer.h:104-133
    for n, want in ((0, 64), (1, 64), (64, 64), (65, 128), (100, 128), (256, 256), (257, 512), (1000, 1024), (4096, 4096)):
        assert ml.DSPBuffer(n).size == want


def test_reference_buffer_test_assertions():
    """Tests/dspBufferTest.cpp: write / read round trip, wrap, overflow keeps the newest data, vector reads."""
    b = ml.DSPBuffer(256)
    x = np.arange(100, dtype=np.float32)
    b.write(x)
    assert b.read_available() == 100 and b.write_available() == 156
    assert (b.read(100) == x).all() and b.read_available() == 0
    # wrap around many times
    for k in range(40):
        chunk = np.arange(k * 37, k * 37 + 37, dtype=np.float32)
        b.write(chunk)
        assert (b.read(37) == chunk).all()
    # overflow: the oldest data is clobbered, the ring reports full
    b = ml.DSPBuffer(64)
    b.write(np.arange(64, dtype=np.float32))
    b.write(np.arange(64, 80, dtype=np.float32))
    assert b.read_available() == 64
    assert (b.read(64) == np.arange(16, 80, dtype=np.float32)).all()
    # DSPVector read(): nothing (zeros) when fewer than 64 samples wait
    b.write(np.ones(63, np.float32))
    ok, v = b.read_vector()
    assert not ok and (v == 0).all() and b.read_available() == 63
    b.write(np.ones(1, np.float32))
    ok, v = b.read_vector()
    assert ok and (v == 1).all() and b.read_available() == 0


@pytest.mark.parametrize("size", [64, 128, 1024])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_operation_sequences_match_the_reference(ref, size, seed):
    rng = np.random.default_rng(seed)
    a, r = ml.DSPBuffer(size), ref.dspbuffer(size)
    for step in range(3000):
        op = rng.integers(0, 8)
        n = int(rng.integers(0, size + 1))   # the reference overruns its storage for n > size
        if op in (0, 1, 2):
            x = rng.standard_normal(n).astype(np.float32)
            a.write(x), r.write(x)
        elif op == 3:
            assert (a.read(n).view(np.uint32) == r.read(n).view(np.uint32)).all(), step
        elif op == 4:
            a.discard(n), r.discard(n)
        elif op == 5:
            m = min(n, a.read_available())
            if m:
                assert (a.peek_most_recent(m) == r.peek_most_recent(m)).all(), step
        elif op == 6:
            w = int(rng.integers(2, max(3, size // 4)))
            ov = int(rng.integers(0, w))
            x = rng.standard_normal(w).astype(np.float32)
            a.write_with_overlap_add(x, ov), r.write_with_overlap_add(x, ov)
        else:
            w = int(rng.integers(1, max(2, size // 4)))
            ov = int(rng.integers(0, w))
            if a.read_available() + ov >= w:   # both implementations read uninitialised tail otherwise
                assert (a.read_with_overlap(w, ov).view(np.uint32) == r.read_with_overlap(w, ov).view(np.uint32)).all(), step
        assert a.read_available() == r.read_available(), (step, op)
        assert a.write_available() == r.write_available(), (step, op)
        if step % 500 == 499:
            a.clear(), r.clear()


def test_spsc_threads():
    """One writer thread, one reader thread, no lock (the use the reference's atomics are for)."""
    import threading
    b = ml.DSPBuffer(4096)
    total, chunk = 200000, 97
    got = []

    def writer():
        pos = 0
        while pos < total:
            n = min(chunk, total - pos)
            if b.write_available() >= n:
                b.write(np.arange(pos, pos + n, dtype=np.float32))
                pos += n

    def reader():
        have = 0
        while have < total:
            x = b.read(128)
            if x.size:
                got.append(x)
                have += x.size

    tw, tr = threading.Thread(target=writer), threading.Thread(target=reader)
    tw.start(), tr.start()
    tw.join(30), tr.join(30)
    assert (np.concatenate(got) == np.arange(total, dtype=np.float32)).all()


def test_windows_match_the_reference(ref):
    """makeWindow(dest, size, dspwindows::shape) for the six shapes and a few sizes (a DSPVector, an odd size, 1): the C-ABI's host
    tables are the reference's floats (host libm on both sides)."""
    import ctypes
    c_f32p = ctypes.POINTER(ctypes.c_float)
    ref.lib.mlref_make_window.restype = None
    ref.lib.mlref_make_window.argtypes = [c_f32p, ctypes.c_size_t, ctypes.c_int]
    for name, shape in ml.WINDOW_SHAPES.items():
        for size in (64, 37, 256, 2, 1):
            want = np.empty(size, np.float32)
            ref.lib.mlref_make_window(want.ctypes.data_as(c_f32p), size, shape)
            got = ml.make_window(size, name)
            assert (got.view(np.uint32) == want.view(np.uint32)).all(), (name, size)
    t = ml.make_window(64, "triangle")
    assert t[0] == 0 and abs(t[31] - t[32]) < 1e-6 and t.max() <= 1.0
    with pytest.raises(ValueError):
        ml.make_window(8, 9)


def test_overlap_add_of_triangle_windows_is_constant(ref):
    """Tests/dspBufferTest.cpp "overlap": eight triangle windows of one DSPVector written with half-vector overlap; past the
    start-up the sums are constant - and every sample equals what the reference's DSPBuffer holds."""
    w = ml.make_window(64, "triangle")
    b, rb = ml.DSPBuffer(256), ref.dspbuffer(256)
    for _ in range(8):
        b.write_with_overlap_add(w, 32)
        rb.write_with_overlap_add(w, 32)
    outs, routs = [], []
    for _ in range(3):
        ok, v = b.read_vector()
        assert ok
        outs.append(v)
        routs.append(rb.read(64))
    assert (outs[1] == outs[2]).all()                       # the reference test's REQUIRE(outputVec == outputVec2)
    for a, r in zip(outs, routs):
        assert (np.asarray(a).view(np.uint32) == np.asarray(r).view(np.uint32)).all()
