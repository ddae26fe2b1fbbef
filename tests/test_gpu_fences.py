"""Fences: ordering between two engines (= two HIP streams) of one device (include/mlgpu.h, mlgpu_fence).
The reference has nothing to mirror here - a SignalProcessor's process() is one host thread (source/app/MLSignalProcessor.h:126-149);
this is what lets the HBM-bound EventsToSignals kernel of block k + 1 run under the VALU-bound voice kernel of block k."""
import numpy as np
import pytest

from inputs import assert_bits_equal
from madronalib_amd.constants import Layout, Proc

pytestmark = pytest.mark.gpu


@pytest.fixture()
def engines():
    import madronalib_amd as ml
    a, b = ml.Engine(0), ml.Engine(0)
    yield a, b
    b.close()
    a.close()


def _producer(eng, V):
    src = eng.bank([Proc.NOISE_GEN], V)
    src.set_state(0, 0, np.arange(1, V + 1, dtype=np.uint32))
    return src


def test_two_streams_ping_pong_equals_one_stream(engines):
    """Producer on engine B, consumer on engine A, two buffers, fences either way round: the consumer's output over
    12 blocks is the one-engine result, bit for bit (a missing wait would let the producer overwrite a buffer that is being read)."""
    import madronalib_amd as ml
    a, b = engines
    V, T, blocks = 32768, 16, 12
    n = V * T * 64

    COEFFS = ml.Lopass.makeCoeffs(0.05, 0.7)

    def run(two):
        pe = b if two else a
        src = _producer(pe, V)
        flt = a.bank([Proc.LOPASS] * 4, V)
        for i in range(4):
            flt.set_coeffs(i, COEFFS)
        bufs = [a.alloc(4 * n), a.alloc(4 * n)]
        ready = [pe.fence(), pe.fence()]
        free = [a.fence(), a.fence()]
        d_out = a.alloc(4 * n)
        acc = []
        for k in range(blocks):
            s = k & 1
            if two:
                pe.wait(free[s])
            src.process(T, bufs[s], Layout.QUAD)
            if two:
                pe.signal(ready[s])
                a.wait(ready[s])
            flt.process(T, d_out, Layout.QUAD, d_in=bufs[s])
            if two:
                a.signal(free[s])
            if k in (0, blocks // 2, blocks - 1):
                acc.append(d_out.download())
        return acc

    one = run(False)
    two = run(True)
    for x, y in zip(one, two):
        assert_bits_equal(x, y, "two engines with fences against one engine")


def test_fence_rules(engines):
    import madronalib_amd as ml
    a, b = engines
    f = a.fence()
    b.wait(f)                      # never signalled: nothing to wait for
    a.signal(f)
    b.wait(f)
    b.sync()
    with pytest.raises(ml.MlgpuError):
        with a.record():
            a.signal(f)
    with pytest.raises(ml.MlgpuError):
        with b.record():
            b.wait(f)
    assert a.L.mlgpu_engine_signal(a.h, None) != 0
    assert a.L.mlgpu_engine_wait(None, f.h) != 0
    f.close()
    f.close()                      # twice is fine


def test_engine_urgency():
    """mlgpu_engine_create_urgency: a second engine whose stream the dispatcher serves first (or last); results do not depend on it."""
    import ctypes
    import madronalib_amd as ml
    outs = []
    for u in (0, 1, -1):
        e = ml.Engine(0, urgency=u)
        b = _producer(e, 4096)
        d = e.alloc(4 * 4096 * 64 * 2)
        b.process(2, d, Layout.QUAD)
        outs.append(d.download())
        e.close()
    assert_bits_equal(outs[0], outs[1], "urgent engine")
    assert_bits_equal(outs[0], outs[2], "background engine")
    h = ctypes.c_void_p()
    L = ml._lib.load()
    assert L.mlgpu_engine_create_urgency(0, 2, ctypes.byref(h)) != 0 and not h.value
    assert L.mlgpu_engine_create_urgency(0, -2, ctypes.byref(h)) != 0 and not h.value
