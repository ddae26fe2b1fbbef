"""Recorded launch sequences (mlgpu_engine_begin_recording / mlgpu_sequence_launch, a hipGraph): a real-time block made of
several small launches replayed with one graph launch gives the same bits as the launches made one by one."""
import time

import numpy as np
import pytest

from madronalib_amd.constants import Layout, Op, Proc


@pytest.fixture(scope="module")
def eng():
    import madronalib_amd as ml
    e = ml.Engine(0)
    yield e
    e.close()


def make_block(eng, V, T):
    """One real-time block: three small banks, two elementwise ops, a mix to one channel. Returns (launch-all fn, objects)."""
    import madronalib_amd as ml
    n = V * T * 64
    saw = eng.bank([Proc.SAW_GEN, Proc.BANDPASS, Proc.GAIN], V)
    saw.clear()
    co = ml.Bandpass.makeCoeffs(0.1, 0.7)
    for i in range(3):
        saw.set_coeff(1, i, float(co[i]))
    saw.set_coeff(2, 0, 0.25)
    saw.set_input_const((55.0 * 2.0 ** (5.0 * np.arange(V) / V) / 48000.0).astype(np.float32))
    noise = eng.bank([Proc.NOISE_GEN, Proc.LOPASS], V)
    noise.set_state(0, 0, np.arange(V, dtype=np.uint32) + 5)
    noise.set_coeffs(1, ml.Lopass.makeCoeffs(0.05, 0.9))
    lfo = eng.bank([Proc.SINE_GEN], V)
    lfo.clear()
    lfo.set_input_const(np.full(V, 3.0 / 48000.0, np.float32))
    bufs = [eng.alloc(4 * n) for _ in range(5)]
    d_mix = eng.alloc(4 * T * 64)
    eng.mixdown_reserve(V, T)

    def block():
        saw.process(T, bufs[0], Layout.QUAD)
        noise.process(T, bufs[1], Layout.QUAD)
        lfo.process(T, bufs[2], Layout.QUAD)
        eng.op_apply(Op.MULTIPLY, bufs[1], bufs[2], None, bufs[3], n)
        eng.op_apply(Op.ADD, bufs[0], bufs[3], None, bufs[4], n)
        eng.mixdown(bufs[4], Layout.QUAD, V, T, d_mix)
    return block, d_mix, (saw, noise, lfo, bufs)


@pytest.mark.gpu
def test_sequence_replay_same_bits(eng):
    V, T, blocks = 4096, 1, 300
    blk_a, mix_a, keep_a = make_block(eng, V, T)
    blk_b, mix_b, keep_b = make_block(eng, V, T)
    # a: launches one by one
    outs_a = []
    for _ in range(6):
        blk_a()
        outs_a.append(mix_a.download(np.float32, T * 64).copy())
    # b: recorded once, replayed
    with eng.record() as seq:
        blk_b()
    assert seq.num_nodes >= 6
    outs_b = []
    for _ in range(6):
        seq.launch()
        outs_b.append(mix_b.download(np.float32, T * 64).copy())
    for a, b in zip(outs_a, outs_b):
        assert (a.view(np.uint32) == b.view(np.uint32)).all()
    assert np.abs(outs_a[-1]).max() > 0
    # cost per block (reported, not asserted: on ROCm 7.2 a graph launch of 7 kernel nodes is not cheaper than 7 launches):
    # throughput = many blocks back to back, latency = one block and wait
    def per_block(fn, wait_each):
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(blocks):
            fn()
            if wait_each:
                eng.sync()
        eng.sync()
        return (time.perf_counter() - t0) / blocks * 1e6
    res = {(name, w): per_block(fn, w) for name, fn in (("one by one", blk_a), ("sequence", seq.launch)) for w in (False, True)}
    print(f"\nblock of 7 kernels, {V} voices x {T} vector, us per block: back to back {res[('one by one', False)]:.1f} one by one / "
          f"{res[('sequence', False)]:.1f} recorded; launch + wait {res[('one by one', True)]:.1f} / {res[('sequence', True)]:.1f}")


@pytest.mark.gpu
def test_what_cannot_be_recorded(eng):
    import madronalib_amd as ml
    from madronalib_amd.constants import Region
    buf = eng.alloc(4 * 64 * 64)
    g = ml.Graph(eng, 64)
    g.add("x", "input")
    (rx,) = g.begin_region(Region.DOWNSAMPLE_2X, ["x"])
    y = g.end_region(rx, "y")
    g.add_output(y)
    g.compile()
    ev = ml.Events(eng, 4, 4)
    with pytest.raises(ml.MlgpuError):
        with eng.record():
            buf.upload(np.zeros(64 * 64, np.float32))          # waits for the device
    with pytest.raises(ml.MlgpuError):
        with eng.record():
            g.process(1, [buf], [buf])                          # counts DSPVectors
    with pytest.raises(ml.MlgpuError):
        with eng.record():
            ev.process(1, 0, [buf] + [None] * 7)                # host work per call
    with eng.record() as seq:                                   # and the engine still records after the refusals
        eng.op_apply(Op.ADD, buf, buf, None, buf, 64 * 64)
    seq.launch()
    eng.sync()
