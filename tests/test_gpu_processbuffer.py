"""mlgpu_process_buffer (the engine's SignalProcessBuffer) against the reference's own SignalProcessBuffer driven
with the same sequence of host block sizes (oracle/_ref/libdropin_ref.so: spb_ref_run), plus mixdown / broadcast."""
import ctypes
import os

import numpy as np
import pytest

from inputs import assert_bits_equal, lcg_noise
from madronalib_amd.constants import Layout, Op, Proc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
c_f32p = ctypes.POINTER(ctypes.c_float)


@pytest.fixture(scope="module")
def eng():
    import madronalib_amd as ml
    e = ml.Engine(0)
    yield e
    e.close()


def _spb_ref(max_frames, blocks, x):
    so = os.path.join(ROOT, "oracle", "_ref", "libdropin_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libdropin_ref.so not available here")
    L = ctypes.CDLL(so)
    L.spb_ref_run.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int, c_f32p, c_f32p, c_f32p]
    o0, o1 = np.zeros_like(x), np.zeros_like(x)
    arr = (ctypes.c_int * len(blocks))(*blocks)
    assert L.spb_ref_run(max_frames, arr, len(blocks), x.ctypes.data_as(c_f32p), o0.ctypes.data_as(c_f32p), o1.ctypes.data_as(c_f32p)) == 0
    return o0, o1


@pytest.mark.gpu
@pytest.mark.parametrize("max_frames,blocks", [(512, [64] * 10), (512, [100, 28, 64, 1, 511, 512, 3, 200, 64, 64, 77]),
                                               (1024, [1000, 24, 1024, 5, 5, 5, 900]), (64, [64, 64, 10, 54, 64])])
def test_block_adaptor_matches_reference_signal_process_buffer(eng, max_frames, blocks):
    import madronalib_amd as ml
    total = sum(blocks)
    x = lcg_noise(np.array([7], np.uint32), total)[0]
    want0, want1 = _spb_ref(max_frames, blocks, x)
    # the same process function as a 1-voice graph: out0 = Lopass(in0), out1 = in0 * 0.5
    desc = [dict(name="x", type="input"), dict(name="half", type="const", value=0.5),
            dict(name="lp", type="proc", kind=Proc.LOPASS, inputs=["x"]), dict(name="y1", type="op", kind=Op.MULTIPLY, inputs=["x", "half"])]
    g = ml.Graph(eng, 1, desc, ["lp", "y1"])
    g.set_coeffs("lp", ml.Lopass.makeCoeffs(0.05, 0.9))
    pb = ml.ProcessBuffer(eng, 1, 2, max_frames)
    calls = []

    def fn(n_vectors, d_in, d_out):
        calls.append(n_vectors)
        g.process(n_vectors, d_in, d_out, Layout.VOICE_MAJOR, Layout.VOICE_MAJOR)

    got0, got1, pos = [], [], 0
    for b in blocks:
        o = pb.process([x[pos:pos + b]], b, fn)
        got0.append(o[0]), got1.append(o[1])
        pos += b
    assert_bits_equal(np.concatenate(got0), want0, True, "SignalProcessBuffer out0")
    assert_bits_equal(np.concatenate(got1), want1, True, "SignalProcessBuffer out1")
    assert len(calls) <= len(blocks)           # at most ONE callback per block (the reference: one per 64 frames)
    assert sum(calls) * 64 >= total
    with pytest.raises(ml.MlgpuError):
        pb.process([x[:1]], max_frames + 1, fn)   # larger than max_frames: refused (the reference silently returns)


@pytest.mark.gpu
def _overdrives_the_reference_rings(max_frames, blocks):
    """SignalProcessBuffer's rings hold nextpow2(max_frames) frames; a block that arrives while they still hold leftovers of an
    out-of-phase one overwrites the oldest samples (MLDSPBuffer.h:162-167) - the reference's own glitch, which the synchronous
    mode reproduces (previous test) and the pipelined mode, with its larger rings, does not have."""
    ring, have = 1 << int(np.ceil(np.log2(max(max_frames, 64)))), 0
    for b in blocks:
        while have < b:
            have += 64
            if have > ring:
                return True
        have -= b
    return False


@pytest.mark.gpu
@pytest.mark.parametrize("max_frames,blocks", [(512, [64] * 12), (512, [100, 28, 64, 1, 511, 512, 3, 200, 64, 64, 77, 448, 64, 128]),
                                               (1024, [1000, 24, 1024, 5, 5, 5, 900, 45, 1024]), (64, [64, 64, 10, 54, 64, 64])])
def test_pipelined_mode_is_the_synchronous_stream_delayed(eng, max_frames, blocks):
    """mlgpu_process_buffer_set_pipelined: two staging sets, a call returns before its own block has run. The output stream
    must be EXACTLY the synchronous mode's (== the reference's SignalProcessBuffer, previous test) delayed by
    latency_frames(), silence first - for any sequence of host block sizes that does not overdrive the reference's rings."""
    import madronalib_amd as ml
    assert not _overdrives_the_reference_rings(max_frames, blocks)
    total = sum(blocks)
    x = lcg_noise(np.array([11], np.uint32), total)[0]
    desc = [dict(name="x", type="input"), dict(name="half", type="const", value=0.5),
            dict(name="lp", type="proc", kind=Proc.LOPASS, inputs=["x"]), dict(name="y1", type="op", kind=Op.MULTIPLY, inputs=["x", "half"])]

    def run(pipelined):
        g = ml.Graph(eng, 1, desc, ["lp", "y1"])
        g.set_coeffs("lp", ml.Lopass.makeCoeffs(0.05, 0.9))
        pb = ml.ProcessBuffer(eng, 1, 2, max_frames)
        if pipelined:
            pb.set_pipelined(True)
        calls = []

        def fn(n_vectors, d_in, d_out):
            calls.append(n_vectors)
            g.process(n_vectors, d_in, d_out, Layout.VOICE_MAJOR, Layout.VOICE_MAJOR)
        outs, pos = [[], []], 0
        for b in blocks:
            o = pb.process([x[pos:pos + b]], b, fn)
            outs[0].append(o[0]), outs[1].append(o[1])
            pos += b
        lat = pb.latency_frames()
        pb.close()
        g.close()
        return np.concatenate(outs[0]), np.concatenate(outs[1]), calls, lat
    s0, s1, scalls, slat = run(False)
    p0, p1, pcalls, plat = run(True)
    assert slat == 0 and plat == 64 * ((max_frames + 63) // 64 + 1)
    assert pcalls == scalls                                  # the same vectors are computed by the same calls
    assert (p0[:plat] == 0).all() and (p1[:plat] == 0).all() if total > plat else True
    n = total - plat
    if n > 0:
        assert_bits_equal(p0[plat:], s0[:n], True, "pipelined out0 == synchronous out0 delayed")
        assert_bits_equal(p1[plat:], s1[:n], True, "pipelined out1 == synchronous out1 delayed")
    assert np.abs(s0).max() > 0.01


@pytest.mark.gpu
@pytest.mark.parametrize("V,T", [(1, 2), (64, 3), (100, 2), (1000, 4), (4097, 1), (262144 + 4096 + 77, 1)])   # 65 groups: two passes of 64 rows; 4162: three
@pytest.mark.parametrize("layout", [Layout.QUAD, Layout.ROWS, Layout.VOICE_MAJOR])
def test_mixdown_vs_oracle(eng, oracle, V, T, layout):
    sig = lcg_noise(np.arange(V, dtype=np.uint32) + 3, 64 * T)
    gains = np.random.default_rng(V).uniform(-1, 1, V).astype(np.float32)
    d_vm = eng.to_device(sig)
    d_sig = d_vm
    if layout != Layout.VOICE_MAJOR:
        d_sig = eng.alloc(sig.nbytes)
        eng.layout_convert(d_vm, Layout.VOICE_MAJOR, d_sig, layout, V, T)
    d_out = eng.alloc(4 * 64 * T)
    if (V, T) == (4097, 1) and layout == Layout.QUAD:
        import madronalib_amd as ml
        small = ml.Engine(0)            # a fresh engine has reserved nothing: a process call refuses instead of allocating
        with pytest.raises(ml.MlgpuError) as ei:
            small.mixdown(d_sig, layout, V, T, d_out)
        assert ei.value.status == ml.Status.ERR_INVALID and "mixdown_reserve" in str(ei.value)
        small.close()
    eng.mixdown_reserve(V, T)           # setup time
    for g in (None, gains):
        eng.mixdown(d_sig, layout, V, T, d_out, None if g is None else eng.to_device(g))
        got = d_out.download(np.float32, 64 * T)
        assert_bits_equal(got, oracle.mixdown(sig, g), True, f"mixdown V={V}")
        # vs a Synth's sequential `outputs += voice` (MLSynth.h:43-57): reassociation only. (For a quarter of a million voices the
        # sequential float sum is itself the less accurate of the two: there the yardstick is the sum in double.)
        if V <= 4097:
            seq = np.zeros(64 * T, np.float32)
            for v in range(V):
                seq = seq + (sig[v] if g is None else sig[v] * g[v])
        else:
            seq = (sig if g is None else sig * g[:, None]).astype(np.float64).sum(0)
        assert np.abs(got - seq).max() <= 1e-5 * max(1.0, np.sqrt(V)) * np.abs(sig).max()


@pytest.mark.gpu
def test_broadcast_input_layout(eng, oracle):
    """One host audio channel (a single-voice signal) feeding every voice of a bank and of a graph."""
    import madronalib_amd as ml
    V, T = 300, 5
    x = lcg_noise(np.array([1], np.uint32), 64 * T)
    co = np.stack([oracle.make_coeffs("lopass", 0.01 + 0.4 * v / V, 0.7) for v in range(V)], 1)
    st = oracle.chain_clear([Proc.LOPASS], V)
    want = oracle.chain_process([Proc.LOPASS], T, co, st, np.repeat(x, V, 0))
    d_x = eng.to_device(x)
    bank = eng.bank([Proc.LOPASS], V)
    bank.set_all_coeffs(co)
    d_out = eng.alloc(4 * V * T * 64)
    bank.process(T, d_out, Layout.VOICE_MAJOR, d_x, Layout.BROADCAST)
    assert_bits_equal(d_out.download(np.float32).reshape(V, -1), want, True, "bank, broadcast input")
    # graph: one broadcast input next to a per-voice input
    desc = [dict(name="x", type="input"), dict(name="g", type="input"), dict(name="lp", type="proc", kind=Proc.LOPASS, inputs=["x"]),
            dict(name="y", type="op", kind=Op.MULTIPLY, inputs=["lp", "g"])]
    gr = ml.Graph(eng, V, desc, ["y"])
    gr.set_coeffs("lp", [np.ascontiguousarray(r) for r in co])
    gr.set_input_layout(0, Layout.BROADCAST)
    gsig = lcg_noise(np.arange(V, dtype=np.uint32) + 50, 64 * T)
    d_g = eng.to_device(gsig)
    gr.process(T, [d_x, d_g], [d_out], Layout.VOICE_MAJOR, Layout.VOICE_MAJOR)
    assert_bits_equal(d_out.download(np.float32).reshape(V, -1), (want * gsig).astype(np.float32), True, "graph, broadcast + per-voice inputs")


@pytest.mark.gpu
@pytest.mark.parametrize("P,groups", [(1, 70), (2, 64), (2, 33), (3, 65), (4, 37), (4, 4096), (5, 1), (6, 200), (8, 131), (16, 1), (16, 63), (16, 1000),
                                      (16, 16385), (17, 129), (32, 10), (40, 66), (70, 9)])
@pytest.mark.parametrize("layout", [Layout.QUAD, Layout.VOICE_MAJOR])
def test_mixdown_groups_voice_order(eng, P, groups, layout):
    """mlgpu_mixdown_groups: every P consecutive voices summed in voice order, starting from zero (Synth::processVector,
    source/app/MLSynth.h:43-57) — the LDS-strip kernel (P <= 62) and the direct one, full and ragged blocks of 64 groups, more
    items than one trip of the grid takes; a group of negative zeros gives +0 like the reference's loop from zero."""
    from inputs import lcg_noise
    T, V = 3, groups * P
    x = lcg_noise(np.arange(V, dtype=np.uint32) + 17, 64 * T)
    d_x, d_q, d_o, d_ov = eng.alloc(4 * V * 64 * T), eng.alloc(4 * V * 64 * T), eng.alloc(4 * groups * 64 * T), eng.alloc(4 * groups * 64 * T)
    d_x.upload(x)
    eng.layout_convert(d_x, Layout.VOICE_MAJOR, d_q, layout, V, T)
    eng.mixdown_groups(d_q, layout, groups, P, T, d_o, layout)
    eng.layout_convert(d_o, layout, d_ov, Layout.VOICE_MAJOR, groups, T)
    got = d_ov.download(np.float32, groups * 64 * T).reshape(groups, 64 * T)
    want = np.zeros((groups, 64 * T), np.float32)
    xs = x.reshape(groups, P, 64 * T)
    for p in range(P):
        want = want + xs[:, p]
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
    # a group of negative zeros sums to +0 (0 + -0), whichever kernel
    d_x.upload(np.full(V * 64 * T, -0.0, np.float32))
    eng.layout_convert(d_x, Layout.VOICE_MAJOR, d_q, layout, V, T)
    eng.mixdown_groups(d_q, layout, groups, P, T, d_o, layout)
    assert (d_o.download(np.uint32, groups * 64 * T) == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("V,T", [(64, 1), (64 * 5, 3), (4096 + 64, 2), (65536 + 4096 + 192, 1),   # 1 / 5 / 65 / 1091 groups: one to three row passes
                                 (1, 2), (100, 1), (4097, 3), (262144 + 77, 1)])                    # a last wavefront that is not full: its missing voices count as +0
@pytest.mark.parametrize("chain,signal", [((Proc.SAW_GEN, Proc.BANDPASS, Proc.GAIN), False), ((Proc.SAW_GEN, Proc.BANDPASS, Proc.GAIN), True),
                                          ((Proc.SINE_GEN, Proc.GAIN), False), ((Proc.PULSE_GEN, Proc.HIPASS, Proc.ONE_POLE), True)])
def test_bank_process_mixdown_same_bits(eng, oracle, V, T, chain, signal):
    """mlgpu_bank_process_mixdown - the voices summed inside the voice kernel, their signals never written - against the two calls it
    replaces (bank_process, then mixdown without gains, whose tree the oracle pins in test_mixdown_vs_oracle): the same bits in the
    mixed signal and in every state word, over three launches; per-voice constant frequencies and a streamed frequency signal."""
    import madronalib_amd as ml
    launches = 3
    rng = np.random.default_rng(V + T)
    freq = (55.0 * 2.0 ** (5.0 * rng.random(V)) / 48000.0).astype(np.float32)
    fsig = (freq[:, None] * (1.0 + 0.3 * np.sin(np.arange(64 * T * launches)[None, :] * 0.003 * (1 + np.arange(V)[:, None] % 7)))).astype(np.float32)
    eng.mixdown_reserve(V, T)
    banks = [eng.bank(list(chain), V) for _ in range(2)]
    for b in banks:
        b.clear()
        for p, kind in enumerate(chain):
            if kind in (Proc.BANDPASS, Proc.HIPASS):
                few = np.stack([oracle.make_coeffs("bandpass" if kind == Proc.BANDPASS else "hipass", 0.02 + 0.3 * j / 16, 0.6) for j in range(16)], 1)
                b.set_coeffs(p, [np.ascontiguousarray(few[i][np.arange(V) % 16]) for i in range(few.shape[0])])
            elif kind == Proc.ONE_POLE:
                b.set_coeffs(p, [float(c) for c in oracle.make_coeffs("onepole", 0.3)])
            elif kind == Proc.GAIN:
                b.set_coeff(p, 0, 0.25)
        if not signal:
            b.set_input_const(freq)
    d_voices = eng.alloc(4 * V * T * 64)
    d_two, d_one = eng.alloc(4 * T * 64), eng.alloc(4 * T * 64)
    for k in range(launches):
        d_in = eng.to_device(np.ascontiguousarray(fsig[:, k * 64 * T:(k + 1) * 64 * T])) if signal else None
        d_g = eng.to_device(rng.uniform(-1, 1, V).astype(np.float32)) if k == 1 else None      # the second launch with per-voice gains
        banks[0].process(T, d_voices, Layout.QUAD, d_in, Layout.VOICE_MAJOR)
        eng.mixdown(d_voices, Layout.QUAD, V, T, d_two, d_g)
        banks[1].process_mixdown(T, d_one, d_in, Layout.VOICE_MAJOR, d_g)
        two, one = d_two.download(np.float32, 64 * T), d_one.download(np.float32, 64 * T)
        assert_bits_equal(one, two, True, f"bank_process_mixdown launch {k}")
        assert np.isfinite(two).all() and np.abs(two).max() > 1e-6, (k, two[:8])
    for p in range(len(chain)):
        for i in range(banks[0].num_state(p)):
            assert (banks[0].get_state(p, i) == banks[1].get_state(p, i)).all(), (p, i)


@pytest.mark.gpu
@pytest.mark.parametrize("V,shards", [(8192, 2), (8192, 8), (64 * 64 * 6, 3), (262144, 2), (64 * 72, 3)])
def test_sharded_mixdown_gives_one_bank_bits(eng, oracle, V, shards):
    """A voice bank split over several engines, the real-time block across the GPUs of a node: every shard's
    mlgpu_bank_process_mixdown_shard (its voices summed inside its voice kernel up to the hand-over level of the tree) and the host's
    mlgpu_mixdown_finish over all shards' rows give the bits ONE engine's mlgpu_bank_process_mixdown gives for all the voices - here
    with the shards as banks of one engine (tests/cpp/multi_engine_test.cpp: as engines of a DeviceGroup); two launches, state carried.
    The same for a signal in memory (mlgpu_mixdown_shard) and for a graph output (mlgpu_graph_set_output_mixdown(.., 2))."""
    import madronalib_amd as ml
    T, chain = 2, [Proc.SAW_GEN, Proc.BANDPASS, Proc.GAIN]
    per = V // shards
    rng = np.random.default_rng(V + shards)
    freq = (55.0 * 2.0 ** (5.0 * rng.random(V)) / 48000.0).astype(np.float32)
    few = np.stack([oracle.make_coeffs("bandpass", 0.02 + 0.3 * j / 16, 0.6) for j in range(16)], 1)

    def make(lo, n):
        b = eng.bank(chain, n)
        b.clear()
        b.set_coeffs(1, [np.ascontiguousarray(few[i][np.arange(lo, lo + n) % 16]) for i in range(3)])
        b.set_coeff(2, 0, 0.25)
        b.set_input_const(freq[lo:lo + n])
        return b
    eng.mixdown_reserve(V, T)
    whole = make(0, V)
    parts = [make(k * per, per) for k in range(shards)]
    nrows = ml.mixdown_shard_rows(per)
    d_one = eng.alloc(4 * 64 * T)
    d_rows = [eng.alloc(4 * nrows * 64 * T) for _ in range(shards)]
    # (the CPU side of the same block, for the smaller banks: the oracle's voices through the oracle's tree - the in-kernel sum is not
    # only compared with other device code)
    cpu_state = oracle.chain_clear(chain, V) if V <= 8192 else None
    cpu_coeffs = np.ascontiguousarray(np.concatenate([np.stack([few[i][np.arange(V) % 16] for i in range(3)]), np.full((1, V), 0.25, np.float32)], 0)) if V <= 8192 else None
    for launch in range(2):
        whole.process_mixdown(T, d_one)
        for b, d in zip(parts, d_rows):
            b.process_mixdown_shard(T, d)
        rows = np.concatenate([d.download(np.float32, nrows * 64 * T).reshape(nrows, 64 * T) for d in d_rows], 0)
        assert_bits_equal(ml.mixdown_finish(rows), d_one.download(np.float32, 64 * T), True, f"{shards} shards of {per} voices, launch {launch}")
        if cpu_state is not None:
            voices = oracle.chain_process(chain, T, cpu_coeffs, cpu_state, None, freq, n_threads=4)
            assert_bits_equal(d_one.download(np.float32, 64 * T), oracle.mixdown(voices), True, f"bank_process_mixdown vs the oracle's voices and tree, launch {launch}")
            shard_rows = np.concatenate([oracle.mixdown_shard(voices[k * per:(k + 1) * per]) for k in range(shards)], 0)
            assert_bits_equal(rows, shard_rows, True, "the shards' rows vs the oracle's")
    # a signal in memory
    sig = lcg_noise(np.arange(V, dtype=np.uint32) + 5, 64 * T)
    gains = rng.uniform(-1, 1, V).astype(np.float32)
    d_sig, d_g = eng.to_device(sig), eng.to_device(gains)
    eng.mixdown(d_sig, Layout.VOICE_MAJOR, V, T, d_one, d_g)
    rows = []
    for k in range(shards):
        d_part, d_gp = eng.to_device(np.ascontiguousarray(sig[k * per:(k + 1) * per])), eng.to_device(np.ascontiguousarray(gains[k * per:(k + 1) * per]))
        eng.mixdown_shard(d_part, Layout.VOICE_MAJOR, per, T, d_rows[k], d_gp)
        rows.append(d_rows[k].download(np.float32, nrows * 64 * T).reshape(nrows, 64 * T))
    rows = np.concatenate(rows, 0)
    assert_bits_equal(ml.mixdown_finish(rows), d_one.download(np.float32, 64 * T), True, "mixdown_shard of a signal")
    assert_bits_equal(ml.mixdown_finish(rows), oracle.mixdown(sig, gains), True, "... and the oracle's unsplit tree")
    # a graph output in shard form
    if V <= 8192:
        desc = [dict(name="x", type="input"), dict(name="g", type="const", value=0.5), dict(name="y", type="op", kind=Op.MULTIPLY, inputs=["x", "g"])]
        gw = ml.Graph(eng, V, desc, ["y"], compile_now=False)
        gw.set_output_mixdown(0)
        gw.compile()
        gw.reserve_mixdown(T)
        gw.process(T, [d_sig], [d_one], Layout.VOICE_MAJOR)
        rows = []
        for k in range(shards):
            gp = ml.Graph(eng, per, desc, ["y"], compile_now=False)
            gp.set_output_mixdown(0, "shard")
            gp.compile()
            gp.reserve_mixdown(T)
            gp.process(T, [eng.to_device(np.ascontiguousarray(sig[k * per:(k + 1) * per]))], [d_rows[k]], Layout.VOICE_MAJOR)
            rows.append(d_rows[k].download(np.float32, nrows * 64 * T).reshape(nrows, 64 * T))
            gp.close()
        assert_bits_equal(ml.mixdown_finish(np.concatenate(rows, 0)), d_one.download(np.float32, 64 * T), True, "graph outputs in shard form")
        gw.close()
    with pytest.raises(ml.MlgpuError):
        eng.bank(chain, 100).process_mixdown_shard(T, d_rows[0])     # not whole first-stage groups: no exact hand-over


@pytest.mark.gpu
def test_bank_process_mixdown_refusals(eng):
    import madronalib_amd as ml
    d_out = eng.alloc(4 * 64)
    eng.mixdown_reserve(4096, 1)
    for procs, V in (((Proc.LOPASS,), 128),                                  # no ahead-of-time summing form of these kernels:
                     ((Proc.SAW_GEN, Proc.LOPASS, Proc.HIPASS, Proc.GAIN), 128)):   # prepare_mixdown at setup
        b = eng.bank(list(procs), V)
        with pytest.raises(ml.MlgpuError) as ei:
            b.process_mixdown(1, d_out)
        assert ei.value.status == ml.Status.ERR_UNSUPPORTED and "prepare_mixdown" in str(ei.value)
    small = ml.Engine(0)
    b = small.bank([Proc.SAW_GEN, Proc.BANDPASS, Proc.GAIN], 8192)
    with pytest.raises(ml.MlgpuError) as ei:
        b.process_mixdown(1, small.alloc(4 * 64))        # nothing reserved: a process call refuses instead of allocating
    assert ei.value.status == ml.Status.ERR_INVALID and "mixdown_reserve" in str(ei.value)
    small.close()


@pytest.mark.gpu
@pytest.mark.parametrize("chain,signal,strict", [((Proc.LOPASS,), True, False), ((Proc.SAW_GEN, Proc.LOPASS, Proc.HIPASS, Proc.GAIN), False, False),
                                                 ((Proc.LOPASS,) * 4, True, False), ((Proc.NOISE_GEN,) + (Proc.LOPASS,) * 8, False, False),
                                                 ((Proc.SAW_GEN, Proc.BANDPASS, Proc.GAIN), False, True)])
def test_bank_prepare_mixdown_any_chain(oracle, chain, signal, strict):
    """mlgpu_bank_prepare_mixdown: the summing form generated at setup for chains that have none ahead of time - a single processor, a
    chain hiprtc fuses, stage-skewed SVF cascades (whose summing form is the plain chain on the same state words), and every chain of
    an engine in strict SVF mode - then mlgpu_bank_process_mixdown against the two calls it replaces, bits and state, three launches."""
    import madronalib_amd as ml
    eng = ml.Engine(0)
    if strict:
        eng.set_strict_svf(True)
    V, T, launches = 1000, 2, 3
    rng = np.random.default_rng(7)
    x = lcg_noise(np.arange(V, dtype=np.uint32) + 11, 64 * T * launches) if chain[0] in (Proc.LOPASS,) else \
        (55.0 * 2.0 ** (5.0 * rng.random((V, 1))) / 48000.0 * np.ones((1, 64 * T * launches))).astype(np.float32)
    eng.mixdown_reserve(V, T)
    banks = [eng.bank(list(chain), V) for _ in range(2)]
    for b in banks:
        b.clear()
        for p, kind in enumerate(chain):
            if kind in (Proc.LOPASS, Proc.HIPASS, Proc.BANDPASS):
                name = {Proc.LOPASS: "lopass", Proc.HIPASS: "hipass", Proc.BANDPASS: "bandpass"}[kind]
                few = np.stack([oracle.make_coeffs(name, 0.02 + 0.3 * j / 16, 0.6) for j in range(16)], 1)
                b.set_coeffs(p, [np.ascontiguousarray(few[i][np.arange(V) % 16]) for i in range(few.shape[0])])
            elif kind == Proc.GAIN:
                b.set_coeff(p, 0, 0.25)
        if not signal and chain[0] != Proc.NOISE_GEN:
            b.set_input_const(np.ascontiguousarray(x[:, 0]))
        if chain[0] == Proc.NOISE_GEN:
            b.set_state(0, 0, np.arange(1, V + 1, dtype=np.uint32))
    banks[1].prepare_mixdown()
    banks[1].prepare_mixdown()      # (idempotent)
    d_voices = eng.alloc(4 * V * T * 64)
    d_two, d_one = eng.alloc(4 * T * 64), eng.alloc(4 * T * 64)
    for k in range(launches):
        d_in = eng.to_device(np.ascontiguousarray(x[:, k * 64 * T:(k + 1) * 64 * T])) if signal else None
        banks[0].process(T, d_voices, Layout.QUAD, d_in, Layout.VOICE_MAJOR)
        eng.mixdown(d_voices, Layout.QUAD, V, T, d_two)
        banks[1].process_mixdown(T, d_one, d_in, Layout.VOICE_MAJOR)
        two, one = d_two.download(np.float32, 64 * T), d_one.download(np.float32, 64 * T)
        assert_bits_equal(one, two, True, f"launch {k}")
        assert np.isfinite(two).all() and np.abs(two).max() > 1e-6
    for p in range(len(chain)):
        for i in range(banks[0].num_state(p)):
            assert (banks[0].get_state(p, i) == banks[1].get_state(p, i)).all(), (p, i)
    eng.close()
