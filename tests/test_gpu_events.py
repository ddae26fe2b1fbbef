"""mlgpu_events (EventsToSignals: host event routing + device signal generation) against the reference's own
EventsToSignals class (oracle/_ref/libdropin_ref.so: e2s_ref_run) on scripted performances: bit-exact on all 8 rows."""
import ctypes
import os

import numpy as np
import pytest

from inputs import assert_bits_equal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NOTE_ON, NOTE_OFF, SUSTAIN, CTRL, BEND, NOTE_PRESS, CHAN_PRESS = 1, 4, 5, 6, 7, 8, 9
ROW_NAMES = ("pitch", "gate", "vox", "z", "x", "y", "mod", "time")


class RefEvent(ctypes.Structure):
    _fields_ = [("type", ctypes.c_uint8), ("channel", ctypes.c_uint8), ("sourceIdx", ctypes.c_uint16), ("time", ctypes.c_int32),
                ("value1", ctypes.c_float), ("value2", ctypes.c_float)]


def ref_run(cfg, events, block_frames, n_blocks):
    so = os.path.join(ROOT, "oracle", "_ref", "libdropin_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libdropin_ref.so not available here")
    L = ctypes.CDLL(so)
    L.e2s_ref_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                              ctypes.c_float, ctypes.c_int, ctypes.POINTER(RefEvent), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                              ctypes.POINTER(ctypes.c_float)]
    P = cfg["polyphony"]
    out = np.zeros((8, P, n_blocks * block_frames), np.float32)
    arr = (RefEvent * max(1, len(events)))(*[RefEvent(*e) for e in events])
    assert L.e2s_ref_run(P, int(cfg.get("mpe", 0)), int(cfg.get("unison", 0)), cfg.get("sr", 48000.0), cfg.get("glide", 0.0), cfg.get("drift", 0.0),
                         cfg.get("bend", 7.0), cfg.get("mpe_bend", 24.0), cfg.get("mod_cc", 16), arr, len(events), block_frames, n_blocks,
                         out.ctypes.data_as(ctypes.POINTER(ctypes.c_float))) == 0
    return out


def ref_run_controllers(cfg, events, block_frames, n_blocks, numbers):
    """(voice rows [8][P][frames], controller signals [len(numbers)][frames]) of the reference's EventsToSignals."""
    so = os.path.join(ROOT, "oracle", "_ref", "libdropin_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libdropin_ref.so not available here")
    L = ctypes.CDLL(so)
    fp = ctypes.POINTER(ctypes.c_float)
    L.e2s_ref_run_controllers.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                          ctypes.c_float, ctypes.c_int, ctypes.POINTER(RefEvent), ctypes.c_int, ctypes.c_int, ctypes.c_int, fp,
                                          ctypes.POINTER(ctypes.c_int), ctypes.c_int, fp]
    P = cfg["polyphony"]
    out = np.zeros((8, P, n_blocks * block_frames), np.float32)
    ctl = np.zeros((len(numbers), n_blocks * block_frames), np.float32)
    arr = (RefEvent * max(1, len(events)))(*[RefEvent(*e) for e in events])
    nums = (ctypes.c_int * len(numbers))(*numbers)
    assert L.e2s_ref_run_controllers(P, int(cfg.get("mpe", 0)), int(cfg.get("unison", 0)), cfg.get("sr", 48000.0), cfg.get("glide", 0.0), cfg.get("drift", 0.0),
                                     cfg.get("bend", 7.0), cfg.get("mpe_bend", 24.0), cfg.get("mod_cc", 16), arr, len(events), block_frames, n_blocks,
                                     out.ctypes.data_as(fp), nums, len(numbers), ctl.ctypes.data_as(fp)) == 0
    return out, ctl


def gpu_run(eng, cfg, per_instrument_events, block_frames, n_blocks, vectors_per_launch, rows=None, watch=None, watch_from_block=0):
    """watch: controller numbers; then returns (rows, controller signals [len(watch)][N][frames]; zeros before watch_from_block)."""
    """per_instrument_events: one event list per instrument. Returns [8][N*P][frames] (rows not wanted: zeros)."""
    import madronalib_amd as ml
    N, P = len(per_instrument_events), cfg["polyphony"]
    ev = ml.Events(eng, N, P, cfg.get("sr", 48000.0))
    ev.configure(mpe=cfg.get("mpe", 0), unison=cfg.get("unison", 0), mod_cc=cfg.get("mod_cc", 16), pitch_bend=cfg.get("bend", 7.0),
                 mpe_pitch_bend=cfg.get("mpe_bend", 24.0), glide_seconds=cfg.get("glide", 0.0), drift=cfg.get("drift", 0.0))
    if rows is not None:
        ev.set_wanted_rows(rows)
    outs, ctl = [], []
    for b in range(n_blocks):
        if watch is not None and b == watch_from_block:
            ev.watch_controllers(watch, vectors_per_launch)
        start = b * block_frames
        batch_i, batch_e = [], []
        for i, evs in enumerate(per_instrument_events):
            for e in evs:
                if start <= e[3] < start + block_frames:
                    if b % 2:      # odd blocks: one call per event; even blocks: the whole block in one call
                        ev.add_event(i, ml.Event(e[0], e[1], e[2], e[3] - start, e[4], e[5]))
                    else:
                        batch_i.append(i)
                        batch_e.append(ml.Event(e[0], e[1], e[2], e[3] - start, e[4], e[5]))
        ev.add_events(batch_i, batch_e)
        vecs = block_frames // 64
        done = 0
        while done < vecs:       # a block may be processed in several launches
            n = min(vectors_per_launch, vecs - done)
            outs.append(ev.process_host(n, done * 64))
            if watch is not None:
                ctl.append(ev.controllers_host(n) if b >= watch_from_block else np.zeros((len(watch), N, 64 * n), np.float32))
            done += n
        ev.clear_events()
    if watch is not None:
        return np.concatenate(outs, 2), np.concatenate(ctl, 2)
    return np.concatenate(outs, 2)


def performance(kind, seed, frames, polyphony):
    """A scripted event list: (type, channel, sourceIdx, time, value1, value2)."""
    rng = np.random.default_rng(seed)
    evs, held, t = [], [], int(rng.integers(0, 200))
    mpe = kind == "mpe"
    while t < frames - 10:
        r = rng.random()
        chan = int(rng.integers(2, 2 + polyphony + 2)) if mpe else 1
        if r < 0.35 or not held:
            key = int(rng.integers(30, 90))
            evs.append((NOTE_ON, chan, key, t, float(np.float32((key - 60) / 12.0)), float(np.float32(rng.uniform(0.1, 1.0)))))
            held.append((chan, key))
        elif r < 0.6:
            c, k = held.pop(int(rng.integers(0, len(held))))
            evs.append((NOTE_OFF, c, k, t, 0.0, 0.0))
        elif r < 0.7:
            evs.append((BEND, chan if rng.random() < 0.7 else 1, 0, t, float(np.float32(rng.uniform(-1, 1))), 0.0))
        elif r < 0.8:
            cc = int(rng.choice([16, 73, 74, 1, 128]))
            evs.append((CTRL, chan, cc, t, float(np.float32(rng.random())), 0.0))
        elif r < 0.87:
            if mpe:
                evs.append((CHAN_PRESS, chan if rng.random() < 0.8 else 1, 0, t, float(np.float32(rng.random())), 0.0))
            elif held and rng.random() < 0.5:
                evs.append((NOTE_PRESS, 1, held[-1][1], t, float(np.float32(rng.random())), 0.0))
            else:
                evs.append((CHAN_PRESS, 1, 0, t, float(np.float32(rng.random())), 0.0))
        elif r < 0.93 and kind == "sustain":
            evs.append((SUSTAIN, 1, 0, t, float(rng.integers(0, 2)), 0.0))
        elif r < 0.96:
            evs.append((CTRL, 1, 123, t, 0.0, 0.0))        # all notes off
            held = []
        t += int(rng.integers(1, 260)) if rng.random() < 0.9 else 0   # now and then two events on the same frame
    return evs


@pytest.fixture(scope="module")
def eng():
    import madronalib_amd as ml
    e = ml.Engine(0)
    yield e
    e.close()


SCENARIOS = {
    "midi_poly4": dict(polyphony=4, glide=0.01, drift=0.5),
    "midi_steal2": dict(polyphony=2, glide=0.003, drift=0.0),
    "midi_poly16": dict(polyphony=16, glide=0.02, drift=1.0, bend=2.0),
    "unison3": dict(polyphony=3, unison=1, glide=0.05, drift=0.2),
    "sustain": dict(polyphony=4, glide=0.0, drift=0.0),
    "mpe5": dict(polyphony=5, mpe=1, glide=0.01, drift=0.3, mpe_bend=48.0),
    "sr44k": dict(polyphony=3, sr=44100.0, glide=0.015, drift=0.7, mod_cc=1),
    "sr8k": dict(polyphony=3, sr=8000.0, glide=0.05, drift=0.3),
    "sr192k": dict(polyphony=4, sr=192000.0, glide=0.002, drift=0.9),
    "poly1": dict(polyphony=1, glide=0.01, drift=0.4),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(SCENARIOS))
def test_events_to_signals_matches_reference(eng, name):
    cfg = SCENARIOS[name]
    block, n_blocks = 512, 12
    kind = "mpe" if cfg.get("mpe") else ("sustain" if name == "sustain" else "midi")
    instruments = [performance(kind, 100 * k + len(name), block * n_blocks, cfg["polyphony"]) for k in range(5)]
    got = gpu_run(eng, cfg, instruments, block, n_blocks, vectors_per_launch=3)
    P = cfg["polyphony"]
    for k, evs in enumerate(instruments):
        want = ref_run(cfg, evs, block, n_blocks)
        for r in range(8):
            assert_bits_equal(got[r, k * P:(k + 1) * P], want[r], True, f"{name}: instrument {k} row {ROW_NAMES[r]}")
    assert np.abs(got[1]).max() > 0 and np.abs(got[0]).max() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("name,rows", [("midi_poly4", None), ("midi_poly16", [0, 1]), ("unison3", None), ("sustain", [0, 1]), ("sr8k", None),
                                       ("sr192k", [0, 1, 3]), ("mpe5", None), ("poly1", [0, 1])])
def test_dense_performances_in_launches_of_eight_vectors(eng, name, rows):
    """The scripted performances (an event every ~130 frames: notes, bends, controllers, pressure, sustain, all-notes-off) in launches of
    8 DSPVectors: blocks of 4 vectors take e2s_kernel's block path whenever they happen to be quiet - between moving pitch glides,
    gliding bends and controllers - and hand over to the general loop and back inside one launch; with only pitch and gate wanted the
    two-row instance of the kernel runs. (MPE keeps the general loop throughout.) Against the reference's class, bit for bit."""
    cfg = SCENARIOS[name]
    block, n_blocks = 512, 12
    kind = "mpe" if cfg.get("mpe") else ("sustain" if name == "sustain" else "midi")
    instruments = [performance(kind, 7000 + 100 * k + len(name), block * n_blocks, cfg["polyphony"]) for k in range(4)]
    got = gpu_run(eng, cfg, instruments, block, n_blocks, vectors_per_launch=8, rows=rows)
    P = cfg["polyphony"]
    for k, evs in enumerate(instruments):
        want = ref_run(cfg, evs, block, n_blocks)
        for r in (range(8) if rows is None else rows):
            assert_bits_equal(got[r, k * P:(k + 1) * P], want[r], True, f"{name}: instrument {k} row {ROW_NAMES[r]}")


def sparse_performance(seed, frames, gap):
    """A few notes and bends with long silences between them: most launches see no event at all."""
    rng = np.random.default_rng(seed)
    evs, held, t = [], [], int(rng.integers(0, gap))
    while t < frames - 10:
        r = rng.random()
        if r < 0.45 or not held:
            key = int(rng.integers(30, 90))
            evs.append((NOTE_ON, 1, key, t, float(np.float32((key - 60) / 12.0)), float(np.float32(rng.uniform(0.1, 1.0)))))
            held.append(key)
        elif r < 0.8:
            evs.append((NOTE_OFF, 1, held.pop(int(rng.integers(0, len(held)))), t, 0.0, 0.0))
        elif r < 0.9:
            evs.append((BEND, 1, 0, t, float(np.float32(rng.uniform(-1, 1))), 0.0))
        else:
            evs.append((CTRL, 1, 16, t, float(np.float32(rng.random())), 0.0))
        t += int(rng.integers(gap // 4, gap))
    return evs


@pytest.mark.gpu
@pytest.mark.parametrize("sr,vectors_per_launch,rows", [(2000.0, 16, None), (1000.0, 9, [0, 1]), (3000.0, 4, [0, 1, 7]), (2000.0, 13, [0])])
def test_quiet_blocks_through_drift_glide_changes(eng, sr, vectors_per_launch, rows):
    """Launches in which nothing happens take e2s_kernel's block path: several DSPVectors of the pitch row per fetch of the drift
    glide's slots. At a low sample rate the drift glide (8 s per glide, a new target every 8..16 s) starts, continues and ends many
    times inside such blocks; note events, bends (whose glide keeps moving for a while) and pitch glides in between send single
    vectors and whole launches down the vector-by-vector path. 2304 DSPVectors, all rows against the reference's class."""
    cfg = dict(polyphony=4, sr=sr, glide=0.05, drift=0.8)
    block, n_blocks = 64 * 144, 16
    instruments = [sparse_performance(300 + k, block * n_blocks, 20000) for k in range(3)] + [[]]
    got = gpu_run(eng, cfg, instruments, block, n_blocks, vectors_per_launch=vectors_per_launch, rows=rows)
    P = cfg["polyphony"]
    for k, evs in enumerate(instruments):
        want = ref_run(cfg, evs, block, n_blocks)
        for r in (range(8) if rows is None else rows):
            assert_bits_equal(got[r, k * P:(k + 1) * P], want[r], True, f"sr {sr}: instrument {k} row {ROW_NAMES[r]}")
    pitch = got[0, :P]
    assert len(np.unique(np.round(np.diff(pitch[0, ::64]), 9))) > 4     # the drift really moves through several glides


@pytest.mark.gpu
@pytest.mark.parametrize("name,rows", [("midi_poly4", [0, 1]), ("mpe5", [0, 1, 3, 6]), ("midi_poly16", [1, 2, 4, 5, 7])])
def test_wanted_rows_only(eng, name, rows):
    """mlgpu_events_set_wanted_rows: the rows asked for are the reference's; the others are neither computed nor written,
    and passing a buffer for one of them is an error."""
    import madronalib_amd as ml
    cfg = SCENARIOS[name]
    block, n_blocks, P = 512, 6, cfg["polyphony"]
    instruments = [performance("mpe" if cfg.get("mpe") else "midi", 31 * k + 5, block * n_blocks, P) for k in range(3)]
    got = gpu_run(eng, cfg, instruments, block, n_blocks, vectors_per_launch=4, rows=rows)
    for k, evs in enumerate(instruments):
        want = ref_run(cfg, evs, block, n_blocks)
        for r in range(8):
            if r in rows:
                assert_bits_equal(got[r, k * P:(k + 1) * P], want[r], True, f"{name}: instrument {k} row {ROW_NAMES[r]}")
            else:
                assert (got[r] == 0).all()
    ev = ml.Events(eng, 2, 4)
    ev.set_wanted_rows([0, 1])
    bufs = [eng.alloc(4 * 8 * 64) for _ in range(8)]
    with pytest.raises(ml.MlgpuError):
        ev.process(1, 0, bufs)


@pytest.mark.gpu
def test_silent_instrument_and_many_instruments(eng):
    """An instrument that never received an event outputs zeros (and its vox row); 3000 instruments in one launch."""
    cfg = dict(polyphony=4, glide=0.01, drift=0.4)
    evs = performance("midi", 7, 512 * 4, 4)
    N = 3000
    instruments = [evs if (k % 7 == 0) else [] for k in range(N)]
    got = gpu_run(eng, cfg, instruments, 512, 4, vectors_per_launch=8)
    want = ref_run(cfg, evs, 512, 4)
    for k in (0, 7, 2996):
        for r in range(8):
            assert_bits_equal(got[r, k * 4:(k + 1) * 4], want[r], True, f"instrument {k} row {ROW_NAMES[r]}")
    silent = got[:, 4:8]
    assert (silent[[0, 1, 3, 4, 5, 6, 7]] == 0).all()
    assert (silent[2] == np.arange(4, dtype=np.float32)[:, None]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["midi_poly4", "mpe5", "unison3"])
def test_events_with_hostile_values(eng, name):
    """The same performances with a third of the event VALUES replaced by what no controller sends: NaN and infinite pitches,
    velocities that are zero, negative, NaN or 3e38, bends and pressures of 1e30, denormals, -0. (Indices stay legal: the
    reference indexes its key and controller tables with them unchecked.) All 8 rows, bit for bit against the reference's own
    class; any NaN equals any NaN."""
    cfg = SCENARIOS[name]
    block, n_blocks = 512, 10
    kind = "mpe" if cfg.get("mpe") else "midi"
    odd = [float("nan"), float("inf"), float("-inf"), 0.0, -0.0, -0.5, 3.0e38, -3.0e38, 1.0e30, 1.0e-40, -1.0e-40, 2.0, 1.0e-38]
    instruments = []
    for k in range(4):
        rng = np.random.default_rng(900 + k)
        evs = []
        for e in performance(kind, 300 * k + len(name), block * n_blocks, cfg["polyphony"]):
            e = list(e)
            if e[0] != CTRL or e[2] not in (123,):      # leave "all notes off" alone
                if rng.random() < 0.33:
                    e[4] = odd[int(rng.integers(0, len(odd)))]
                if rng.random() < 0.33 and e[0] == NOTE_ON:
                    e[5] = odd[int(rng.integers(0, len(odd)))]
            evs.append(tuple(e))
        instruments.append(evs)
    watch = [16, 73, 74, 1, 128]
    got, ctl = gpu_run(eng, cfg, instruments, block, n_blocks, vectors_per_launch=4, watch=watch)
    P = cfg["polyphony"]
    for k, evs in enumerate(instruments):
        want, want_ctl = ref_run_controllers(cfg, evs, block, n_blocks, watch)
        for r in range(8):
            assert_bits_equal(got[r, k * P:(k + 1) * P], want[r], True, f"hostile {name}: instrument {k} row {ROW_NAMES[r]}")
        for c, num in enumerate(watch):     # the smoothed controllers glide towards (and from) infinities and NaNs too
            assert_bits_equal(ctl[c, k], want_ctl[c], True, f"hostile {name}: instrument {k} controller {num}")


def _two_kernel_and_fused(eng, cfg, instruments, block, n_blocks, vectors_per_launch, switch_at=None, voice_sum=False):
    """The config-5 voice driven by EventsToSignals two ways, from fresh objects: (a) e2s_kernel writes pitch and gate, the voice
    graph reads them; (b) the voice graph computes the two rows itself (event_row nodes). switch_at: from that block on, (b) goes
    back to the two-kernel form - the two forms share the events object's state. Returns (audio_a, audio_b) [V][frames].
    voice_sum: the instruments' audio instead, [N][frames] - (a) mlgpu_mixdown_groups over the voices, (b) the sum made inside the
    fused voice kernel (mlgpu_graph_set_output_group_sum)."""
    import madronalib_amd as ml
    from madronalib_amd import patches
    from madronalib_amd.constants import Layout
    from madronalib_amd.sharding import cfg5_voice_params
    N, P = len(instruments), cfg["polyphony"]
    V = N * P
    outs = []
    for fused in (False, True):
        ev = ml.Events(eng, N, P, cfg.get("sr", 48000.0))
        ev.configure(mpe=0, unison=cfg.get("unison", 0), mod_cc=cfg.get("mod_cc", 16), pitch_bend=cfg.get("bend", 7.0),
                     glide_seconds=cfg.get("glide", 0.0), drift=cfg.get("drift", 0.0))
        ev.set_wanted_rows([0, 1])
        params, coeffs, seeds = cfg5_voice_params(0, V, V, ml)
        graphs = {}
        for form in ((True, False) if fused else (False,)):
            desc, outn = patches.synth16(pitch_input=True, event_rows=form)
            g = ml.Graph(eng, V, desc, outn, output_groups={0: P} if (voice_sum and form) else None)
            g.clear()
            for k, v in params.items():
                if k != "pitch":
                    g.set_param(k, v if np.ndim(v) else float(v))
            for k, c in coeffs.items():
                g.set_coeffs(k, [np.ascontiguousarray(r) for r in c])
            g.set_state("noise", 0, seeds)
            if form:
                assert "mlev::CtlVoice" in g.source and not g.inputs
                g.bind_events(ev)
            graphs[form] = g
        n = V * vectors_per_launch * 64
        rows = [eng.alloc(4 * n), eng.alloc(4 * n)]
        d_out = eng.alloc(4 * n)
        d_mix = eng.alloc(4 * n // P)
        chunks = []
        for b in range(n_blocks):
            start = b * block
            bi, be = [], []
            for i, evs in enumerate(instruments):
                for e in evs:
                    if start <= e[3] < start + block:
                        bi.append(i)
                        be.append(ml.Event(e[0], e[1], e[2], e[3] - start, e[4], e[5]))
            ev.add_events(bi, be)
            use_fused = fused and (switch_at is None or b < switch_at)
            if fused and not use_fused and b == switch_at:   # the plain graph takes over the voice state of the fused one
                for nd in [d["name"] for d in patches.synth16(pitch_input=True)[0] if d["type"] == "proc"]:
                    for i in range(graphs[False].num_state(nd)):
                        graphs[False].set_state(nd, i, graphs[True].get_state(nd, i))
            done = 0
            while done < block // 64:
                T = min(vectors_per_launch, block // 64 - done)
                if use_fused:
                    graphs[True].process_events(T, done * 64, [], [d_mix if voice_sum else d_out], out_layout=Layout.VOICE_MAJOR)
                else:
                    g = graphs[False]
                    ev.process(T, done * 64, [rows[0], rows[1]] + [None] * 6, Layout.QUAD)
                    g.process(T, [rows[1] if nm == "gate" else rows[0] for nm in g.inputs], [d_out], out_layout=Layout.VOICE_MAJOR)
                    if voice_sum:
                        eng.mixdown_groups(d_out, Layout.VOICE_MAJOR, N, P, T, d_mix, Layout.VOICE_MAJOR)
                if voice_sum:
                    chunks.append(d_mix.download(np.float32, N * T * 64).reshape(N, T * 64).copy())
                else:
                    chunks.append(d_out.download(np.float32, V * T * 64).reshape(V, T * 64).copy())
                done += T
            ev.clear_events()
        outs.append(np.concatenate(chunks, 1))
        for g in graphs.values():
            g.close()
    return outs


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["midi_poly4", "midi_steal2", "midi_poly16", "unison3", "sustain", "sr44k", "sr8k", "sr192k", "poly1"])
def test_event_rows_inside_the_voice_graph(eng, name):
    """EventsToSignals' pitch and gate rows as source nodes of the voice graph (mlgpu_graph_add_event_row / bind_events /
    process_events): the same audio, bit for bit, as e2s_kernel writing the rows and the graph reading them - which is itself
    bit-exact against the reference's class (above) - on the scripted performances, with several launches per block."""
    cfg = SCENARIOS[name]
    block, n_blocks = 512, 8
    kind = "sustain" if name == "sustain" else "midi"
    instruments = [performance(kind, 100 * k + len(name), block * n_blocks, cfg["polyphony"]) for k in range(6)]
    a, b = _two_kernel_and_fused(eng, cfg, instruments, block, n_blocks, vectors_per_launch=3)
    assert_bits_equal(b, a, True, f"{name}: fused event rows vs two kernels")
    assert np.abs(a).max() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["midi_poly4", "midi_steal2", "midi_poly16"])
def test_instrument_bank_in_one_voice_kernel(eng, name):
    """The whole instrument bank as bench.py --workload synthfused runs it - control records -> the fused voice kernel, which also
    adds up every instrument's voices (groups of 16 through LDS, smaller ones with lane shifts) - against the three-kernel form:
    e2s_kernel rows -> voice graph -> mlgpu_mixdown_groups. 37 instruments (a partial last wavefront), drift and portamento on."""
    cfg = SCENARIOS[name]
    block, n_blocks = 512, 6
    instruments = [performance("midi", 31 * k + len(name), block * n_blocks, cfg["polyphony"]) for k in range(37)]
    a, b = _two_kernel_and_fused(eng, cfg, instruments, block, n_blocks, vectors_per_launch=8, voice_sum=True)
    assert_bits_equal(b, a, True, f"{name}: instrument audio, one kernel vs three")
    assert np.abs(a).max() > 0


@pytest.mark.gpu
def test_event_rows_through_a_whole_drift_cycle(eng):
    """The drift glide is the one state machine the voice kernel owns (mlev::CtlVoice): 8 s per glide, a new target every 8-16 s. At
    8 kHz that is 1 000 DSPVectors per glide: 2 600 vectors take every voice through the ramp that starts a glide, its middle, its
    end, the rest at the target and the start of the next one from a resting mCurrVec - with note events, portamento and pitch bends
    on top. Fused against the two-kernel form (itself bit-exact against the reference's class on shorter runs)."""
    cfg = SCENARIOS["sr8k"]
    block, n_blocks = 512, 325
    instruments = [performance("midi", 900 + k, block * n_blocks, cfg["polyphony"]) for k in range(3)]
    instruments = [[e for e in evs if e[3] % 7 == 0] for evs in instruments]      # a sparser performance: the long stretches are the point
    a, b = _two_kernel_and_fused(eng, cfg, instruments, block, n_blocks, vectors_per_launch=8)
    assert_bits_equal(b, a, True, "fused event rows vs two kernels over 2 600 vectors")
    assert np.abs(a).max() > 0


@pytest.mark.gpu
def test_event_rows_and_events_kernel_share_their_state(eng):
    """A block may be processed by either form: half way through, the fused graph hands over to e2s_kernel + plain graph."""
    cfg = SCENARIOS["midi_poly4"]
    block, n_blocks = 512, 8
    instruments = [performance("midi", 40 * k + 7, block * n_blocks, cfg["polyphony"]) for k in range(5)]
    a, b = _two_kernel_and_fused(eng, cfg, instruments, block, n_blocks, vectors_per_launch=4, switch_at=4)
    assert_bits_equal(b, a, True, "fused for four blocks, then two kernels")


@pytest.mark.gpu
def test_event_rows_error_paths(eng):
    """What the event-row API refuses: rows other than pitch / gate, a row twice, running without an events object or through
    the plain process call, an MPE object, an object with another number of voices, output group sums of other sizes."""
    import madronalib_amd as ml
    from madronalib_amd.constants import Op
    V = 64
    g = ml.Graph(eng, V)
    with pytest.raises(ml.MlgpuError):
        g.add("time", "event_row", 7)
    g.add("pitch", "event_row", 0)
    with pytest.raises(ml.MlgpuError):
        g.add("pitch2", "event_row", 0)
    g.add("gate", "event_row", 1)
    g.add("y", "op", Op.MULTIPLY, ["pitch", "gate"])
    g.add_output("y")
    with pytest.raises(ml.MlgpuError):
        g.set_output_group_sum(0, 3)
    with pytest.raises(ml.MlgpuError):
        g.set_output_group_sum(1, 4)          # no such output
    g.compile()
    d_out = eng.alloc(4 * V * 64)
    with pytest.raises(ml.MlgpuError):
        g.process_events(1, 0, [], [d_out])   # nothing bound
    mpe = ml.Events(eng, V // 4, 4, 48000.0)
    mpe.configure(mpe=1)
    with pytest.raises(ml.MlgpuError):
        g.bind_events(mpe)
    other = ml.Events(eng, V // 4 + 1, 4, 48000.0)
    with pytest.raises(ml.MlgpuError):
        g.bind_events(other)
    ev = ml.Events(eng, V // 4, 4, 48000.0)
    g.bind_events(ev)
    with pytest.raises(ml.MlgpuError):
        g.process(1, [], [d_out])             # a graph with event rows runs through process_events
    g.process_events(1, 0, [], [d_out])       # no events yet: the reference's processVector does nothing before the first event
    assert not np.any(d_out.download(np.float32, V * 64))
    plain = ml.Graph(eng, V, [dict(name="x", type="input"), dict(name="z", type="op", kind=Op.ADD, inputs=["x", "x"])], ["z"])
    with pytest.raises(ml.MlgpuError):
        plain.bind_events(ev)                 # no event rows in it
    for o in (g, plain):
        o.close()


@pytest.mark.gpu
def test_event_rows_full_size(eng):
    """The instrument bank at bench size - 16 384 instruments x 16 voices = 262 144 voices - with sparse note events: the voice graph
    with its event rows inside against e2s_kernel + the plain graph, every voice, bit for bit, over three launches."""
    import madronalib_amd as ml
    from madronalib_amd import patches
    from madronalib_amd.constants import Layout
    from madronalib_amd.sharding import cfg5_voice_params
    N, P, T, launches = 16384, 16, 2, 3
    V = N * P
    rng = np.random.default_rng(5)
    script = []
    for _ in range(launches):
        insts = rng.integers(0, N, N // 40)
        evs = []
        for i in insts:
            key = int(rng.integers(36, 84))
            kind = NOTE_ON if rng.random() < 0.7 else NOTE_OFF
            evs.append((int(i), ml.Event(kind, 1, key, int(rng.integers(0, 64 * T)), (key - 60) / 12.0, 0.8 if kind == NOTE_ON else 0.0)))
        script.append(evs)
    params, coeffs, seeds = cfg5_voice_params(0, V, V, ml)
    results = []
    for fused in (False, True):
        ev = ml.Events(eng, N, P, 48000.0)
        ev.configure(glide_seconds=0.01, drift=0.5)
        ev.set_wanted_rows([0, 1])
        desc, outn = patches.synth16(pitch_input=True, event_rows=fused)
        g = ml.Graph(eng, V, desc, outn)
        g.clear()
        for k, v in params.items():
            if k != "pitch":
                g.set_param(k, v if np.ndim(v) else float(v))
        for k, c in coeffs.items():
            g.set_coeffs(k, [np.ascontiguousarray(r) for r in c])
        g.set_state("noise", 0, seeds)
        if fused:
            g.bind_events(ev)
        n = V * T * 64
        rows = [eng.alloc(4 * n), eng.alloc(4 * n)]
        d_out = eng.alloc(4 * n)
        outs = []
        for evs in script:
            ev.add_events([i for i, _ in evs], [e for _, e in evs])
            if fused:
                g.process_events(T, 0, [], [d_out])
            else:
                ev.process(T, 0, [rows[0], rows[1]] + [None] * 6, Layout.QUAD)
                g.process(T, [rows[1] if nm == "gate" else rows[0] for nm in g.inputs], [d_out])
            ev.clear_events()
            outs.append(d_out.download(np.float32, n).copy())
        results.append(outs)
        g.close()
    for k in range(launches):
        assert_bits_equal(results[1][k], results[0][k], True, f"full-size instrument bank, launch {k}")
    assert np.abs(results[0][-1]).max() > 0


# ---- smoothed controller signals (AudioContext::getInputController) ------------------------------------------------------------

WATCHED = [16, 128, 74, 5, 1]      # 5 is never sent: its signal stays at zero


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["midi_poly4", "mpe5", "sr44k", "sr8k", "poly1"])
def test_controller_signals_match_reference(eng, name):
    """mlgpu_events_watch_controllers / controller_signal against EventsToSignals::getController(n).output of the reference's
    class on the scripted performances (controllers 1, 16, 73, 74, 128, MIDI channel pressure -> 128), several launches per block,
    one instrument that never receives an event (asleep: zeros); the voice rows are the same as without watching."""
    cfg = SCENARIOS[name]
    block, n_blocks = 512, 10
    kind = "mpe" if cfg.get("mpe") else "midi"
    instruments = [performance(kind, 100 * k + len(name), block * n_blocks, cfg["polyphony"]) for k in range(4)] + [[]]
    rows, ctl = gpu_run(eng, cfg, instruments, block, n_blocks, vectors_per_launch=3, watch=WATCHED)
    P = cfg["polyphony"]
    moved = 0
    for k, evs in enumerate(instruments):
        want_rows, want_ctl = ref_run_controllers(cfg, evs, block, n_blocks, WATCHED)
        for r in range(8):
            assert_bits_equal(rows[r, k * P:(k + 1) * P], want_rows[r], True, f"{name}: instrument {k} row {ROW_NAMES[r]}")
        for c, num in enumerate(WATCHED):
            assert_bits_equal(ctl[c, k], want_ctl[c], True, f"{name}: instrument {k} controller {num}")
        moved += int(np.abs(np.diff(want_ctl, axis=1)).max() > 0)
    assert moved >= 3 and not ctl[:, -1].any() and not ctl[3].any()


@pytest.mark.gpu
def test_controllers_watched_later_start_settled(eng):
    """A smoother that starts being watched after its controller last moved (more than the 20 ms glide ago) continues exactly
    like the reference's, which has been running all along."""
    cfg = SCENARIOS["midi_poly4"]
    block, n_blocks = 512, 8
    evs = [(NOTE_ON, 1, 60, 10, 0.0, 0.8), (CTRL, 1, 74, 100, 0.625, 0.0), (CTRL, 1, 16, 300, 0.25, 0.0),
           (CTRL, 1, 74, 5 * 512 + 77, 0.125, 0.0), (CHAN_PRESS, 1, 0, 6 * 512 + 3, 0.5, 0.0)]
    rows, ctl = gpu_run(eng, cfg, [evs], block, n_blocks, vectors_per_launch=8, watch=[74, 16, 128], watch_from_block=4)
    _, want = ref_run_controllers(cfg, evs, block, n_blocks, [74, 16, 128])
    assert_bits_equal(ctl[:, 0, 4 * block:], want[:, 4 * block:], True, "controllers watched from block 4 on")
    assert want[0, 4 * block] == np.float32(0.625) and want[0, -1] == np.float32(0.125) and want[2, -1] == np.float32(0.5)


@pytest.mark.gpu
def test_watch_controllers_error_paths(eng):
    import madronalib_amd as ml
    ev = ml.Events(eng, 3, 4)
    for bad in ([1, 1], [129], [-1], list(range(33))):
        with pytest.raises(ml.MlgpuError):
            ev.watch_controllers(bad, 4)
    with pytest.raises(ml.MlgpuError):
        ev.watch_controllers([1], 0)
    with pytest.raises(ml.MlgpuError):
        ev.controller_signal(0)
    ev.watch_controllers([7, 1], 2)
    with pytest.raises(ml.MlgpuError) as ei:
        ev.process_host(3)                       # more vectors than the controller signals were reserved for
    assert ei.value.status == ml.Status.ERR_RANGE
    ev.process_host(2)
    assert ev.controllers_host(2).shape == (2, 3, 128)
    ev.watch_controllers([], 0)                  # releases them
    ev.process_host(3)
    ev.close()


@pytest.mark.gpu
def test_controller_signal_as_a_group_input_of_the_voice_graph(eng):
    """What a process function that calls ctx->getInputController(n) becomes: the controller's signal - one row per instrument -
    read by every voice of that instrument (mlgpu_graph_set_input_group), here multiplied with the gate row computed in the same
    kernel (event rows) and, in a second graph, with the gate row e2s_kernel wrote. Both equal reference gate x reference
    controller signal (one multiply: bit-exact)."""
    import madronalib_amd as ml
    from madronalib_amd.constants import Layout, Op
    cfg = SCENARIOS["midi_poly4"]
    P, N, block, n_blocks, T = 4, 6, 512, 6, 4
    V = N * P
    instruments = [performance("midi", 31 * k + 5, block * n_blocks, P) for k in range(N)]
    want = np.zeros((V, block * n_blocks), np.float32)
    for k, evs in enumerate(instruments):
        rows, ctl = ref_run_controllers(cfg, evs, block, n_blocks, [74])
        want[k * P:(k + 1) * P] = rows[1] * ctl[0][None, :]
    for fused in (True, False):
        ev = ml.Events(eng, N, P)
        ev.configure(glide_seconds=cfg["glide"], drift=cfg["drift"])
        ev.set_wanted_rows([0, 1])
        ev.watch_controllers([74], T)
        desc = [dict(name="gate", type="event_row", kind=1) if fused else dict(name="gate", type="input"), dict(name="cc74", type="input"),
                dict(name="out", type="op", kind=Op.MULTIPLY, inputs=["gate", "cc74"])]
        g = ml.Graph(eng, V, desc, ["out"], input_groups={(0 if fused else 1): P})
        if fused:
            g.bind_events(ev)
        d_out, d_gate, d_pitch = eng.alloc(4 * V * T * 64), eng.alloc(4 * V * T * 64), eng.alloc(4 * V * T * 64)
        chunks = []
        for b in range(n_blocks):
            bi, be = [], []
            for i, evs in enumerate(instruments):
                for e in evs:
                    if b * block <= e[3] < (b + 1) * block:
                        bi.append(i)
                        be.append(ml.Event(e[0], e[1], e[2], e[3] - b * block, e[4], e[5]))
            ev.add_events(bi, be)
            for done in range(0, block // 64, T):
                if fused:
                    g.process_events(T, done * 64, [ev.controller_signal(0)], [d_out], out_layout=Layout.VOICE_MAJOR)
                else:
                    ev.process(T, done * 64, [d_pitch, d_gate] + [None] * 6, Layout.QUAD)
                    g.process(T, [d_gate, ev.controller_signal(0)], [d_out], out_layout=Layout.VOICE_MAJOR)
                chunks.append(d_out.download(np.float32, V * T * 64).reshape(V, T * 64).copy())
            ev.clear_events()
        assert_bits_equal(np.concatenate(chunks, 1), want, True, f"gate x controller 74, fused={fused}")
        g.close()
        ev.close()
    assert np.abs(want).max() > 0


@pytest.mark.gpu
def test_controller_signals_bank_size(eng):
    """4096 instruments x 4 controllers = 16 384 controller lanes (many wavefronts, records of many lanes in one upload): eight
    performances dealt round-robin, every instrument equal to the reference's run of its performance."""
    cfg = SCENARIOS["midi_poly4"]
    block, n_blocks, N = 512, 6, 4096
    base = [performance("midi", 40 + k, block * n_blocks, 4) for k in range(7)] + [[]]
    watch = [1, 16, 74, 128]
    _, ctl = gpu_run(eng, cfg, [base[i % 8] for i in range(N)], block, n_blocks, vectors_per_launch=8, watch=watch)
    for k in range(8):
        _, want = ref_run_controllers(cfg, base[k], block, n_blocks, watch)
        for c in range(len(watch)):
            mine = ctl[c, k::8]
            assert_bits_equal(mine[0], want[c], True, f"performance {k}, controller {watch[c]}")
            assert (mine.view(np.uint32) == mine[0].view(np.uint32)[None, :]).all(), f"instruments of performance {k} differ"


@pytest.mark.gpu
def test_controller_events_on_the_same_frame(eng):
    """addEvent inserts with lower_bound (MLEventsToSignals.cpp:372): of two controller events on the same frame the one added LATER is
    processed FIRST, so the value added first is what the controller ends the vector with."""
    cfg = SCENARIOS["midi_poly4"]
    evs = [(NOTE_ON, 1, 60, 0, 0.0, 0.5), (CTRL, 1, 74, 70, 0.25, 0.0), (CTRL, 1, 74, 70, 0.75, 0.0), (CTRL, 1, 74, 70, 0.5, 0.0),
           (CTRL, 1, 7, 900, 1.0, 0.0), (CTRL, 2, 7, 900, 0.125, 0.0)]
    _, ctl = gpu_run(eng, cfg, [evs], 512, 5, vectors_per_launch=8, watch=[74, 7])
    _, want = ref_run_controllers(cfg, evs, 512, 5, [74, 7])
    assert_bits_equal(ctl[:, 0], want, True, "same-frame controller events")
    assert want[0, -1] == np.float32(0.25) and want[1, -1] == np.float32(1.0)


@pytest.mark.gpu
def test_controller_signals_survive_a_longer_reservation(eng):
    """watch_controllers again with the same numbers only changes the reserved launch length: the smoothers go on mid-glide."""
    import madronalib_amd as ml
    cfg = SCENARIOS["midi_poly4"]
    evs = [(NOTE_ON, 1, 60, 0, 0.0, 0.5), (CTRL, 1, 74, 100, 0.75, 0.0), (CTRL, 1, 1, 130, 0.5, 0.0), (CTRL, 1, 74, 700, 0.125, 0.0)]
    ev = ml.Events(eng, 1, 4)
    ev.configure(glide_seconds=cfg["glide"], drift=cfg["drift"])
    ev.watch_controllers([74, 1], 2)
    ev.add_events([0] * len(evs), [ml.Event(*e) for e in evs])
    chunks = []
    for start, n in ((0, 2), (128, 2), (256, 8), (768, 8), (1280, 4)):       # the glides of frames 100 / 130 are under way at 256
        if n > 2:
            ev.watch_controllers([74, 1], 8)
        ev.process_host(n, start)
        chunks.append(ev.controllers_host(n))
    got = np.concatenate(chunks, 2)
    _, want = ref_run_controllers(cfg, evs, 1536, 1, [74, 1])
    assert_bits_equal(got[:, 0], want, True, "controllers across a re-reservation")
    assert 0.125 < want[0, 300] < 0.75 or 0 < want[0, 300] < 0.75      # still gliding when the reservation changed
    ev.close()


@pytest.mark.gpu
def test_controller_signal_pointers_stay_put_inside_the_reservation(eng):
    """watch_controllers with the same numbers and a length INSIDE what is already reserved must not move any slot's
    signal (a host, or a recorded sequence, may hold the pointers): the slots are laid out by the reserved capacity, not by the
    current limit - slot 1's pointer used to move with every change of the limit. And the values after shrinking and growing again
    are still the reference's."""
    import madronalib_amd as ml
    cfg = SCENARIOS["midi_poly4"]
    evs = [(NOTE_ON, 1, 60, 0, 0.0, 0.5), (CTRL, 1, 74, 100, 0.75, 0.0), (CTRL, 1, 1, 130, 0.5, 0.0), (CTRL, 1, 74, 700, 0.125, 0.0)]
    ev = ml.Events(eng, 3, 4)
    ev.configure(glide_seconds=cfg["glide"], drift=cfg["drift"])
    ev.watch_controllers([74, 1], 8)
    p0 = [ev.controller_signal(0), ev.controller_signal(1)]
    ev.add_events([0] * len(evs), [ml.Event(*e) for e in evs])
    chunks = []
    for start, n, limit in ((0, 2, 2), (128, 2, 2), (256, 8, 8), (768, 4, 4), (1024, 8, 8)):
        ev.watch_controllers([74, 1], limit)                 # shrink, grow back: always inside the first reservation of 8
        assert [ev.controller_signal(0), ev.controller_signal(1)] == p0, (start, limit)
        ev.process_host(n, start)
        chunks.append(ev.controllers_host(n))
    got = np.concatenate(chunks, 2)
    _, want = ref_run_controllers(cfg, evs, 1536, 1, [74, 1])
    assert_bits_equal(got[:, 0], want, True, "controllers across limits inside one reservation")
    ev.watch_controllers([74, 1], 16)                        # beyond it: the buffer is replaced, and says so by moving
    assert ev.controller_signal(1) != p0[1] or ev.controller_signal(0) != p0[0]
    ev.close()


@pytest.mark.gpu
def test_two_rare_coincidences_inside_one_vector(eng):
    """Found by tools/events_soak.py (8 of 2 100 random configurations), both about a voice that gets two note events in one DSPVector:
    (1) a sustain-pedal release makes its note-off with Event's default time 0 (MLEventsToSignals.cpp:833-836); for a voice whose note
    STARTED earlier in the same vector that sets nextFrameToProcess BACK to 0 (:141), and the frames written so far are written again -
    gate 0 from the vector's start, the new pitch from its start, the event age counted twice over those frames (mlev::note_rewind);
    (2) two steals of one voice on a vector's FIRST frame: the second retrigger rewrites the frame the first one wrote (:163-175), its
    reset of the event age included. Against the reference's class, every row."""
    cases = {
        "pedal release rewinds": (dict(polyphony=3, glide=0.01, drift=0.0),
                                  [(SUSTAIN, 1, 0, 10, 1.0, 0.0), (NOTE_ON, 1, 60, 64 + 21, 0.0, 0.7), (NOTE_OFF, 1, 60, 64 + 38, 0.0, 0.0),
                                   (SUSTAIN, 1, 0, 64 + 40, 0.0, 0.0), (NOTE_ON, 1, 64, 64 + 50, 0.33, 0.5), (NOTE_OFF, 1, 64, 300, 0.0, 0.0),
                                   # ... and with the three events on ONE frame (the buffer holds them in reverse order of arrival, :372-377)
                                   (SUSTAIN, 1, 0, 320, 1.0, 0.0), (NOTE_ON, 1, 50, 384 + 7, -0.8, 0.9), (SUSTAIN, 1, 0, 384 + 30, 0.0, 0.0), (NOTE_OFF, 1, 50, 384 + 30, 0.0, 0.0)]),
        "two steals on frame 0": (dict(polyphony=2, glide=0.02, drift=0.3),
                                  [(NOTE_ON, 1, 40, 5, -1.6, 0.8), (NOTE_ON, 1, 45, 30, -1.25, 0.6), (NOTE_ON, 1, 70, 128, 0.83, 0.4), (NOTE_ON, 1, 72, 128, 1.0, 0.9),
                                   (NOTE_ON, 1, 74, 256, 1.16, 0.5), (NOTE_ON, 1, 76, 256, 1.33, 0.3), (NOTE_ON, 1, 78, 256, 1.5, 0.2), (NOTE_OFF, 1, 78, 400, 0.0, 0.0)]),
        "one voice, two steals on frame 0": (dict(polyphony=1, glide=0.0, drift=0.0),
                                             [(NOTE_ON, 1, 40, 5, -1.6, 0.8), (NOTE_ON, 1, 70, 192, 0.83, 0.4), (NOTE_ON, 1, 72, 192, 1.0, 0.9)]),
    }
    for name, (cfg, evs) in cases.items():
        for block, vpl in ((512, 3), (64, 1), (256, 4)):
            n_blocks = 1024 // block
            got = gpu_run(eng, cfg, [evs, evs[:3]], block, n_blocks, vectors_per_launch=vpl)
            P = cfg["polyphony"]
            for k, e in enumerate([evs, evs[:3]]):
                want = ref_run(cfg, e, block, n_blocks)
                for r in range(8):
                    assert_bits_equal(got[r, k * P:(k + 1) * P], want[r], True, f"{name}, blocks of {block}: instrument {k} row {ROW_NAMES[r]}")


@pytest.mark.gpu
def test_events_random_configurations(eng):
    """A short run of tools/events_soak.py: random polyphony / protocol / sample rate / glide / drift / block and launch sizes, voice rows and
    controller signals against the reference's class. (2 100 configurations of it at the round's end: profiles/r06_events_soak.txt.)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("events_soak", os.path.join(ROOT, "tools", "events_soak.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(120, 3, eng) == 0
