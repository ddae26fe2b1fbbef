"""TEST INFRASTRUCTURE: 512-voice instances of the widened rows' bench workloads (the instrument bank, plucked strings,
EventsToSignals, the Downsampler, the reference's reverb example) on the device and on the CPU checker. Used twice: as GPU parity tests
(tests/test_gpu_widened_parity.py, bit for bit) and by bench.py's cpu_baseline leg, which prints the CRC-32 of both sides in the
driver's line (`<leg>_crc_match`). Every case returns (got, want, checker) with got / want float32 arrays of the same shape.

The workloads are bench.py's own (same graphs, same per-voice parameter functions, same kinds of input); only the size differs."""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VOICES = 512


def _have_ref():
    return os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libdropin_ref.so")) and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libexamples_ref.so"))


def resample_case(eng, orc):
    """bench.py --workload resample: Downsampler, 2 octaves, NoiseGen input (seed = voice index), 32 DSPVectors in -> 8 out."""
    import madronalib_amd as ml
    from inputs import lcg_noise
    V, T = VOICES, 32
    x = lcg_noise(np.arange(V, dtype=np.uint32), 64 * T)
    r = ml.Resampler(eng, V, 2, False)
    got = r.process_host(x)
    st = np.zeros((2 * 9, V), np.float32)
    want = orc.resample(2, False, st, x)
    r.close()
    return got, want, "oracle"


def strings_case(eng, orc, layout=2):
    """bench.py --workload strings: noise -> + feedback x 0.995 -> FractionalDelay of per-voice length (46 .. 736 Hz, scattered) -> OnePole,
    in ring layout `layout`; 512 voices x 16 DSPVectors x 2 launches."""
    import madronalib_amd as ml
    from graph_oracle import evaluate_stream, new_stream_state
    from inputs import lcg_noise
    from madronalib_amd.constants import Layout, Op, Proc
    V, T, launches = VOICES, 16, 2
    desc = [dict(name="x", type="input"), dict(name="g", type="const", value=0.995),
            dict(name="fb", type="feedback", source="damp"),
            dict(name="fbg", type="op", kind=Op.MULTIPLY, inputs=["fb", "g"]),
            dict(name="sum", type="op", kind=Op.ADD, inputs=["x", "fbg"]),
            dict(name="line", type="proc", kind=Proc.FRACTIONAL_DELAY, inputs=["sum"], max_delay=1024.0),
            dict(name="damp", type="proc", kind=Proc.ONE_POLE, inputs=["line"])]
    length = (48000.0 / (46.0 * 2.0 ** (4.0 * ((np.arange(V) * 7919) % V) / V)) - 64.0).astype(np.float32)
    x = lcg_noise(np.arange(V, dtype=np.uint32), 64 * T * launches)
    co = orc.make_coeffs("onepole", 0.3)
    g = ml.Graph(eng, V, desc, ["damp"], delay_windows=layout)
    g.set_coeffs("damp", [np.full(V, c, np.float32) for c in co])
    st = new_stream_state(orc, desc, V)
    fs = np.stack([orc.fractional_delay_state(float(d)) for d in length], 1)
    st["line"][3] = fs[0].view(np.uint32)
    st["line"][4] = fs[1].view(np.uint32)
    for i in range(st["line"].shape[0]):
        g.set_state("line", i, st["line"][i])
    got, want = [], []
    for k in range(launches):
        part = {"x": np.ascontiguousarray(x[:, k * 64 * T:(k + 1) * 64 * T])}
        got.append(g.process_host(T, part, Layout.QUAD)[0])
        want.append(evaluate_stream(orc, desc, ["damp"], V, T, part, {}, {"damp": np.repeat(np.asarray(co, np.float32).reshape(-1, 1), V, 1)}, st)[0])
    g.close()
    return np.concatenate(got, 1), np.concatenate(want, 1), "oracle"


def _performances(N, P, frames):
    from test_gpu_events import performance
    return [performance("midi", 7 * k + 3, frames, P) for k in range(N)]


EVENTS_CFG = dict(polyphony=16, glide=0.01, drift=0.5)     # bench.py: ev.configure(glide_seconds=0.01, drift=0.5), 16 voices per instrument


def events_case(eng):
    """bench.py --workload events: 32 instruments x 16 voices, all 8 rows, scripted performances, against the reference's own
    EventsToSignals class (oracle/_ref/libdropin_ref.so), instrument by instrument."""
    from test_gpu_events import gpu_run, ref_run
    P = EVENTS_CFG["polyphony"]
    N, block, n_blocks = VOICES // P, 1024, 2
    inst = _performances(N, P, block * n_blocks)
    got = gpu_run(eng, EVENTS_CFG, inst, block, n_blocks, vectors_per_launch=16)
    want = np.concatenate([ref_run(EVENTS_CFG, evs, block, n_blocks) for evs in inst], 1)
    return got, want, "reference"


def synth_case(eng, orc):
    """bench.py --workload synth (the instrument bank in its one-voice-kernel form): note events -> control records -> the 16-node voice
    with pitch and gate made inside the voice kernel -> each instrument's 16 voices summed inside it. CPU side: pitch and gate rows from
    the reference's EventsToSignals class, the voice graph node by node with the oracle, `outputs += voice` in voice order
    (source/app/MLSynth.h:43-57)."""
    import madronalib_amd as ml
    from graph_oracle import evaluate
    from madronalib_amd import patches
    from madronalib_amd.constants import Layout
    from madronalib_amd.sharding import cfg5_voice_params
    from test_gpu_events import ref_run
    P = EVENTS_CFG["polyphony"]
    N, T, n_blocks = VOICES // P, 16, 2
    V, block = N * P, 64 * T
    inst = _performances(N, P, block * n_blocks)
    ev = ml.Events(eng, N, P, 48000.0)
    ev.configure(glide_seconds=EVENTS_CFG["glide"], drift=EVENTS_CFG["drift"])
    ev.set_wanted_rows([0, 1])
    ev.reserve_for_graph(T)
    desc, outn = patches.synth16(pitch_input=True, event_rows=True)
    g = ml.Graph(eng, V, desc, outn, output_groups={0: P})
    g.bind_events(ev)
    g.clear()
    params, coeffs, seeds = cfg5_voice_params(0, V, V, ml)
    for k, v in params.items():
        if k != "pitch":
            g.set_param(k, v if np.ndim(v) else float(v))
    for k, c in coeffs.items():
        g.set_coeffs(k, [np.ascontiguousarray(r) for r in c])
    g.set_state("noise", 0, seeds)
    d_mix = eng.alloc(4 * N * T * 64)
    chunks = []
    for b in range(n_blocks):
        bi, be = [], []
        for i, evs in enumerate(inst):
            for e in evs:
                if b * block <= e[3] < (b + 1) * block:
                    bi.append(i)
                    be.append(ml.Event(e[0], e[1], e[2], e[3] - b * block, e[4], e[5]))
        ev.add_events(bi, be)
        g.process_events(T, 0, [], [d_mix], out_layout=Layout.VOICE_MAJOR)
        ev.clear_events()
        chunks.append(d_mix.download(np.float32, N * T * 64).reshape(N, T * 64).copy())
    got = np.concatenate(chunks, 1)
    g.close()
    # CPU: rows of the reference class -> the same voice graph with the rows as inputs -> voices added up in voice order
    rows = np.concatenate([ref_run(EVENTS_CFG, evs, block, n_blocks) for evs in inst], 1)      # [8][V][frames]
    desc_in, outn_in = patches.synth16(pitch_input=True)
    states = {d["name"]: orc.chain_clear([d["kind"]], V) for d in desc_in if d["type"] == "proc"}
    states["noise"][0] = seeds
    cpu_params = {k: v for k, v in params.items() if k != "pitch"}
    cpu_coeffs = {k: np.ascontiguousarray(np.stack([np.broadcast_to(np.asarray(r, np.float32), (V,)) for r in c])) for k, c in coeffs.items()}
    voices = evaluate(orc, desc_in, outn_in, V, T * n_blocks, {"pitch": rows[0], "gate": rows[1]}, cpu_params, cpu_coeffs, states)[0]
    want = np.zeros((N, voices.shape[1]), np.float32)
    for p in range(P):
        want = (want + voices.reshape(N, P, -1)[:, p]).astype(np.float32)
    return got, want, "reference (EventsToSignals) + oracle (voice graph)"


def reverb_case(eng_unused=None):
    """bench.py --workload reverb: the reference's examples/audio-and-midi/reverb.cpp compiled unchanged against the shim
    (tests/cpp/libexamples_gpu.so) and against the reference (oracle/_ref/libexamples_ref.so); stereo noise in, both outputs."""
    from inputs import lcg_noise
    V, T = VOICES, 16
    G = ctypes.CDLL(os.path.join(ROOT, "tests", "cpp", "libexamples_gpu.so"))
    R = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libexamples_ref.so"))
    fp = ctypes.POINTER(ctypes.c_float)
    G.example_reverb_gpu_run.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, fp, fp, fp, fp, ctypes.c_char_p, ctypes.c_size_t]
    R.example_reverb_ref_run.argtypes = [ctypes.c_size_t, ctypes.c_size_t, fp, fp, fp, fp]
    x0 = (lcg_noise(np.arange(V, dtype=np.uint32), 64 * T) * np.float32(0.05)).astype(np.float32)
    x1 = (lcg_noise(np.arange(V, dtype=np.uint32) + (1 << 20), 64 * T) * np.float32(0.05)).astype(np.float32)
    g0, g1, w0, w1 = (np.zeros((V, 64 * T), np.float32) for _ in range(4))
    p = lambda a: a.ctypes.data_as(fp)  # noqa: E731
    err = ctypes.create_string_buffer(2048)
    st = G.example_reverb_gpu_run(V, T, 2, p(x0), p(x1), p(g0), p(g1), err, 2048)
    if st:
        raise RuntimeError("reverb example on the device: " + err.value.decode())
    assert R.example_reverb_ref_run(V, T, p(x0), p(x1), p(w0), p(w1)) == 0
    return np.stack([g0, g1]), np.stack([w0, w1]), "reference"


def all_cases(eng, orc):
    """name -> thunk; the cases that need the compiled reference are left out where it is absent."""
    cases = {"resample": lambda: resample_case(eng, orc), "strings": lambda: strings_case(eng, orc)}
    if _have_ref():
        cases.update({"events": lambda: events_case(eng), "synth": lambda: synth_case(eng, orc)})
        if os.path.exists(os.path.join(ROOT, "tests", "cpp", "libexamples_gpu.so")):
            cases["reverb"] = lambda: reverb_case()
    return cases
