#!/usr/bin/env python3
"""bench.py — voice-samples/s of the mldsp.h hot path on N MI355X (one process per GPU).

    python bench.py --gpus N --steps K --warmup W
        N > 1 by itself: starts N rank processes, rank r on GPU r (or `--launcher threads`: N host threads, one engine per
        device, in this process); under `python -m torch.distributed.run --nproc-per-node N ...` each process is one rank.
        Asking for more GPUs than are visible is an error, never a silent smaller run.

Default workload = BASELINE.json configs[2], the configuration the metric
"voice-samples/sec (SawGen->SVF chain)" is quoted on:
    262 144 voices per GPU, SawGen -> Bandpass(k=0.5) -> x0.25, per-voice freq 55 Hz..1.76 kHz at
    48 kHz, scalar-freq mode, free-running (max-throughput) streaming.
A "step" is one pass of the hot path over one batch = ONE SECOND OF AUDIO for every voice of this
rank's partition: 750 DSPVectors (SURVEY §8d "sustained for >= 1 s of audio"), issued as 25
launches of the fused voice-bank kernel x 30 DSPVectors each into a ring of two output signals;
per-voice freq, coefficients and state are resident in HBM and carried from launch to launch.
Voices shard embarrassingly: rank g owns voices [g*V, (g+1)*V) — weak scaling, no data-path
collective; ranks meet only for the barrier on both sides of the timed region and the max-over-ranks
time (madronalib_amd/rendezvous.py: files, threads or gloo — never RCCL, there is nothing to reduce).

Prints ONE JSON line on rank 0 with
  roofline      HBM bound: algorithmic bytes per launch / average launch duration measured with HIP
                events on the engine's stream over the timed region; `traffic` = HBM bytes per
                launch from the rocprofv3 PMC passes of exactly this workload and size (profiles/pmc_workloads.json,
                else null); `valu` = the VALU-issue bound from SQ_INSTS_VALU of the same passes; `bound` names
                whichever fraction is higher
  cpu_baseline  (N=1 only) the compiled reference (oracle/_ref) when present, else the plain-C port,
                timed on the host cores over a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOADS = {
    # name: (voices per GPU, DSPVectors per launch, launches per step)
    "cfg3": (262144, 30, 25),   # 750 vectors = 1 s of audio at 48 kHz
    "cfg4": (131072, 32, 16),   # long-buffer: 512 vectors per step
    "cfg2": (65536, 1, 64),     # elementwise: 64 vector-steps per step
    "cfg5": (262144, 16, 16),   # synth16 graph: 256 vectors per step
    "cfg5full": (262144, 16, 16),  # the same voice as SURVEY 8d lists it: + filter ADSR, cutoff exp2Approx, Lopass(x, omega, k)
    # widened rows, measured to the same bar; not BASELINE configs
    "events": (262144, 16, 8),  # EventsToSignals: 16384 instruments x 16 voices, 8 control rows out
    "resample": (262144, 32, 8),  # Downsampler, 2 octaves: 32 vectors in -> 8 out per launch
    # the instrument bank end to end (16 384 instruments x 16 voices): note events -> EventsToSignals -> synth16 voices -> per-instrument sum.
    # "synth": the form a host should use (round 5) - the events' control-rate half as a light kernel (16-byte control records), the voice
    # kernel expanding them and adding up each instrument's voices; "synthrows": the three-kernel form (e2s_kernel writes pitch and gate
    # rows, the voice graph reads them, mlgpu_mixdown_groups adds up the voices; MLGPU_BENCH_MIXDOWN=graph: the sum inside the voice kernel)
    "synth": (262144, 16, 8),
    "synthrows": (262144, 16, 8),
    "cfg5mix": (262144, 16, 16),  # config 5's voices summed to one channel in the voice kernel (mlgpu_graph_set_output_mixdown)
    "strings": (262144, 16, 8),  # a plucked-string model per voice: noise burst -> FractionalDelay (per-voice length) -> OnePole -> feedback
    "mixgroups": (262144, 8, 16),  # the per-instrument voice sum alone, 16 voices per instrument
    "allpass4": (131072, 16, 8),  # 4 x Allpass<PitchbendableDelay> in series per voice (8 rings of 4096 samples), per-voice delay times
    # the reference's own examples/audio-and-midi/reverb.cpp (the Aaltoverb algorithm: 10 x Allpass<PitchbendableDelay> + 2 PitchbendableDelay = 24
    # rings, two kept DSPVectors), compiled UNCHANGED against the shim (tests/cpp/libexamples_gpu.so), one stereo reverb per voice
    "reverb": (65536, 16, 8),
    # BASELINE.json north_star "Target": >= 10^6 SawGen -> SVF -> gain voices at 48 kHz real time on one GPU (paced, not free-running)
    "rt": (1048576, 1, 75),      # 2^20 voices, one 64-frame block per call paced at the 1333 us block period; a step = 75 blocks = 0.1 s of audio
}


# What the dominant kernel of the workload being set up reads with wide coalesced loads, per launch, when that is NOT all of
# its reads (the delay-ring graphs read 32-byte sectors at per-voice positions besides their streamed signals). rocprofv3's
# FETCH_SIZE tallies a 128-byte request of a coalesced 16 B / lane stream at 64 B (the guide's "doubles on gfx950") but a
# scattered sector read at its true 64 B (tools/fetchcal.hip: 1 GiB read once = 0.50 GiB counted as a stream, 0.97 GiB as
# 64-byte sectors, 1.91 GiB as scattered 32-byte sectors - the half line that comes along is real traffic). So
#   fetched bytes = raw + min(raw, coalesced / 2)      (everything coalesced: 2 x raw, the guide's rule)
META = {"coalesced_read_bytes": None, "alg_is_upper_bound": False}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


from madronalib_amd.sharding import cfg3_voice_params, partition  # noqa: E402


def cfg3_params(lo, hi, total):
    """Per-voice freq and Bandpass coefficients of config 3 for global voices [lo, hi)."""
    import madronalib_amd as ml
    return cfg3_voice_params(lo, hi, total, ml.Bandpass.makeCoeffs)


def setup_workload(eng, name, V, T, lo, total):
    """Returns (launch_fn, algorithmic_bytes_per_launch, kernel_name, description)."""
    import madronalib_amd as ml
    from madronalib_amd.constants import Layout, Op, Proc
    n = V * T * 64
    if name == "cfg3":
        bank = eng.bank([Proc.SAW_GEN, Proc.BANDPASS, Proc.GAIN], V)
        bank.clear()
        freq, co = cfg3_params(lo, lo + V, total)
        for i in range(3):
            bank.set_coeff(1, i, co[i])
        bank.set_coeff(2, 0, 0.25)
        bank.set_input_const(freq)
        outs = [eng.alloc(4 * n), eng.alloc(4 * n)]
        k = [0]

        def launch():
            bank.process(T, outs[k[0] & 1], Layout.QUAD)
            k[0] += 1
        # 4 B/voice-sample written + per launch and voice
        # (4 phase + 8 ic + 12 coeff + 4 gain + 4 freq) read and 12 written
        alg = 4.0 * n + V * (4 + 8 + 12 + 4 + 4 + 12)
        desc = ("BASELINE configs[2]: 262144 voices/GPU SawGen->Bandpass(k=0.5)->gain 0.25, scalar-freq, "
                "free-running; step = 1 s of audio (750 DSPVectors) per voice")
        return launch, alg, bank.kernel_name, desc, bank
    if name == "cfg4":
        bank = eng.bank([Proc.LOPASS] * 8, V)
        for i in range(8):
            bank.set_coeffs(i, ml.Lopass.makeCoeffs(float(np.float32(0.02) * np.float32(i + 1)), 0.7))
        nb = eng.bank([Proc.NOISE_GEN], V)
        nb.set_state(0, 0, np.arange(lo, lo + V, dtype=np.uint32))
        d_x = eng.alloc(4 * n)
        nb.process(T, d_x, Layout.QUAD)
        outs = [eng.alloc(4 * n), eng.alloc(4 * n)]
        k = [0]

        def launch():
            bank.process(T, outs[k[0] & 1], Layout.QUAD, d_x, Layout.QUAD)
            k[0] += 1
        alg = 8.0 * n + V * (4 * (24 + 16) + 4 * 16)
        return launch, alg, bank.kernel_name, "BASELINE configs[3]: 131072 channels x 8 cascaded Lopass, streamed noise input", bank
    if name == "cfg2":
        x = np.tile(np.linspace(-np.pi, np.pi, 4096, dtype=np.float32), (V * 64 * T) // 4096)
        d_x = eng.to_device(x)
        d_y = eng.alloc(4 * n)

        def launch():
            eng.op_apply(Op.EXP_APPROX_OF_SIN_APPROX, d_x, None, None, d_y, n)
        return launch, 8.0 * n, "op_kernel<22>", \
            "BASELINE configs[1]: 65536 voices x 1 DSPVector elementwise expApprox(sinApprox(x)) (32 MiB: Infinity-Cache resident)", (d_x, d_y)
    if name in ("cfg5", "cfg5full", "cfg5mix"):
        from madronalib_amd import patches
        from madronalib_amd.sharding import cfg5_gate_quad, cfg5_voice_params
        full = name == "cfg5full"
        desc, outs = patches.synth16(full=full)
        g = ml.Graph(eng, V, desc, outs, voices_per_lane=int(os.environ.get("MLGPU_VOICES_PER_LANE", "0")), autotune=bool(os.environ.get("MLGPU_BENCH_AUTOTUNE")),
                     compile_now=False)
        if name == "cfg5mix":      # config 5's voices to ONE channel: the output mixed down inside the voice kernel (round 5)
            g.set_output_mixdown(0)
        g.compile()
        if name == "cfg5mix":
            g.reserve_mixdown(T)
        g.clear()
        params, coeffs, seeds = cfg5_voice_params(lo, lo + V, total, ml, full=full)
        for k, v in params.items():
            g.set_param(k, v if np.ndim(v) else float(v))
        for k, c in coeffs.items():
            g.set_coeffs(k, [np.ascontiguousarray(r) for r in c])
        g.set_state("noise", 0, seeds)
        d_gate = eng.to_device(cfg5_gate_quad(lo, lo + V, T))
        outs_d = [eng.alloc(4 * (T * 64 if name == "cfg5mix" else n)) for _ in range(2)]
        k = [0]

        def launch():
            g.process(T, [d_gate], [outs_d[k[0] & 1]])
            k[0] += 1
        if name == "cfg5mix":
            return launch, 4.0 * n + V * 4.0 * (5 + 14 + 19 + 19), "mlgpu_graph_kernel", (
                "BASELINE configs[4]'s 16-node voice, 262144 voices/GPU mixed to ONE channel inside the voice kernel "
                "(mlgpu_graph_set_output_mixdown: gate in, no per-voice audio out)"), g
        # gate in + audio out per voice-sample; per launch and voice: 5 params + 14 coeffs + 19 state words
        # read, 19 state words written (patches.synth16: NC = 3+4+2+1+4, NS = 1+1+1+1+2+2+1+2+8)
        alg = 8.0 * n + V * 4.0 * (5 + 14 + 19 + 19)
        if full:   # 9 params; coefficients: hp 4 + smooth 2 + dc 1 + two ADSRs 4 + 4 (the Lopass has none); state: + the second ADSR's 8 words
            alg = 8.0 * n + V * 4.0 * (9 + 15 + 27 + 27)
            return launch, alg, "mlgpu_graph_kernel", ("BASELINE configs[4] as SURVEY 8d lists it: 22 processor/op nodes per voice incl. a filter ADSR, the cutoff "
                                                        "through exp2Approx and Lopass(x, omega, k) with per-sample coefficients (two libm sinf per sample on the "
                                                        "device); 262144 voices/GPU, streamed gate in, audio out"), g
        return launch, alg, "mlgpu_graph_kernel", ("BASELINE configs[4]: 16-node synth patch (run-time graph fused by hiprtc), "
                                                    "262144 voices/GPU, streamed gate in, audio out"), g
    if name == "events":
        P = 16
        N = V // P
        ev = ml.Events(eng, N, P, 48000.0)
        ev.configure(glide_seconds=0.01, drift=0.5)
        rows = [int(r) for r in os.environ.get("MLGPU_EVENT_ROWS", "0,1,2,3,4,5,6,7").split(",")]   # e.g. "0,1": pitch and gate only
        ev.set_wanted_rows(rows)
        rng = np.random.default_rng(lo + 1)
        held = {}
        outs = [[eng.alloc(4 * n) if r in rows else None for r in range(8)] for _ in range(2)]
        k = [0]

        quiet_after = int(os.environ.get("MLGPU_BENCH_EVENTS_UNTIL", "-1"))   # >= 0: no more events after that many launches

        def launch():
            # a sparse performance: every launch ~2 % of the instruments get a note on or off somewhere in the block
            insts, evs = [], []
            for i in (rng.integers(0, N, max(1, N // 50)) if (quiet_after < 0 or k[0] < quiet_after) else []):
                i = int(i)
                t = int(rng.integers(0, 64 * T))
                insts.append(i)
                if held.get(i):
                    evs.append(ml.Event(4, 1, held[i].pop(), t, 0.0, 0.0))
                else:
                    key = int(rng.integers(36, 84))
                    held.setdefault(i, []).append(key)
                    evs.append(ml.Event(1, 1, key, t, (key - 60) / 12.0, 0.8))
            ev.add_events(insts, evs)
            ev.process(T, 0, outs[k[0] & 1], Layout.QUAD)
            ev.clear_events()
            k[0] += 1
        # 8 rows x 4 B per voice-sample written; per voice and DSPVector 7 glides x 5 state words read and written, per
        # voice and launch 23 scalar state words read and written
        glides = sum({0: 2, 3: 2, 4: 1, 5: 1, 6: 1}.get(r, 0) for r in rows) + (2 if 0 not in rows else 0)   # bend and drift always run
        alg = 4.0 * len(rows) * n + V * T * 4.0 * 5 * glides * 2 + V * 4.0 * 23 * 2
        return launch, alg, "e2s_kernel", (f"EventsToSignals: 16384 instruments x 16 voices, {len(rows)} control signals out, sparse note events "
                                           "(host routing + record upload inside the step)"), ev
    if name in ("synth", "synthrows"):
        fusedRows = name == "synth"
        # A bank of polyphonic instruments end to end, as ml::gpu::SynthProgram runs a Synth subclass: EventsToSignals (only the
        # rows the voice reads) -> the fused voice graph -> the per-instrument voice sum (Synth::processVector, MLSynth.h:43-57)
        from madronalib_amd import patches
        from madronalib_amd.sharding import cfg5_voice_params
        P = 16
        N = V // P
        # --two-streams: EventsToSignals on an engine (= HIP stream) of its own. The events kernel of block k + 1 (HBM writes)
        # then runs under the voice kernel of block k (VALU); the two meet at fences around the double-buffered row signals.
        two = bool(getattr(eng, "_bench_two_streams", False)) and not fusedRows
        evEng = ml.Engine(eng.device, urgency=int(os.environ.get("MLGPU_BENCH_EVENTS_URGENCY", "1"))) if two else eng
        ev = ml.Events(evEng, N, P, 48000.0)
        ev.configure(glide_seconds=0.01, drift=0.5)
        ev.set_wanted_rows([0, 1])
        desc, outs = patches.synth16(pitch_input=True, event_rows=fusedRows)
        # MLGPU_BENCH_MIXDOWN=graph: the per-instrument voice sum is made inside the voice kernel (mlgpu_graph_set_output_group_sum)
        # instead of by mlgpu_mixdown_groups
        sumInKernel = os.environ.get("MLGPU_BENCH_MIXDOWN", "graph" if fusedRows else "kernel") == "graph"
        g = ml.Graph(eng, V, desc, outs, voices_per_lane=int(os.environ.get("MLGPU_VOICES_PER_LANE", "0")), autotune=bool(os.environ.get("MLGPU_BENCH_AUTOTUNE")),
                     output_groups={0: P} if sumInKernel else None)
        if fusedRows:
            g.bind_events(ev)
        g.clear()
        params, coeffs, seeds = cfg5_voice_params(lo, lo + V, total, ml)
        for k, v in params.items():
            if k != "pitch":
                g.set_param(k, v if np.ndim(v) else float(v))
        for k, c in coeffs.items():
            g.set_coeffs(k, [np.ascontiguousarray(r) for r in c])
        g.set_state("noise", 0, seeds)
        rows = [eng.alloc(4 * n), eng.alloc(4 * n)]
        rowSets = [rows, [eng.alloc(4 * n), eng.alloc(4 * n)]] if two else [rows]
        rowsReady = [evEng.fence() for _ in rowSets]     # the events kernel has written this set
        rowsFree = [eng.fence() for _ in rowSets]        # the voice kernel has read it
        d_voices = eng.alloc(4 * n)
        d_mix = [eng.alloc(4 * N * T * 64), eng.alloc(4 * N * T * 64)]
        rng = np.random.default_rng(lo + 1)
        held = {}
        k = [0]
        names = [d["name"] for d in desc if d["type"] == "input"]   # graph input order: gate, pitch

        # The performance - a sparse one: every block ~2 % of the instruments get a note on or off somewhere in the block - is synthetic
        # INPUT, made before the clock starts like every other workload's input (round 4 made it inside the step, in Python, about a
        # millisecond per block: fine next to a 1.3 ms block, not next to a 0.8 ms one). The host's routing of these events into
        # per-voice records and their upload stay inside the step: that is mlgpu's own work.
        def make_block():
            insts, evs = [], []
            for i in rng.integers(0, N, max(1, N // 50)):
                i = int(i)
                t = int(rng.integers(0, 64 * T))
                insts.append(i)
                if held.get(i):
                    evs.append(ml.Event(4, 1, held[i].pop(), t, 0.0, 0.0))
                else:
                    key = int(rng.integers(36, 84))
                    held.setdefault(i, []).append(key)
                    evs.append(ml.Event(1, 1, key, t, (key - 60) / 12.0, 0.8))
            return ml.Events.pack_events(insts, evs)
        blocks = [make_block() for _ in range(int(getattr(eng, "_bench_launches", 0)) + 40 + 260 + 30)]   # (+ the lap leg's launches)

        def launch():
            ev.add_events_packed(blocks[k[0]] if k[0] < len(blocks) else make_block())
            if sumInKernel and fusedRows:
                g.process_events(T, 0, [], [d_mix[k[0] & 1]])
                ev.clear_events()
                k[0] += 1
                return
            if sumInKernel:
                ev.process(T, 0, [rows[0], rows[1]] + [None] * 6, Layout.QUAD)
                ev.clear_events()
                g.process(T, [rows[1] if nm == "gate" else rows[0] for nm in names], [d_mix[k[0] & 1]])
                k[0] += 1
                return
            if fusedRows:
                g.process_events(T, 0, [], [d_voices])
                ev.clear_events()
            elif two:
                s_ = k[0] & 1
                rs = rowSets[s_]
                evEng.wait(rowsFree[s_])                     # block k - 2's voice kernel is done with this set
                ev.process(T, 0, [rs[0], rs[1]] + [None] * 6, Layout.QUAD)
                ev.clear_events()
                evEng.signal(rowsReady[s_])
                eng.wait(rowsReady[s_])
                g.process(T, [rs[1] if nm == "gate" else rs[0] for nm in names], [d_voices])
                eng.signal(rowsFree[s_])
            else:
                ev.process(T, 0, [rows[0], rows[1]] + [None] * 6, Layout.QUAD)
                ev.clear_events()
                g.process(T, [rows[1] if nm == "gate" else rows[0] for nm in names], [d_voices])
            eng.mixdown_groups(d_voices, Layout.QUAD, N, P, T, d_mix[k[0] & 1])
            k[0] += 1
        # pitch + gate written and read, voice audio written and read, instrument audio written
        # fused rows (round 5): per voice and DSPVector a 16-byte control record written and read, and the drift LinearGlide's mCurrVec slot
        # of every sample read and rewritten by the voice kernel while the glide moves (8 s per glide, a new target every 8-16 s: counted
        # for every voice-sample, an upper bound)
        alg = ((8.0 if fusedRows else 8.0 + 8.0) + (0.0 if sumInKernel else 4.0 + 4.0)) * n + (32.0 * V * T if fusedRows else 0.0) + 4.0 * N * T * 64
        META["alg_is_upper_bound"] = fusedRows   # (voices whose drift glide rests, or whose instrument has not played yet, move no slot)
        return launch, alg, "mlgpu_graph_kernel", ("16384 instruments x 16 voices end to end: note events -> EventsToSignals (pitch, gate) -> 16-node "
                                                    "voice graph -> per-instrument voice sum"
                                                    + ("; pitch and gate computed inside the voice kernel, never written" if fusedRows else "")
                                                    + ("; the voice sum made inside the voice kernel" if sumInKernel else "")
                                                    + ("; EventsToSignals on a HIP stream of its own, overlapped with the voice kernel of the block before" if two else "")), (ev, g, evEng if two else None, rowSets, rowsReady, rowsFree)
    if name == "mixgroups":
        # the per-instrument voice sum alone (Synth::processVector, MLSynth.h:43-57): 16 384 instruments x 16 voices
        P = 16
        N = V // P
        src = eng.bank([Proc.NOISE_GEN], V)
        src.set_state(0, 0, np.arange(lo, lo + V, dtype=np.uint32))
        d_x = eng.alloc(4 * n)
        src.process(T, d_x, Layout.QUAD)
        d_mix = eng.alloc(4 * N * T * 64)

        def launch():
            eng.mixdown_groups(d_x, Layout.QUAD, N, P, T, d_mix)
        alg = 4.0 * n + 4.0 * N * T * 64
        return launch, alg, "mixdown_groups_kernel", "per-instrument voice sum, 16384 instruments x 16 voices, voice signals streamed in", (src,)
    if name == "resample":
        r = ml.Resampler(eng, V, 2, False)
        x = eng.bank([Proc.NOISE_GEN], V)
        x.set_state(0, 0, np.arange(lo, lo + V, dtype=np.uint32))
        d_x = eng.alloc(4 * n)
        x.process(T, d_x, Layout.QUAD)
        d_y = eng.alloc(n)

        def launch():
            r.process(T, d_x, d_y)
        alg = 4.0 * n + 1.0 * n + V * 4.0 * 18 * 2
        return launch, alg, "downsample_kernel<2>", "Downsampler, 2 octaves (two HalfBandFilters per voice), streamed noise in", (r, x)
    if name == "strings":
        # Karplus-Strong: every voice is a delay line whose length is its pitch - the case the windowed ring layout is for
        desc = [dict(name="x", type="input"), dict(name="g", type="const", value=0.995),
                dict(name="fb", type="feedback", source="damp"),
                dict(name="fbg", type="op", kind=Op.MULTIPLY, inputs=["fb", "g"]),
                dict(name="sum", type="op", kind=Op.ADD, inputs=["x", "fbg"]),
                dict(name="line", type="proc", kind=Proc.FRACTIONAL_DELAY, inputs=["sum"], max_delay=1024.0),
                dict(name="damp", type="proc", kind=Proc.ONE_POLE, inputs=["line"])]
        g = ml.Graph(eng, V, desc, ["damp"], delay_windows=int(os.environ.get("MLGPU_DELAY_WINDOWS", "0")))
        g.set_coeffs("damp", ml.OnePole.makeCoeffs(0.3))
        if os.environ.get("MLGPU_UNIFORM_DELAY"):
            length = np.full(V, 200.0)
        else:
            length = 48000.0 / (46.0 * 2.0 ** (4.0 * ((np.arange(V) * 7919) % V) / V)) - 64.0   # 46 Hz .. 736 Hz, scattered over the voices: 979 .. 1.2 samples
            # (to round 5's first profiles this was 55 .. 880 Hz, whose top 6 % of voices got NEGATIVE delay times - outside the reference's contract)
        st = np.stack([ml.FractionalDelay.makeState(float(d)) for d in np.unique(np.round(length, 2))])
        uniq = {float(d): st[i] for i, d in enumerate(np.unique(np.round(length, 2)))}
        words = np.stack([uniq[float(d)] for d in np.round(length, 2)], 1).astype(np.float32)   # [2][V]: delayInt bits, allpass coefficient
        g.set_state("line", 3, words[0].view(np.uint32))
        g.set_state("line", 4, words[1].view(np.uint32))
        nb = eng.bank([Proc.NOISE_GEN], V)
        nb.set_state(0, 0, np.arange(lo, lo + V, dtype=np.uint32))
        d_x = eng.alloc(4 * n)
        nb.process(T, d_x, Layout.QUAD)
        outs_d = [eng.alloc(4 * n), eng.alloc(4 * n)]
        k = [0]

        def launch():
            g.process(T, [d_x], [outs_d[k[0] & 1]])
            k[0] += 1
        # in + out + one ring (write + read) + the kept DSPVector (read + write)
        alg = (8.0 + 8.0 + 8.0) * n
        if not os.environ.get("MLGPU_UNIFORM_DELAY"):
            META["coalesced_read_bytes"] = 8.0 * n      # x and the kept DSPVector stream in; the ring is read sector by sector
        return launch, alg, "mlgpu_graph_kernel", "plucked strings: FractionalDelay of per-voice length (46..736 Hz) -> OnePole -> feedback, 262144 voices", (g, nb)
    if name == "allpass4":
        from madronalib_amd import patches
        desc = [dict(name="x", type="input"), dict(name="dl", type="param")]
        src = "x"
        for j in range(4):
            sub, src = patches.allpass(f"ap{j}_", src, Proc.PITCHBENDABLE_DELAY, 4096.0 - 64.0, "dl")
            desc += sub
        g = ml.Graph(eng, V, desc, [src], delay_windows=int(os.environ.get("MLGPU_DELAY_WINDOWS", "0")))
        for j in range(4):
            g.set_param(f"ap{j}_gain", 0.6)
        if os.environ.get("MLGPU_UNIFORM_DELAY"):
            g.set_param("dl", 1900.0)      # one delay time for every voice: ring reads coalesce
        else:
            g.set_param("dl", (400.0 + 3000.0 * (np.arange(V) % 97) / 96.0).astype(np.float32))
        nb = eng.bank([Proc.NOISE_GEN], V)
        nb.set_state(0, 0, np.arange(lo, lo + V, dtype=np.uint32))
        d_x = eng.alloc(4 * n)
        nb.process(T, d_x, Layout.QUAD)
        outs_d = [eng.alloc(4 * n), eng.alloc(4 * n)]
        k = [0]

        def launch():
            g.process(T, [d_x], [outs_d[k[0] & 1]])
            k[0] += 1
        # in + out, per allpass: two rings (write + read each) and one feedback vector (read + write)
        alg = 8.0 * n + 4 * (2 * 8.0 + 8.0) * n
        if not os.environ.get("MLGPU_UNIFORM_DELAY"):
            META["coalesced_read_bytes"] = (4.0 + 4 * 4.0) * n   # x and four kept DSPVectors
        if not os.environ.get("MLGPU_UNIFORM_DELAY") and g.delay_layout == 4:
            META["coalesced_read_bytes"] = (4.0 + 4 * 4.0) * n
        return launch, alg, "mlgpu_graph_kernel", f"4 x Allpass<PitchbendableDelay> in series (8 rings per voice), per-voice delay times 400..3400 samples, {V} voices, ring layout {g.delay_layout}", (g, nb)
    if name == "reverb":
        import ctypes
        so = os.path.join(ROOT, "tests", "cpp", "libexamples_gpu.so")
        if not os.path.exists(so):
            raise RuntimeError("tests/cpp/libexamples_gpu.so is not built (it is made from the reference's example source where /root/reference exists: __graft_entry__.build())")
        X = ctypes.CDLL(so)
        X.example_reverb_gpu_open.restype = ctypes.c_void_p
        X.example_reverb_gpu_open.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
        X.example_reverb_gpu_graph.restype = ctypes.c_void_p
        X.example_reverb_gpu_graph.argtypes = [ctypes.c_void_p]
        X.example_reverb_gpu_close.argtypes = [ctypes.c_void_p]
        err = ctypes.create_string_buffer(2048)
        options = int(os.environ.get("MLGPU_REVERB_OPTIONS", "0"))   # bit 0: VoiceProgramOptions::delayWindows (ring layout 3)
        prog = X.example_reverb_gpu_open(eng.h, V, options, err, 2048)
        if not prog:
            raise RuntimeError("reverb example through the shim: " + err.value.decode())
        gh = ctypes.c_void_p(X.example_reverb_gpu_graph(prog))

        class _Prog:
            def close(self_inner):
                X.example_reverb_gpu_close(ctypes.c_void_p(prog))
        nb = eng.bank([Proc.NOISE_GEN, Proc.GAIN], V)
        nb.set_coeff(1, 0, 0.05)
        d_in = []
        for seed0 in (0, 1 << 20):
            nb.set_state(0, 0, np.arange(lo + seed0, lo + seed0 + V, dtype=np.uint32))
            d = eng.alloc(4 * n)
            nb.process(T, d, Layout.QUAD)
            d_in.append(d)
        outs_d = [[eng.alloc(4 * n), eng.alloc(4 * n)] for _ in range(2)]
        pp = ctypes.c_void_p * 2
        ins_c = pp(d_in[0].ptr, d_in[1].ptr)
        outs_c = [pp(o[0].ptr, o[1].ptr) for o in outs_d]
        QUAD = int(Layout.QUAD)
        k = [0]

        def launch():
            st = eng.L.mlgpu_graph_process(gh, T, ins_c, QUAD, outs_c[k[0] & 1], QUAD)
            if st:
                raise ml.MlgpuError(st, "(mlgpu_graph_process of the captured reverb)")
            k[0] += 1
        # per reverb-sample: 2 inputs + 2 outputs (16 B), 24 delay rings written and read (8 B each: every PitchbendableDelay feeds both of
        # its FractionalDelays and reads both, MLDSPFilters.h:1050-1109), and 12 kept DSPVectors read and written (8 B each): the example's
        # two (mvFeedbackL / R) and the one every Allpass keeps of its delay's output (vy1, :1110-1160) - state the reference holds in
        # objects and this engine between DSPVectors in HBM. (PMC: 21.1 GB per launch of 65 536 voices x 16 DSPVectors against 20.4 GB
        # algorithmic = 1.03 x, profiles/r06_reverb_*.)
        alg = (16.0 + 24 * 8.0 + 12 * 8.0) * n
        return launch, alg, "mlgpu_graph_kernel", (f"the reference's examples/audio-and-midi/reverb.cpp unchanged through the shim: {V} independent stereo reverbs "
                                                    "(10 allpasses + 2 delays = 24 rings per voice, one delay time for all voices), noise in"), (_Prog(), nb, d_in, outs_d)
    raise SystemExit(f"unknown workload {name}")


# What an instruction class costs a SIMD that is kept full, ns per wave-instruction (tools/instbench.hip): plain FP32
# add / mul / fma and the simple integer and move instructions 1.09; conversions, compares, selects that read an SGPR mask, min / max /
# med3 / fract and the packed forms 1.8; double precision 1.85; the transcendental unit 3.5. The class counters tell FP32 add / mul /
# fma, conversions, transcendentals, integer and double apart; the rest of SQ_INSTS_VALU (compares, selects, min / max, moves, bit
# operations) is a mix of the first two prices, hence a low and a high figure.
ISSUE_NS = {"plain": 1.09, "slow": 1.8, "f64": 1.85, "trans": 3.5}
N_SIMD = 256 * 4
# The same classes as shares of a SIMD's 4-cycle ISSUE SLOT (round 5; calibrated on tools/instbench.hip under the occupancy counters,
# profiles/r05_valu_calibration.txt): a conversion / compare / select / min / max / packed / double-precision instruction takes a slot, a
# transcendental two, and the plain FP32 / integer / move class takes HALF a slot when the SIMD finds a second one to issue beside it
# (two per slot: 1.51 instructions per slot measured on a saturated SIMD) and a whole slot when it does not - which depends on what the
# other wavefronts are doing, not on the instruction. So the model is a bracket: every plain instruction paired (low) .. none (high,
# capped at 1); the measured figure (roofline.valu.busy_measured) says where in it the launch was. A saturated single-class kernel
# reads 0.87-0.96 on the measured scale (launch edges, the odd stall): that is its ceiling, not 1.
ISSUE_SLOTS = {"plain_paired": 0.5, "plain_alone": 1.0, "slow": 1.0, "f64": 1.0, "trans": 2.0}
BUSY_MEASURED_CEILING = 0.90

# with two wavefronts per SIMD instead of four the same instructions issue slower (profiles/archive/r03_bankbench.txt: 1.28 / 2.09 ns)
ISSUE_NS_2_WAVES = {"plain": 1.28, "slow": 2.09}


def valu_busy(pmc, kernel_ms, waves_per_simd=4.0, cycles=None):
    """(low, high) fraction of the launch during which the SIMDs' vector issue is occupied, from the instruction-class counters.
    cycles: shader cycles of the launch (GRBM_GUI_ACTIVE / 8) - the classes are then priced in 4-cycle issue slots (ISSUE_SLOTS) over those cycles: a bracket;
    without it in nanoseconds (ISSUE_NS) over kernel_ms."""
    total = pmc.get("valu_wave_insts_per_launch")
    if not total or "SQ_INSTS_VALU_ADD_F32" not in pmc:
        return None
    f32 = sum(pmc.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32"))
    f64 = sum(pmc.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
    cvt, trans = pmc.get("SQ_INSTS_VALU_CVT", 0.0), pmc.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
    integer = pmc.get("SQ_INSTS_VALU_INT32", 0.0) + pmc.get("SQ_INSTS_VALU_INT64", 0.0)
    other = max(0.0, total - f32 - f64 - cvt - trans - integer)
    # packed FP32 (two lanes' worth per instruction, v_pk_*_f32) is counted once by the F32 class counters and issues at the slow
    # class's rate: its share among the kernel's FP32 instructions comes with the record (static, from the shipped code object)
    packed = float(pmc.get("packed_f32_share") or 0.0)
    known = (f32 * (packed * ISSUE_NS["slow"] + (1.0 - packed) * ISSUE_NS["plain"]) + f64 * ISSUE_NS["f64"] + cvt * ISSUE_NS["slow"]
             + trans * ISSUE_NS["trans"] + integer * ISSUE_NS["plain"])
    span = N_SIMD * kernel_ms * 1e6
    lo, hi = (known + other * ISSUE_NS["plain"]) / span, (known + other * ISSUE_NS["slow"]) / span
    if cycles:
        c = ISSUE_SLOTS
        slots = N_SIMD * cycles / 4.0
        fixed = f32 * packed * c["slow"] + f64 * c["f64"] + cvt * c["slow"] + trans * c["trans"]
        plain = f32 * (1.0 - packed) + integer
        lo = (fixed + (plain + other) * c["plain_paired"]) / slots
        hi = min(1.0, (fixed + plain * c["plain_alone"] + other * c["slow"]) / slots)
        return {"busy_frac": [lo, hi], "packed_f32_share": packed, "wavefronts_per_simd": waves_per_simd,
                "classes_per_launch": {"f32_add_mul_fma": f32, "f64": f64, "cvt": cvt, "trans": trans, "int": integer, "other": other},
                "model": f"issue slots (4 shader cycles per SIMD) the launch's vector instructions need, by class - {ISSUE_SLOTS} of a slot each "
                         "(tools/instbench.hip under SQ_ACTIVE_INST_VALU / _VALU2, profiles/r05_valu_calibration.txt) - over the launch's slots "
                         "(1024 SIMDs x GRBM_GUI_ACTIVE / 8 / 4): low = every plain FP32 / integer / move instruction issued beside another one, "
                         "high = none (capped at 1); 'other' (compares, selects, min / max, moves) half a slot in the low and a slot in the high "
                         "figure; packed FP32 a slot. A bracket, not an estimate: busy_measured is the number"}
    if waves_per_simd <= 2.0:
        # a launch that fills two wavefront slots per SIMD (config 4: 131 072 channels): the high figure at the two-wavefront rates
        r = ISSUE_NS_2_WAVES
        hi = (f32 * (packed * r["slow"] + (1.0 - packed) * r["plain"]) + f64 * ISSUE_NS["f64"] * r["slow"] / ISSUE_NS["slow"] + cvt * r["slow"]
              + trans * ISSUE_NS["trans"] + integer * r["plain"] + other * r["slow"]) / span
    return {"busy_frac": [lo, hi],
            "packed_f32_share": packed, "wavefronts_per_simd": waves_per_simd,
            "classes_per_launch": {"f32_add_mul_fma": f32, "f64": f64, "cvt": cvt, "trans": trans, "int": integer, "other": other},
            "model": "sum over instruction classes of SQ_INSTS_VALU_* x the class's measured issue time (ns per wave-instruction per SIMD: "
                     f"{ISSUE_NS}; tools/instbench.hip, DESIGN 3.11) / (1024 SIMDs x launch time); 'other' (compares, selects, min / max, moves) priced "
                     "at the plain and at the slow rate gives the low and the high figure; packed FP32 (packed_f32_share of the FP32 class, static "
                     "from the shipped kernel: tools/kernel_mix.py) at the slow rate; a launch of two wavefronts per SIMD is priced at the "
                     f"two-wavefront rates {ISSUE_NS_2_WAVES} for the high figure; integer multiplies (quarter rate) are priced plain"}


def pattern_ceiling(eng, V, T, streamed_input, reps=12):
    """What this launch's ACCESS PATTERN reaches with no arithmetic to speak of, measured now on this box: the voice-bank kernel with a
    single Gain processor - one 16-byte nontemporal store per lane and quad into the QUAD layout, XCD-aware workgroup order, and the
    same for the loads when the workload streams an input - over the same voices x DSPVectors. GB/s of the bytes it moves."""
    from madronalib_amd.constants import Layout, Proc
    n = V * T * 64
    bank = eng.bank([Proc.GAIN], V)
    bank.set_coeff(0, 0, 0.5)
    d_out = eng.alloc(4 * n)
    d_in = None
    if streamed_input:
        d_in = eng.alloc(4 * n)
    else:
        bank.set_input_const(np.full(V, 0.25, np.float32))

    def go():
        if d_in is None:
            bank.process(T, d_out, Layout.QUAD)
        else:
            bank.process(T, d_out, Layout.QUAD, d_in, Layout.QUAD)
    for _ in range(4):
        go()
    eng.sync()
    eng.timer_start()
    for _ in range(reps):
        go()
    ms = eng.timer_stop_ms() / reps
    bank.close()
    d_out.free()
    if d_in is not None:
        d_in.free()
    return (8.0 if streamed_input else 4.0) * n / (ms * 1e-3) / 1e9, ms


def usable_cores():
    """Threads this process may really run at once: its CPU affinity mask, cut to the cgroup's CPU quota (a container on a 256-thread
    host is often allowed a fraction of it; os.cpu_count() says 256 all the same)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())   # cgroup v1
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return max(1, n)


def median_after_first(run, n=5):
    """discard the first run, report the median of at least five."""
    run()
    return float(np.median([run() for _ in range(n)]))


def cpu_baseline_cfg3(budget_s=12.0):
    """The same chain on the host cores over a bounded sample (~10-20 s of CPU work): every usable thread, one thread, and the steps
    between (the scaling says what kind of machine the number comes from: SMT siblings, a CPU quota)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cpu_checkers import Oracle, Ref, ref_available
    from madronalib_amd.constants import Proc
    cores = usable_cores()
    kind = "reference" if ref_available() else "port"
    if kind == "reference":
        ref = Ref()

        def seconds(Vs, T, n):
            freq, co = cfg3_params(0, Vs, Vs)
            return ref.bench_saw_bandpass_gain(Vs, T, freq, co[0], co[1], co[2], 0.25, n)[0]
    else:
        orc = Oracle()
        procs = [Proc.SAW_GEN, Proc.BANDPASS, Proc.GAIN]

        def seconds(Vs, T, n):
            freq, co = cfg3_params(0, Vs, Vs)
            coeffs = np.ascontiguousarray(np.concatenate([co, np.full((1, Vs), 0.25, np.float32)], 0))
            st = orc.chain_clear(procs, Vs)
            return orc.chain_time(procs, T, coeffs, st, None, freq, n)

    def rate(n, share):
        Vs = 4096 * n                                   # 4096 voices per thread: ~100 KiB of objects, cache resident like a real voice bank's
        t_cal = seconds(Vs, 16, n)
        T = int(max(16, min(4096, 16 * budget_s * share / max(t_cal, 1e-6) / 6)))
        return Vs * T * 64 / median_after_first(lambda: seconds(Vs, T, n)), Vs, T
    steps = sorted({n for n in (1, 8, 64, cores) if n <= cores})
    scaling = {}
    for n in steps:
        scaling[n], Vs, T = rate(n, 0.55 if n == cores else 0.45 / max(1, len(steps) - 1))
    note = ""
    if cores > 1 and scaling[cores] / scaling[1] < 0.5 * cores:
        note = (f"; {cores} threads give {scaling[cores] / scaling[1]:.1f} x one thread - the threads are not {cores} independent cores "
                "(SMT siblings share a core's vector units, and the host may run other jobs)")
    return {"value": scaling[cores], "unit": "voice-samples/s", "cores": cores, "kind": kind, "value_one_core": scaling[1],
            "thread_scaling": {str(k): v for k, v in scaling.items()}, "host_logical_cpus": os.cpu_count(),
            "sample": f"{Vs} voices x {T} DSPVectors of the same chain/params on {cores} threads (usable: affinity mask and cgroup quota; the host "
                      f"reports {os.cpu_count()} logical CPUs), threads started before the clock, first run discarded, median of 5 "
                      f"({'compiled reference headers, g++ -O2 -fno-strict-aliasing, SSE2' if kind == 'reference' else 'plain-C oracle port, gcc -O2'})"
                      f"; thread_scaling: the same at {', '.join(str(k) for k in steps)} threads with 4096 voices per thread" + note}


def cpu_baseline_cfg4(budget_s=10.0):
    """Config 4 on the host cores: the compiled reference (8 Lopass objects per channel, g++ -O2) when present."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cpu_checkers import Ref, ref_available
    import madronalib_amd as ml
    if not ref_available():
        return {"value": None, "unit": "voice-samples/s", "cores": os.cpu_count(), "kind": "port", "sample": "compiled reference not available"}
    ref = Ref()
    cores = usable_cores()
    Vs = 1024 * max(1, min(cores, 128))
    co = np.stack([ml.Lopass.makeCoeffs(float(np.float32(0.02) * np.float32(i + 1)), 0.7) for i in range(8)])
    t_cal, _ = ref.bench_lopass_cascade8(Vs, 16, co, cores)
    T = int(max(16, min(8192, 16 * budget_s / max(t_cal, 1e-6) / 6)))
    med = median_after_first(lambda: ref.bench_lopass_cascade8(Vs, T, co, cores)[0])
    return {"value": Vs * T * 64 / med, "unit": "voice-samples/s", "cores": cores, "kind": "reference", "host_logical_cpus": os.cpu_count(),
            "sample": f"{Vs} channels x {T} DSPVectors, 8 cascaded Lopass per channel, {cores} usable threads started before the clock, first run discarded, "
                      "median of 5 (compiled reference headers, g++ -O2 -fno-strict-aliasing, SSE2)"}


def cpu_baseline_cfg2(budget_s=8.0):
    """Config 2 on the host cores: expApprox(sinApprox(x)) over the same 4 Mi samples with the reference's ops (g++ -O2 -fno-strict-aliasing, SSE2)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cpu_checkers import Ref, ref_available
    from madronalib_amd.constants import Op
    if not ref_available():
        return {"value": None, "unit": "voice-samples/s", "cores": os.cpu_count(), "kind": "port", "sample": "compiled reference not available"}
    ref = Ref()
    cores = usable_cores()
    n = 65536 * 64
    x = np.tile(np.linspace(-np.pi, np.pi, 4096, dtype=np.float32), n // 4096)
    t_cal = ref.bench_op(Op.EXP_APPROX_OF_SIN_APPROX, x, cores, 8)
    reps = int(max(8, min(20000, 8 * budget_s / max(t_cal, 1e-6) / 6)))
    med = median_after_first(lambda: ref.bench_op(Op.EXP_APPROX_OF_SIN_APPROX, x, cores, reps))
    return {"value": n * reps / med, "unit": "voice-samples/s", "cores": cores, "kind": "reference", "host_logical_cpus": os.cpu_count(),
            "sample": f"65536 voices x 1 DSPVector, {reps} passes, {cores} usable threads started before the clock, first run discarded, median of 5 "
                      "(reference ops, g++ -O2 -fno-strict-aliasing, SSE2; the data stays in the CPU caches)"}


def cpu_baseline_cfg5full(budget_s=10.0):
    return cpu_baseline_cfg5(budget_s, full=True)


def cpu_baseline_cfg5(budget_s=10.0, full=False):
    """Config 5 on the host cores: the same 16-node voice written with the reference's objects (g++ -O2), one struct per voice."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cpu_checkers import Ref, ref_available
    import madronalib_amd as ml
    from madronalib_amd.sharding import cfg5_voice_params
    if not ref_available():
        return {"value": None, "unit": "voice-samples/s", "cores": os.cpu_count(), "kind": "port", "sample": "compiled reference not available"}
    ref = Ref()
    cores = usable_cores()
    Vs = 256 * max(1, min(cores, 256))
    params, coeffs, seeds = cfg5_voice_params(0, Vs, Vs, ml, full=full)

    def run(T):
        gate = np.zeros((Vs, 64 * T), np.float32)
        gate[:, 64:] = 0.8   # every voice sounds from the second vector on
        return (ref.synth16full_run if full else ref.synth16_run)(params, coeffs, seeds, gate, cores)[1]
    t_cal = run(16)
    T = int(max(16, min(1024, 16 * budget_s / max(t_cal, 1e-6) / 6)))
    med = median_after_first(lambda: run(T))
    return {"value": Vs * T * 64 / med, "unit": "voice-samples/s", "cores": cores, "kind": "reference", "host_logical_cpus": os.cpu_count(),
            "sample": f"{Vs} voices x {T} DSPVectors of the synth16{' (full)' if full else ''} voice, {cores} usable threads started before the clock, first run "
                      "discarded, median of 5 (reference objects, g++ -O2 -fno-strict-aliasing, SSE2)"}


# ---- every BASELINE config and the north_star target inside the one command the driver runs --------------------------------------
# `python bench.py --gpus 1` (the default workload, cfg3) keeps its headline line exactly as it is and, after its timed region,
# times the other configurations in the same process - each from >= 5 timed steps, HIP events on the engine's stream, algorithmic
# bytes as DESIGN 3.0 states them - and adds them as FLAT scalar keys under "roofline" (a driver that keeps scalars and drops nested
# objects still sees them). The CPU comparison of each configuration (a 512-voice instance against the oracle, CRC of the bits) is
# made in the cpu_baseline leg, outside every timed region.
EXTRA_CONFIGS = ("cfg4", "cfg5", "cfg5full")


def _release(*objs):
    import gc
    for o in objs:
        for x in (o if isinstance(o, (tuple, list)) else (o,)):
            for y in (x if isinstance(x, (tuple, list)) else (x,)):
                if hasattr(y, "close"):
                    try:
                        y.close()
                    except Exception:
                        pass
    gc.collect()


def timed_case(eng, name, steps=5, warm=2, env=None):
    """One more workload in this process: `steps` timed steps of its own launches-per-step after `warm` untimed ones.
    env: the workload's variant switches (what `MLGPU_DELAY_WINDOWS=2 bench.py --workload strings` sets), for the set-up only."""
    V, T, L = WORKLOADS[name]
    META["coalesced_read_bytes"] = None
    META["alg_is_upper_bound"] = False
    eng._bench_launches = (warm + steps) * L + 1
    saved = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        launch, alg, kname, desc, keep = setup_workload(eng, name, V, T, 0, V)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    for _ in range(warm * L):
        launch()
    eng.sync()
    eng.timer_start()
    for _ in range(steps * L):
        launch()
    ms = eng.timer_stop_ms() / (steps * L)
    res = {"kernel": kname, "kernel_ms": ms, "algorithmic_bytes_per_launch": alg, "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "voice_samples_per_s": float(V) * T * 64 / (ms * 1e-3), "timed_launches": steps * L, "voices": V, "vectors_per_launch": T,
           "workload": desc, **({"variant": env} if env else {}), **({"algorithmic_bytes_are_an_upper_bound": True} if META.get("alg_is_upper_bound") else {})}
    del launch
    _release(keep)
    return res


def cfg2_cases(eng, steps=5):
    """Config 2 at BASELINE's own size (65 536 voices x 1 DSPVector = 32 MiB in + out: Infinity-Cache resident, `on_die`) and the same
    kernel over 4 194 304 voices (1 GiB: the HBM figure)."""
    from madronalib_amd.constants import Op
    out = {}
    for label, V, reps in (("on_die", 65536, 64 * steps), ("hbm_1GiB", 4194304, 8 * steps)):
        n = V * 64
        d_x, d_y = eng.alloc(4 * n), eng.alloc(4 * n)
        eng.op_apply(Op.SIN_APPROX, d_x, None, None, d_y, n)     # (values are irrelevant to the rate: no data-dependent branches)
        for _ in range(8):
            eng.op_apply(Op.EXP_APPROX_OF_SIN_APPROX, d_x, None, None, d_y, n)
        eng.sync()
        eng.timer_start()
        for _ in range(reps):
            eng.op_apply(Op.EXP_APPROX_OF_SIN_APPROX, d_x, None, None, d_y, n)
        ms = eng.timer_stop_ms() / reps
        out[label] = {"kernel": "op_kernel<22>", "kernel_ms": ms, "algorithmic_bytes_per_launch": 8.0 * n, "voices": V,
                      "frac": 8.0 * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "voice_samples_per_s": n / (ms * 1e-3), "timed_launches": reps}
        d_x.free()
        d_y.free()
    return out


def rt_case(eng, V=1048576, frames=64, blocks=1500, lead_in=50):
    """BASELINE.json north_star "Target", measured as a real-time host would run it (tools/rt_target.cpp is the C++ original of this
    loop; the reference's is SignalProcessBuffer::process, source/app/MLSignalProcessBuffer.cpp:57-78): V SawGen -> Bandpass -> gain
    voices, one call per 64-frame block through mlgpu_process_buffer_process - voice bank, mixdown to one channel, D2H of that channel -
    with the host waiting out each 48 kHz block period. Reports the wall time of a call (p50 / p99 / max) against the period, the
    calls that took longer than the period, and the voice kernel's free-running HIP-event time per block against the HBM peak."""
    import ctypes
    import gc
    import madronalib_amd as ml
    from madronalib_amd.constants import Layout, Proc
    T = frames // 64
    bank = eng.bank([Proc.SAW_GEN, Proc.BANDPASS, Proc.GAIN], V)
    bank.clear()
    freq, co = cfg3_params(0, V, V)
    for i in range(3):
        bank.set_coeff(1, i, co[i])
    bank.set_coeff(2, 0, 0.25)
    bank.set_input_const(freq)
    d_voices = eng.alloc(4 * V * (T + 1) * 64)
    eng.mixdown_reserve(V, T + 1)
    reps = max(20, min(400, int(2.0e11 / (float(V) * frames))))
    for _ in range(10):
        bank.process(T, d_voices, Layout.QUAD)
    eng.sync()
    eng.timer_start()
    for _ in range(reps):
        bank.process(T, d_voices, Layout.QUAD)
    kernel_us = eng.timer_stop_ms() * 1e3 / reps
    alg = float(V) * (4.0 * frames + 44.0)
    period_us = frames / 48000.0 * 1e6
    L, QUAD = eng.L, int(Layout.QUAD)
    pb = ml.ProcessBuffer(eng, 0, 1, frames)
    out = np.zeros(frames, np.float32)
    pout = (ctypes.c_void_p * 1)(out.ctypes.data)
    pin = (ctypes.c_void_p * 1)(None)
    vptr = ctypes.c_void_p(d_voices.ptr)

    def cb_two(_user, n_vectors, _d_in, d_out):
        st = L.mlgpu_bank_process(bank.h, n_vectors, None, QUAD, vptr, QUAD)
        return st or L.mlgpu_mixdown(eng.h, vptr, QUAD, V, n_vectors, None, d_out[0])

    def cb_one(_user, n_vectors, _d_in, d_out):     # round 5: the voices summed inside the voice kernel, their signals never written
        return L.mlgpu_bank_process_mixdown(bank.h, n_vectors, None, QUAD, None, d_out[0])
    form = os.environ.get("MLGPU_RT_FORM", "fused")
    fused = form in ("fused", "sequence")
    clock = time.perf_counter
    seq_state = {"ptr": None, "seq": None}

    def cb_seq(_user, n_vectors, _d_in, d_out):      # experiment: the block's launches recorded once (hipGraph), replayed per block
        ptr = int(d_out[0])
        if seq_state["seq"] is not None and ptr == seq_state["ptr"] and n_vectors == T:
            return L.mlgpu_sequence_launch(seq_state["seq"].h)
        if seq_state["ptr"] == ptr and n_vectors == T and seq_state["seq"] is None:
            with eng.record() as sq:
                st = L.mlgpu_bank_process_mixdown(bank.h, n_vectors, None, QUAD, None, d_out[0])
            seq_state["seq"] = sq
            return st or L.mlgpu_sequence_launch(sq.h)
        seq_state["ptr"] = ptr
        return L.mlgpu_bank_process_mixdown(bank.h, n_vectors, None, QUAD, None, d_out[0])

    def paced(cb, n_blocks):
        cbf = ml.ProcessBuffer._CB(cb)
        us, misses, peak = [], 0, 0.0
        nxt = clock()
        for b in range(n_blocks + lead_in):
            while clock() < nxt:
                pass
            t0 = clock()
            st = L.mlgpu_process_buffer_process(pb.h, pin, pout, frames, cbf, None)
            t1 = clock()
            if st:
                raise ml.MlgpuError(st, "(process_buffer_process in the paced loop)")
            if b >= lead_in:
                d = (t1 - t0) * 1e6
                us.append(d)
                misses += d > period_us
            peak = max(peak, float(np.abs(out).max()))
            nxt = max(nxt + period_us * 1e-6, t1)
        us.sort()
        return us, misses, peak
    gc_was = gc.isenabled()
    gc.disable()           # a collection in the middle of a block is this script's, not the device's
    try:
        us, misses, peak = paced((cb_seq if form == "sequence" else cb_one) if fused else cb_two, blocks)
        us2 = paced(cb_two, max(100, blocks // 5))[0] if fused else None
    finally:
        if gc_was:
            gc.enable()
    res = {"voices": V, "frames_per_block": frames, "blocks": blocks, "block_period_us": period_us, "voice_kernel_us_free_running": kernel_us,
           "algorithmic_bytes_per_block": alg, "kernel_frac": alg / (kernel_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
           "call_us_p50": us[len(us) // 2], "call_us_p99": us[int(len(us) * 0.99)], "call_us_max": us[-1], "misses": int(misses),
           "p99_over_period": us[int(len(us) * 0.99)] / period_us, "output_peak": peak,
           "form": "mlgpu_bank_process_mixdown (the voices summed inside the voice kernel)" if fused else "mlgpu_bank_process + mlgpu_mixdown",
           "two_calls_us_p50": us2[len(us2) // 2] if us2 else None,
           "voice_samples_per_s_free_running": float(V) * frames / (kernel_us * 1e-6),
           "what": "one synchronous mlgpu_process_buffer_process call per 64-frame block (voice bank + mixdown -> D2H of the channel), the host "
                   "(this Python process, ctypes callback) waiting out each 1333 us block period; kernel_frac from the voice kernel's "
                   "free-running HIP-event time per block (launch gap included)"}
    pb.close()
    bank.close()
    d_voices.free()
    return res


def extra_legs(eng):
    """-> (flat scalars for `roofline`, the full records)."""
    flat, full = {}, {}

    def guarded(name, fn):
        try:
            full[name] = fn()
        except Exception as ex:  # an extra leg never costs the headline
            full[name] = {"error": repr(ex)}
        return full[name]
    c2 = guarded("cfg2", lambda: cfg2_cases(eng))
    if "error" not in c2:
        flat.update({"cfg2_frac": c2["hbm_1GiB"]["frac"], "cfg2_kernel_ms": c2["hbm_1GiB"]["kernel_ms"],
                     "cfg2_on_die_frac": c2["on_die"]["frac"], "cfg2_on_die_kernel_ms": c2["on_die"]["kernel_ms"],
                     "cfg2_on_die_voice_samples_per_s": c2["on_die"]["voice_samples_per_s"]})
    for name in EXTRA_CONFIGS:
        r = guarded(name, lambda name=name: timed_case(eng, name))
        if "error" not in r:
            flat.update({f"{name}_frac": r["frac"], f"{name}_kernel_ms": r["kernel_ms"], f"{name}_voice_samples_per_s": r["voice_samples_per_s"]})
    # the widened rows, >= 5 timed steps each, in the forms a host should use: the instrument bank as one voice kernel (ms per
    # 16-vector block), plucked strings in ring layout 2, EventsToSignals (all 8 rows), the Downsampler, the reference's reverb example
    if not os.environ.get("MLGPU_BENCH_NO_WIDENED"):
        for name, env in (("synth", None), ("strings", {"MLGPU_DELAY_WINDOWS": "3"}), ("allpass4", {"MLGPU_DELAY_WINDOWS": "3"}), ("events", None), ("resample", None), ("reverb", None)):
            r = guarded(name, lambda name=name, env=env: timed_case(eng, name, env=env))
            if "error" not in r:
                flat.update({f"{name}_frac": r["frac"], f"{name}_kernel_ms": r["kernel_ms"], f"{name}_voice_samples_per_s": r["voice_samples_per_s"]})
                if name == "synth":
                    flat["synth_block_ms"] = r["kernel_ms"]     # (the block: control kernel + voice kernel with the per-instrument sum inside)
    r = guarded("rt", lambda: rt_case(eng))
    if "error" not in r:
        flat.update({"rt_voices": r["voices"], "rt_block_p50_us": r["call_us_p50"], "rt_block_p99_us": r["call_us_p99"], "rt_block_max_us": r["call_us_max"],
                     "rt_block_period_us": r["block_period_us"], "rt_misses": r["misses"], "rt_kernel_frac": r["kernel_frac"],
                     "rt_kernel_us": r["voice_kernel_us_free_running"], "rt_two_calls_block_p50_us": r["two_calls_us_p50"]})
    return flat, full


def _crc(a):
    import zlib
    return zlib.crc32(np.ascontiguousarray(a).view(np.uint8).tobytes()) & 0xFFFFFFFF


def parity_crcs(eng, Vs=512):
    """CPU leg, outside every timed region: a 512-voice instance of every configuration (the bench's own per-voice parameter functions,
    first launch from the cleared state) on the device and on the CPU checker - the compiled reference where it was built, else the
    plain-C oracle -, compared as CRC-32 of the output bits. -> {config: {"crc_gpu", "crc_cpu", "match"}}"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import madronalib_amd as ml
    from cpu_checkers import Oracle, Ref, ref_available
    from madronalib_amd import patches
    from madronalib_amd.constants import Layout, Op, Proc
    from madronalib_amd.sharding import cfg5_gate_quad, cfg5_voice_params
    orc = Oracle()
    ref = Ref() if ref_available() else None
    out = {}

    def rec(name, got, want, checker):
        g, w = _crc(got), _crc(want)
        out[name] = {"crc_gpu": g, "crc_cpu": w, "match": bool(g == w), "voices": Vs, "checker": checker}

    def vm(q, T):   # QUAD [16 T][V][4] -> VOICE_MAJOR [V][64 T]
        return np.ascontiguousarray(q.reshape(T * 16, Vs, 4).transpose(1, 0, 2).reshape(Vs, T * 64))
    # config 2: the fused op over 512 x 64 elements of the bench's ramp
    x = np.tile(np.linspace(-np.pi, np.pi, 4096, dtype=np.float32), Vs * 64 // 4096)
    rec("cfg2", eng.op(Op.EXP_APPROX_OF_SIN_APPROX, x), orc.op(Op.EXP_APPROX_OF_SIN_APPROX, x), "oracle")
    # config 3
    T = 30
    procs = [Proc.SAW_GEN, Proc.BANDPASS, Proc.GAIN]
    freq, co = cfg3_params(0, Vs, Vs)
    coeffs = np.ascontiguousarray(np.concatenate([co, np.full((1, Vs), 0.25, np.float32)], 0))
    bank = eng.bank(procs, Vs)
    bank.clear()
    bank.set_all_coeffs(coeffs)
    bank.set_input_const(freq)
    got = bank.process_host(T, None, Layout.QUAD)
    chk = ref or orc
    rec("cfg3", got, chk.chain_process(procs, T, coeffs, orc.chain_clear(procs, Vs), None, freq, n_threads=4), "reference" if ref else "oracle")
    bank.close()
    # config 4
    T = 32
    procs = [Proc.LOPASS] * 8
    cs = [ml.Lopass.makeCoeffs(float(np.float32(0.02) * np.float32(i + 1)), 0.7) for i in range(8)]
    bank = eng.bank(procs, Vs)
    for i in range(8):
        bank.set_coeffs(i, cs[i])
    nb = eng.bank([Proc.NOISE_GEN], Vs)
    nb.set_state(0, 0, np.arange(Vs, dtype=np.uint32))
    xin = nb.process_host(T, None, Layout.QUAD)
    got = bank.process_host(T, xin, Layout.QUAD)
    co8 = np.ascontiguousarray(np.repeat(np.concatenate(cs)[:, None], Vs, 1))
    rec("cfg4", got, chk.chain_process(procs, T, co8, orc.chain_clear(procs, Vs), xin, None, n_threads=4), "reference" if ref else "oracle")
    bank.close()
    nb.close()
    # config 5, both patches: the compiled reference's own objects (no plain-C form of the whole patch outside tests/graph_oracle.py)
    if ref is not None:
        T = 16
        for full in (False, True):
            desc, outs = patches.synth16(full=full)
            g = ml.Graph(eng, Vs, desc, outs)
            g.clear()
            params, cf, seeds = cfg5_voice_params(0, Vs, Vs, ml, full=full)
            for k, v in params.items():
                g.set_param(k, v if np.ndim(v) else float(v))
            for k, c in cf.items():
                g.set_coeffs(k, [np.ascontiguousarray(r) for r in c])
            g.set_state("noise", 0, seeds)
            gate_q = cfg5_gate_quad(0, Vs, T)
            d_gate, d_out = eng.to_device(gate_q), eng.alloc(4 * Vs * T * 64)
            g.process(T, [d_gate], [d_out])
            got = vm(d_out.download(np.float32), T)
            want = (ref.synth16full_run if full else ref.synth16_run)(params, cf, seeds, vm(gate_q, T), 4)[0]
            rec("cfg5full" if full else "cfg5", got, want, "reference")
            g.close()
            d_gate.free()
            d_out.free()
    # the widened rows' workloads (tests/widened_parity.py; the same cases are GPU tests): 512 voices each
    try:
        import widened_parity as wp
        for name, case in wp.all_cases(eng, orc).items():
            try:
                got, want, checker = case()
                rec(name, got, want, checker)
            except Exception as ex:
                out[name] = {"match": False, "error": repr(ex)}
    except Exception as ex:
        out["widened"] = {"match": False, "error": repr(ex)}
    return out


# The order of `roofline`'s keys: a reader that keeps only the first scalars (the round-5 driver record kept 19) sees the ones that
# carry a claim - every BASELINE config, the north_star target, the parity CRCs, the widened rows - before the diagnostics.
ROOFLINE_KEY_ORDER = ("bound", "achieved", "peak", "unit", "frac", "traffic",
                      "cfg4_frac", "cfg5_frac", "cfg5full_frac", "rt_misses", "rt_block_p99_us", "rt_kernel_frac", "crc_all_match",
                      "synth_block_ms", "strings_frac", "allpass4_frac", "events_frac", "resample_frac", "reverb_frac", "kernel_ms_p50",
                      "kernel_ms_p10", "kernel_ms_p90", "kernel_ms", "kernel", "cfg2_frac", "cfg2_on_die_frac",
                      "cfg2_crc_match", "cfg3_crc_match", "cfg4_crc_match", "cfg5_crc_match", "cfg5full_crc_match",
                      "synth_crc_match", "strings_crc_match", "events_crc_match", "resample_crc_match", "reverb_crc_match",
                      "rt_voices", "rt_block_p50_us", "rt_block_max_us", "rt_block_period_us", "rt_kernel_us")


def order_roofline(roof):
    head = {k: roof[k] for k in ROOFLINE_KEY_ORDER if k in roof}
    rest_scalars = {k: v for k, v in roof.items() if k not in head and not isinstance(v, dict) and not k.startswith("valu_")}
    diag = {k: v for k, v in roof.items() if k not in head and k not in rest_scalars}
    return {**head, **rest_scalars, **diag}


VALU_PEAK_LANE_INST = 256 * 4 * 32 * 2.4e9   # 7.86e13: 256 CUs x 4 SIMD-32 x 2.4 GHz (MI355X_MICROARCH.md: v_fma_f32 wave64 = 2 cycles)


import threading  # noqa: E402
_SETUP_LOCK = threading.Lock()
_T_PROCESS = time.perf_counter()   # this rank's process reached bench.py's top level (interpreter + numpy import are before it)


def live_traffic(workload, kernel_name, timeout_s=150):
    """HBM bytes per launch of `kernel_name`, MEASURED NOW: two rocprofv3 passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE, each a run of its own with
    --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) over a 2-step run of this script's own workload in a child
    process, corrected as the guide says (KiB; FETCH_SIZE doubled on gfx950 for coalesced streams). -> dict or None (no rocprofv3, a refused
    or failed pass: the caller falls back to the stored record of profiles/pmc_workloads.json)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not prof:
        return None
    out = {}
    env = dict(os.environ, TMPDIR="/tmp", MLGPU_BENCH_CHILD="1")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix=f"mlgpu_pmc_{counter}_", dir="/tmp")
        try:
            cmd = [prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "bench.py"),
                   "--workload", workload, "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--no-live-counters"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            if r.returncode != 0:
                return None
            hits = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
            if not hits:
                return None
            per = {}
            for row in csv.DictReader(open(hits[0])):
                if row["Counter_Name"] == counter and kernel_name.split("<")[0] in row["Kernel_Name"] and kernel_name[:30] in row["Kernel_Name"].replace("void ", ""):
                    per[row["Dispatch_Id"]] = per.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
            if len(per) < 8:
                return None
            out[counter] = 1024.0 * sum(per.values()) / len(per)
            out["launches"] = len(per)
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return {"fetch_bytes_per_launch_raw": out["FETCH_SIZE"], "write_bytes_per_launch": out["WRITE_SIZE"], "launches": out["launches"],
            "bytes_per_launch": 2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"],
            "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (--kernel-trace only, two runs of their own) of this workload, taken by this bench.py run in child "
                      "processes after its timed region; KiB x 1024, FETCH_SIZE x 2 (gfx950, coalesced 16 B / lane streams)"}


def workload_key(name, V, T):
    """What identifies one measured case in profiles/pmc_workloads.json: the workload, its variant switches and its size."""
    var = [f"{k.lower().replace('mlgpu_', '')}={os.environ[k]}" for k in ("MLGPU_DELAY_WINDOWS", "MLGPU_UNIFORM_DELAY", "MLGPU_EVENT_ROWS",
                                                                            "MLGPU_VOICES_PER_LANE", "MLGPU_FTZ") if os.environ.get(k)]
    return ":".join([name] + var + [f"{V}x{T}"])


def pmc_record(key):
    """PMC counters of the dominant kernel for exactly this case (rocprofv3 --pmc passes of tools/gpu_profile_all.sh,
    summarised per WORKLOAD in profiles/pmc_workloads.json), or None: a counter cannot be read from inside the run."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_workloads.json")) as f:
            return json.load(f).get("workloads", {}).get(key)
    except Exception:
        return None


def run_rank(args, rank, local_rank, world, rdv):
    """One rank = one engine on one device. Returns the result dict on rank 0, None elsewhere."""
    import madronalib_amd as ml
    dV, dT, dL = WORKLOADS[args.workload]
    V = args.voices or dV
    T = args.vectors or dT
    L = args.launches or dL
    total = V * world                 # weak scaling: per-GPU work fixed
    lo, hi = partition(total, world, rank)
    assert hi - lo == V

    t_rank = time.perf_counter()
    eng = ml.Engine(local_rank)
    info = eng.device_info()
    t_engine = time.perf_counter()
    if args.cascade_lanes is not None:
        eng.set_cascade_lanes(args.cascade_lanes)
    if args.strict_svf:
        eng.set_strict_svf(True)
    eng._bench_two_streams = args.two_streams
    eng._bench_launches = (args.warmup + args.steps) * L + 1   # (what a workload that prepares its input per launch makes ahead)
    if args.workload == "rt":
        return run_rt(args, eng, info, V, T, L, rank, world, rdv)
    # (ranks that are THREADS of one process take turns here: the set-up is Python and numpy under one interpreter lock, and eight
    # threads fighting for it took 7-8.5 s each in round 3 - profiles/archive/r03_multi_gpu_launch_paths.txt - where one after the other
    # they take what a process takes)
    with _SETUP_LOCK:
        launch, alg_bytes, kernel_name, desc, _keep = setup_workload(eng, args.workload, V, T, lo, total)
    launch()
    eng.sync()
    t_first = time.perf_counter()
    # what a rank spends before it can launch (the 8-rank start of a node: N engines, N parameter set-ups, one hiprtc compile
    # shared through the disk cache): reported per rank so that a slow start is visible, never part of the timed region
    startup = {"import_s": t_rank - _T_PROCESS, "engine_s": t_engine - t_rank, "setup_to_first_launch_s": t_first - t_engine,
               "jit": ml.jit_stats()}

    def step():
        for _ in range(L):
            launch()

    def fence():                      # device-wide sync (every stream of this device), then all ranks
        eng.sync()
        eng.device_sync()
        if "torch" in sys.modules and sys.modules["torch"].cuda.is_available():
            sys.modules["torch"].cuda.synchronize()
        rdv.barrier()

    for _ in range(args.warmup):
        step()
    # a graph that tunes itself over its first launches (mlgpu_graph_set_autotune) finishes that before the clock starts
    graphs = [o for o in (_keep if isinstance(_keep, tuple) else (_keep,)) if hasattr(o, "tuning")]
    for g in graphs:
        for _ in range(16):
            if g.tuning()[0]:
                break
            launch()
    t_b = time.perf_counter()
    fence()
    startup["wait_at_first_barrier_s"] = time.perf_counter() - t_b
    t0 = time.perf_counter()
    eng.timer_start()
    for _ in range(args.steps):
        step()
    kernel_ms = eng.timer_stop_ms() / (args.steps * L)   # HIP events on the engine's stream
    eng.sync()
    eng.device_sync()
    my_elapsed = time.perf_counter() - t0
    rdv.barrier()
    elapsed = rdv.max(my_elapsed)
    ranks = rdv.gather({"rank": rank, "device": local_rank, "pci_bus_id": info["pci_bus_id"], "name": info["name"], "pid": os.getpid(),
                        "voices": [lo, hi], "ms_per_step": my_elapsed / args.steps * 1e3, "kernel_ms": kernel_ms, "startup": startup})
    if rank != 0:
        return None
    buses = [r["pci_bus_id"] for r in ranks]
    if len(set(buses)) != world and not args.oversubscribe:
        raise SystemExit(f"bench.py: {world} ranks but only {len(set(buses))} distinct GPUs ({buses}): every rank must own its own device")

    units = float(total) * T * 64 * L * args.steps      # voice-samples over all ranks
    value = units / elapsed
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    units_per_launch = float(V) * T * 64
    key = workload_key(args.workload, V, T)
    pmc = pmc_record(key) or {}
    # Counters cannot be read inside this run: they come from the rocprofv3 passes of tools/gpu_profile_all.sh. They describe
    # THIS code only if they were recorded under the same device-source fingerprint (every .hip / .hpp + compiler flags)
    # and for the kernel this run launched; otherwise they are dropped and the line says so.
    pmc_stale = None
    if pmc:
        same_code = pmc.get("device_source_hash") == ml.device_source_hash()
        same_kernel = kernel_name in pmc.get("kernel", "") or args.workload not in ("cfg3", "cfg4")
        if not (same_code and same_kernel):
            pmc_stale = "recorded for another build of the device code" if not same_code else f"recorded for kernel {pmc.get('kernel', '?')[:60]}"
            pmc = {}
    traffic = None
    if pmc.get("fetch_bytes_per_launch_raw") is not None:
        raw, co = pmc["fetch_bytes_per_launch_raw"], META["coalesced_read_bytes"]
        traffic = pmc["write_bytes_per_launch"] + (2.0 * raw if co is None else raw + min(raw, 0.5 * co))
    traffic_source = "profiles/pmc_workloads.json (recorded for this build of the device code)" if traffic is not None else None
    if (world == 1 and args.workload == "cfg3" and not args.no_live_counters and not args.voices and not args.vectors and not os.environ.get("MLGPU_BENCH_CHILD")
            and not any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) and "rocprof" not in os.environ.get("LD_PRELOAD", "")):   # (not under a profiler already)
        # the headline's HBM traffic measured by THIS run (weak point of round 5: the driver never saw a counter): ~15 s
        lt = live_traffic(args.workload, kernel_name)
        if lt:
            traffic, traffic_source = lt["bytes_per_launch"], lt["source"]
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": traffic_source, "traffic_over_algorithmic": (traffic / alg_bytes if traffic else None),
            "kernel": kernel_name, "kernel_ms": kernel_ms,
            "algorithmic_bytes_per_launch": alg_bytes, "pmc_case": key if pmc else None,
            "device_code": ml.device_source_hash()[:16]}   # (what the PMC records are matched against)
    if pmc_stale:
        roof["pmc_stale"] = True
        roof["pmc_stale_reason"] = pmc_stale
    # A block of several kernels (the instrument bank: events -> voices -> per-instrument sums) is timed as a whole; the counters are
    # the voice kernel's alone, so its clock and issue figures are taken over ITS duration (as measured under the counter pass)
    span_ms, span_what = kernel_ms, "live launch duration"
    if args.workload in ("synth", "synthrows") and pmc.get("mean_us_under_pmc"):
        span_ms, span_what = pmc["mean_us_under_pmc"] * 1e-3, "the named kernel's mean duration under the counter pass (the block holds other kernels too)"
    if pmc.get("GRBM_GUI_ACTIVE"):
        # GRBM_GUI_ACTIVE sums the 8 XCDs: / 8 = shader cycles per launch; over the live launch time = the clock the chip
        # held under this load (it clocks to its power budget: MI355X_MICROARCH.md, DVFS)
        cyc = pmc["GRBM_GUI_ACTIVE"] / 8.0
        roof["clock"] = {"cycles_per_launch": cyc, "ghz_live": cyc / (span_ms * 1e-3) / 1e9,
                         "source": "GRBM_GUI_ACTIVE / 8 XCDs (profiles/pmc_workloads.json) / " + span_what}
    # what the same access pattern reaches with (almost) no arithmetic, on this box, now: the honest ceiling of an HBM-bound launch
    streamed = args.workload in ("cfg4", "cfg5", "cfg5full", "cfg2")
    if world == 1 and not args.sustained:
        # The DISTRIBUTION of the kernel's launch durations, a leg of its own after the timed region (an event after every launch; the
        # headline's mean keeps its two events): the board runs these launches at its power cap and the launch time follows the shader
        # clock it is granted (profiles/r06_cfg3_spread.md) - p10 is not a speed the chip holds.
        try:
            for _ in range(3 * L):     # (the board's clock settles over the first launches after any pause: profiles/r06_cfg3_spread.md)
                launch()
            laps = np.sort(eng.lap_times_ms(launch, max(50, min(250, 4 * L))))
            roof.update({"kernel_ms_p10": float(laps[len(laps) // 10]), "kernel_ms_p50": float(laps[len(laps) // 2]),
                         "kernel_ms_p90": float(laps[len(laps) * 9 // 10]), "kernel_ms_min": float(laps[0]), "kernel_ms_max": float(laps[-1]),
                         "kernel_ms_laps": int(len(laps))})
        except Exception as ex:
            roof["kernel_ms_laps_error"] = repr(ex)
    if world == 1 and args.workload in ("cfg3", "cfg4", "cfg5", "cfg5full"):
        try:
            gbs, cms = pattern_ceiling(eng, V, T, streamed)
            # (rounds 3-5 called this store_ceiling / stream_ceiling. It is a REFERENCE KERNEL - the same access pattern next to one multiply,
            # not power-limited, 2.38 GHz - and no ceiling for a single launch: a launch of the full chain on a boosted clock beats its mean)
            roof["same_pattern_gain_kernel"] = {
                "GB/s": gbs, "frac_of_peak": gbs / HBM_PEAK_GBS, "ms": cms,
                "what": ("one Gain processor in the same voice-bank kernel over the same voices x DSPVectors: 16-byte nontemporal stores into the QUAD layout"
                         + (", loads of the streamed input likewise" if streamed else "") + ", mean of 12 launches measured in this run")}
            roof["frac_of_ceiling"] = achieved / gbs     # (key kept: achieved / the same-pattern kernel's rate)
        except Exception as ex:
            roof["same_pattern_gain_kernel"] = {"error": str(ex)}
    if pmc.get("valu_wave_insts_per_launch"):
        # the second bound (SURVEY 8d "report both bounds"): VALU issue. SQ_INSTS_VALU counts wave-instructions; x64 lanes.
        ipu = pmc["valu_wave_insts_per_launch"] * 64.0 / units_per_launch
        lane_rate = ipu * units_per_launch / (span_ms * 1e-3)
        roof["valu"] = {"insts_per_unit": ipu, "achieved_lane_inst_per_s": lane_rate, "peak": VALU_PEAK_LANE_INST,
                        "frac": lane_rate / VALU_PEAK_LANE_INST,
                        "source": "SQ_INSTS_VALU per launch (profiles/pmc_workloads.json) x 64 lanes / " + span_what,
                        "note": "an instruction count, not a utilisation: packed FP32, compares, selects and conversions occupy the SIMD "
                                "twice as long as a plain add / mul / fma (DESIGN 3.11); config 4 at 'frac 0.33' has its VALU port 85 % busy "
                                "(profiles/archive/r03_cfg4_account.md)"}
        busy = valu_busy(pmc, span_ms, waves_per_simd=max(1.0, V / 64.0 / N_SIMD), cycles=(pmc["GRBM_GUI_ACTIVE"] / 8.0 if pmc.get("GRBM_GUI_ACTIVE") else None))
        if busy:
            roof["valu"].update(busy)
            roof["valu"]["scalar_insts_per_unit"] = pmc.get("SQ_INSTS_SALU", 0.0) * 64.0 / units_per_launch if pmc.get("SQ_INSTS_SALU") else None
        # MEASURED (round 5): the share of a SIMD's issue slots in which it issued vector work. SQ_ACTIVE_INST_VALU counts one per vector
        # instruction on gfx950 (its value equals SQ_INSTS_VALU to 0.3 %; "quad-cycles", i.e. one 4-cycle issue slot each) and
        # SQ_ACTIVE_INST_VALU2 the slots in which TWO were issued (the plain FP32 / integer class runs two per slot: 2.6 cycles per
        # instruction in tools/instbench.hip), so ACTIVE - VALU2 = slots with at least one; over the launch's slots = 1024 SIMDs x shader
        # cycles / 4, the cycles from GRBM_GUI_ACTIVE / 8 XCDs of the same counter pass. tools/instbench.hip's single-class kernels under
        # the same counters calibrate the reading (profiles/r05_valu_calibration.txt: a kernel of nothing but plain or nothing but slow
        # instructions reads 0.97-1.0). The derived metric rocprofv3 ships (VALUBusy = ACTIVE / CU_NUM / GUI_ACTIVE) reads 1.4 on this
        # chip for the same kernels: it charges four cycles to instructions that take two.
        if pmc.get("SQ_ACTIVE_INST_VALU") and pmc.get("SQ_ACTIVE_INST_VALU2") is not None and pmc.get("GRBM_GUI_ACTIVE"):
            slots = N_SIMD * (pmc["GRBM_GUI_ACTIVE"] / 8.0) / 4.0
            roof["valu"]["busy_measured"] = (pmc["SQ_ACTIVE_INST_VALU"] - pmc["SQ_ACTIVE_INST_VALU2"]) / slots
            roof["valu"]["dual_issue_share_of_busy_slots"] = pmc["SQ_ACTIVE_INST_VALU2"] / max(1.0, pmc["SQ_ACTIVE_INST_VALU"] - pmc["SQ_ACTIVE_INST_VALU2"])
            if pmc.get("SQ_BUSY_CU_CYCLES"):
                roof["valu"]["simd_slots_with_a_wavefront_resident"] = pmc["SQ_BUSY_CU_CYCLES"] / slots
            if pmc.get("SQ_INST_CYCLES_SALU"):
                roof["valu"]["scalar_busy_measured"] = pmc["SQ_INST_CYCLES_SALU"] / slots
            roof["valu"]["busy_measured_ceiling"] = BUSY_MEASURED_CEILING   # what a kernel of nothing but vector instructions reads (0.87-0.96)
            roof["valu"]["busy_measured_source"] = ("(SQ_ACTIVE_INST_VALU - SQ_ACTIVE_INST_VALU2) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 / 4), one counter pass "
                                                    "(profiles/pmc_workloads.json)")
        if roof.get("clock"):
            # the instruction rate against the vector peak AT THE CLOCK THE CHIP HELD under this load (it clocks to its power budget:
            # 1.8-2.3 GHz of a 2.4 GHz maximum), next to the nominal one above
            roof["valu"]["frac_at_live_clock"] = lane_rate / (N_SIMD * 32.0 * roof["clock"]["ghz_live"] * 1e9)
        # Which wall the launch stands at: the memory side as a fraction of what its access pattern can reach, the vector-issue side
        # as the MEASURED busy fraction where the counters are there (else the model's middle); whichever is nearer its ceiling names
        # the bound, and both are on the line.
        mem_side = roof.get("frac_of_ceiling", roof["frac"] / 0.79)          # 0.79: the store pattern's ceiling of round 3 (6.3 TB/s)
        valu_side = roof["valu"].get("busy_measured", sum(busy["busy_frac"]) / 2.0 if busy else roof["valu"]["frac"])
        roof["bound"] = "valu" if valu_side > mem_side else "hbm"
        roof["bound_evidence"] = {"memory_side_frac_of_pattern_ceiling": mem_side, "valu_issue_busy_frac": valu_side,
                                  "valu_side_is": "measured" if "busy_measured" in roof["valu"] else "modelled"}
        # (flat copies: a reader that keeps only the scalars of `roofline` still sees them)
        for k_flat, k_src in (("valu_busy_measured", "busy_measured"), ("valu_frac_at_live_clock", "frac_at_live_clock"), ("valu_insts_per_unit", "insts_per_unit")):
            if roof["valu"].get(k_src) is not None:
                roof[k_flat] = roof["valu"][k_src]
        if busy:
            roof["valu_busy_model_low"], roof["valu_busy_model_high"] = busy["busy_frac"]
        if roof.get("clock"):
            roof["ghz_live"] = roof["clock"]["ghz_live"]
    if META.get("alg_is_upper_bound"):
        roof["algorithmic_bytes_note"] = ("an upper bound: the drift glide's slot of every voice-sample read and rewritten (8 B); voices whose glide rests or whose "
                                          "instrument has not played yet move none - the PMC traffic is the bytes that did move")
    if traffic is not None and traffic < 0.5 * alg_bytes and not META.get("alg_is_upper_bound"):
        roof["bound"] = "on-die"   # the working set never leaves the Infinity Cache: not an HBM figure
        roof["bound_evidence"] = {"hbm_traffic_over_algorithmic_bytes": traffic / alg_bytes}
    out = {
        "metric": "voice-samples/sec (SawGen->SVF chain)" if args.workload == "cfg3" else f"voice-samples/sec ({args.workload})",
        "value": value, "unit": "voice-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": desc, "voices_per_gpu": V, "total_voices": total, "vectors_per_launch": T,
                   "launches_per_step": L, "vectors_per_step": T * L, "samples_per_vector": 64,
                   "layout": "QUAD [S/4][V][4]", "parallelism": f"voices x{world} (no collective)", "launcher": rdv.kind,
                   "realtime_48k_voices": value / 48000.0,
                   **({"graph_kernel_form": ("%d voice(s) per lane, %d quad(s) per trip" % graphs[0].tuning()[1:]) + (" (tuned online)" if os.environ.get("MLGPU_BENCH_AUTOTUNE") else " (the default form)"),
                       "hiprtc": ml.jit_stats()} if graphs else {})},
        "roofline": roof,
        "ranks": ranks,
    }
    if args.sustained and world == 1:
        out["sustained"] = sustained_run(args, eng, step, launch, L, float(V) * T * 64, value)
    if args.oversubscribe:
        out["oversubscribed"] = f"{world} ranks on {len(set(buses))} GPU(s): launch-path test, not a scaling measurement"
    extras = world == 1 and args.workload == "cfg3" and not args.no_extras and not args.voices and not args.vectors
    if extras:
        # the other BASELINE configs and the north_star target, timed in this process after the headline's region (flat keys)
        del launch, step
        _release(_keep)
        t_x = time.perf_counter()
        flat, full = extra_legs(eng)
        roof.update(flat)
        out["other_configs"] = full
        out["other_configs"]["seconds"] = time.perf_counter() - t_x
    if args.workload == "cfg2":
        # each op and the fused pair, on the ramp and on noise (no data-dependent branches: same rate)
        from madronalib_amd.constants import Op
        d_x, d_y = _keep
        n_el = V * T * 64
        d_noise = eng.to_device(np.random.default_rng(1).uniform(-np.pi, np.pi, n_el).astype(np.float32))
        per_op = {}
        reps = max(16, min(256, int(2 ** 34 / (8.0 * n_el))))
        for label, op in (("sinApprox", Op.SIN_APPROX), ("expApprox", Op.EXP_APPROX), ("expApprox(sinApprox)", Op.EXP_APPROX_OF_SIN_APPROX)):
            for data, src in (("ramp", d_x), ("noise", d_noise)):
                for _ in range(8):
                    eng.op_apply(op, src, None, None, d_y, n_el)
                eng.timer_start()
                for _ in range(reps):
                    eng.op_apply(op, src, None, None, d_y, n_el)
                per_op[f"{label} / {data}"] = n_el * reps / (eng.timer_stop_ms() * 1e-3)
        out["config"]["per_op_voice_samples_per_s"] = per_op
        # BASELINE's own size (65 536 voices: 32 MiB in + out) lives in the Infinity Cache; the HBM figure of the same kernel is taken
        # at 4 194 304 voices (1 GiB) in the same run, and the line carries both
        if world == 1 and V * T * 64 * 8 < (256 << 20):
            roof["bound"] = "on-die"
            roof["bound_evidence"] = {"working_set_MiB": V * T * 64 * 8 / 2 ** 20, "infinity_cache_MiB": 256}
            # (this run launches the same kernel over 1 GiB too, below: per-kernel PMC means would mix the two sizes)
            for k in ("valu", "clock", "traffic", "pmc_case"):
                roof[k] = None
            Vb = 4194304
            nb = Vb * 64
            d_bx, d_by = eng.alloc(4 * nb), eng.alloc(4 * nb)
            eng.op_apply(Op.SIN_APPROX, d_bx, None, None, d_by, nb)   # (values irrelevant to the rate: no data-dependent branches)
            for _ in range(4):
                eng.op_apply(Op.EXP_APPROX_OF_SIN_APPROX, d_bx, None, None, d_by, nb)
            eng.sync()
            eng.timer_start()
            for _ in range(32):
                eng.op_apply(Op.EXP_APPROX_OF_SIN_APPROX, d_bx, None, None, d_by, nb)
            bms = eng.timer_stop_ms() / 32
            out["roofline_at_1GiB"] = {"bound": "hbm", "voices": Vb, "achieved": 8.0 * nb / (bms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": 8.0 * nb / (bms * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_ms": bms, "voice_samples_per_s": nb / (bms * 1e-3),
                                       "note": "the same op_kernel over 4 194 304 voices x 1 DSPVector (512 MiB in, 512 MiB out): the HBM figure of config 2"}
            d_bx.free()
            d_by.free()
    baselines = {"cfg2": cpu_baseline_cfg2, "cfg3": cpu_baseline_cfg3, "cfg4": cpu_baseline_cfg4, "cfg5": cpu_baseline_cfg5, "cfg5full": cpu_baseline_cfg5full}
    if world == 1 and not args.no_cpu_baseline and args.workload in baselines:
        try:
            out["cpu_baseline"] = baselines[args.workload]()
        except Exception as ex:  # the baseline is a report, never a reason to lose the GPU number
            out["cpu_baseline"] = {"value": None, "unit": "voice-samples/s", "cores": os.cpu_count(), "kind": "port",
                                   "sample": f"failed: {ex}"}
    if extras and not args.no_cpu_baseline:
        try:
            out["parity_512"] = parity_crcs(eng)
            for k, v in out["parity_512"].items():
                roof[f"{k}_crc_match"] = v["match"]
            roof["crc_all_match"] = bool(all(v["match"] for v in out["parity_512"].values()))
            roof["crc_cases"] = len(out["parity_512"])
        except Exception as ex:
            out["parity_512"] = {"error": repr(ex)}
    out["roofline"] = order_roofline(roof)
    return out


def run_rt_group(args, eng, info, V, T, L, rank, world, rdv):
    """--workload rt --gpus N (N > 1): the real-time block ACROSS the GPUs of a node. Every rank owns V voices of the V x N bank; per
    64 T-frame block, on the common 48 kHz deadline, it runs its voices with their sum made inside the voice kernel up to the level of
    the mixdown tree its shard is whole at (mlgpu_bank_process_mixdown_shard), copies the rows (256 bytes per row and DSPVector) to the
    host and hands them to rank 0 (madronalib_amd.sharding.RowExchange: shared memory of the node, no collective library); rank 0
    finishes the tree (mlgpu_mixdown_finish) - the channel one engine would give for all V x N voices, bit for bit
    (tests/cpp/multi_engine_test.cpp, tests/test_gpu_processbuffer.py::test_sharded_mixdown_gives_one_bank_bits). Reported: rank 0's
    time from a block's deadline to the finished channel (p50 / p99 / max: it includes waiting for the slowest rank) and the blocks
    that took longer than the period. The reference's counterpart: Synth::processVector's `outputs += voice` loop, MLSynth.h:43-57."""
    import gc
    import madronalib_amd as ml
    from madronalib_amd.constants import Proc
    from madronalib_amd.sharding import RowExchange
    total = V * world
    lo, hi = partition(total, world, rank)
    rows = ml.mixdown_shard_rows(V)
    if rows == 0:
        raise SystemExit("bench.py --workload rt --gpus N: the voices per GPU must be a multiple of 64")
    bank = eng.bank([Proc.SAW_GEN, Proc.BANDPASS, Proc.GAIN], V)
    bank.clear()
    freq, co = cfg3_params(lo, hi, total)
    for i in range(3):
        bank.set_coeff(1, i, co[i])
    bank.set_coeff(2, 0, 0.25)
    bank.set_input_const(freq)
    eng.mixdown_reserve(V, T)
    d_rows = eng.alloc(4 * rows * 64 * T)
    for _ in range(10):
        bank.process_mixdown_shard(T, d_rows)
    eng.sync()
    ex = RowExchange(rdv, rank, world, rows, T)
    blocks, lead_in = L * args.steps, max(1, L * args.warmup)
    period = 64 * T / 48000.0
    clock = time.perf_counter            # (CLOCK_MONOTONIC: the same clock in every process of the node)
    start = max(rdv.gather(clock() + 0.25 if rank == 0 else 0.0))
    us, late, peak = [], 0, 0.0
    gc_was = gc.isenabled()
    gc.disable()
    try:
        for b in range(lead_in + blocks):
            deadline = start + b * period
            while clock() < deadline:
                time.sleep(0)      # (ranks may be threads of one process: a busy wait would keep the interpreter from the others)
            bank.process_mixdown_shard(T, d_rows)
            ex.put(b, d_rows.download(np.float32, rows * 64 * T).reshape(rows, 64 * T))
            if rank == 0:
                channel = ml.mixdown_finish(ex.collect(b, T))
                d = (clock() - deadline) * 1e6
                if b >= lead_in:
                    us.append(d)
                    late += d > period * 1e6
                peak = max(peak, float(np.abs(channel).max()))
    finally:
        if gc_was:
            gc.enable()
    rdv.barrier()
    ex.close()
    ranks = rdv.gather({"rank": rank, "device": info.get("device", None), "pci_bus_id": info["pci_bus_id"], "voices": [lo, hi]})
    if rank != 0:
        return None
    us.sort()
    buses = {r["pci_bus_id"] for r in ranks}
    res = {"voices_total": total, "voices_per_gpu": V, "gpus": world, "distinct_gpus": len(buses), "frames_per_block": 64 * T, "blocks": blocks,
           "block_period_us": period * 1e6, "rows_per_gpu": rows, "bytes_exchanged_per_block": world * rows * 256 * T,
           "block_us_p50": us[len(us) // 2], "block_us_p99": us[int(len(us) * 0.99)], "block_us_max": us[-1], "late_blocks": int(late), "output_peak": peak,
           "what": "per block and rank: voices + their sum inside the voice kernel up to the shard's hand-over level (mlgpu_bank_process_mixdown_shard), D2H of "
                   "the rows; rank 0: all ranks' rows through shared memory, the tree finished on the host (mlgpu_mixdown_finish); time from the block's "
                   "deadline to the finished channel on rank 0"}
    if len(buses) != world:
        res["unmeasured_on_hardware"] = f"{world} ranks on {len(buses)} GPU(s): the launch path and the bits, not a multi-GPU measurement"
    paced_s = blocks * period
    return {"metric": "voice-samples/sec (rt: SawGen->SVF chain paced at 48 kHz, all voices of the node to one channel)", "value": float(total) * 64 * T * blocks / paced_s,
            "unit": "voice-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": paced_s / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"north_star target across {world} GPUs: {V} SawGen->Bandpass->gain voices per GPU paced at 48 kHz in {64 * T}-frame blocks, the "
                                   f"{total} voices summed to ONE channel (per-GPU tree rows + the host's finish: no collective)",
                       "voices_per_gpu": V, "total_voices": total, "vectors_per_launch": T, "launches_per_step": L, "launcher": rdv.kind,
                       "note": "a paced run delivers exactly real time unless a block is late: read rt_group.late_blocks and block_us_p99, not `value`"},
            "roofline": {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                         "rt_group_block_p50_us": res["block_us_p50"], "rt_group_block_p99_us": res["block_us_p99"], "rt_group_late_blocks": res["late_blocks"],
                         "rt_voices": total, "note": "the per-GPU voice kernel's fraction is the N = 1 run's (rt_kernel_frac in the default line)"},
            "rt_group": res, "ranks": ranks}


def run_rt(args, eng, info, V, T, L, rank, world, rdv):
    """--workload rt: the paced real-time leg as the whole run. A step = L blocks of 64 T frames, each started on its 48 kHz deadline;
    `value` is what was delivered per wall second (= V x 48 000 while no block is late), the roofline object is the voice kernel's."""
    if world > 1:
        return run_rt_group(args, eng, info, V, T, L, rank, world, rdv)
    rdv.barrier()
    t0 = time.perf_counter()
    r = rt_case(eng, V, 64 * T, blocks=L * args.steps, lead_in=max(1, L * args.warmup))
    elapsed = rdv.max(time.perf_counter() - t0)
    if rank != 0:
        return None
    paced_s = L * args.steps * r["block_period_us"] * 1e-6
    return {"metric": "voice-samples/sec (rt: SawGen->SVF chain paced at 48 kHz)", "value": float(V) * world * 64 * T * L * args.steps / paced_s, "unit": "voice-samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": paced_s / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"north_star target: {V} SawGen->Bandpass->gain voices per GPU paced at 48 kHz in {64 * T}-frame blocks through "
                                   "mlgpu_process_buffer_process (voices + mixdown -> D2H), host waiting out each block period",
                       "voices_per_gpu": V, "total_voices": V * world, "vectors_per_launch": T, "launches_per_step": L, "wall_s_incl_setup": elapsed,
                       "note": "a paced run delivers exactly real time unless a block is late: read rt.misses and rt.call_us_p99, not `value`"},
            "roofline": {"bound": "hbm", "achieved": r["algorithmic_bytes_per_block"] / (r["voice_kernel_us_free_running"] * 1e-6) / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": r["kernel_frac"], "traffic": None, "kernel": "chain_kernel<SawGen, Bandpass, Gain>",
                         "kernel_ms": r["voice_kernel_us_free_running"] * 1e-3, "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_block"],
                         "rt_voices": r["voices"], "rt_block_p50_us": r["call_us_p50"], "rt_block_p99_us": r["call_us_p99"], "rt_misses": r["misses"],
                         "rt_kernel_frac": r["kernel_frac"], "rt_two_calls_block_p50_us": r["two_calls_us_p50"]},
            "rt": r}


def sustained_run(args, eng, step, launch, L, units_per_launch, short_value):
    """Is the number sustained? Run the timed region's steps back to back for --sustained-seconds, one HIP-event bracket per
    step (a step is L launches, ~10 ms: the sync between steps is < 0.3 % of it), then time 400 single launches. Reports the
    rate of every wall second, the step-time distribution and the droop against the short timed region."""
    def dist(xs):
        xs = sorted(xs)
        pick = lambda q: xs[min(len(xs) - 1, int(q * len(xs)))]  # noqa: E731
        return {"n": len(xs), "min": xs[0], "p01": pick(0.01), "median": pick(0.5), "p99": pick(0.99), "max": xs[-1],
                "mean": sum(xs) / len(xs)}
    step_ms, t_end = [], time.perf_counter() + args.sustained_seconds
    wall0 = time.perf_counter()
    marks = []
    while time.perf_counter() < t_end:
        eng.timer_start()
        step()
        step_ms.append(eng.timer_stop_ms())
        marks.append(time.perf_counter() - wall0)
    per_second, k0 = [], 0
    for sec in range(1, int(marks[-1]) + 1):
        k1 = k0
        while k1 < len(marks) and marks[k1] <= sec:
            k1 += 1
        if k1 > k0:
            per_second.append(units_per_launch * L * (k1 - k0) / (sum(step_ms[k0:k1]) * 1e-3))
        k0 = k1
    launch_ms = []
    for _ in range(400):
        eng.timer_start()
        launch()
        launch_ms.append(eng.timer_stop_ms())
    device_rate = units_per_launch * L * len(step_ms) / (sum(step_ms) * 1e-3)
    res = {"seconds": marks[-1], "steps": len(step_ms), "launches_per_step": L,
           "rate_device_time": device_rate, "rate_wall": units_per_launch * L * len(step_ms) / marks[-1],
           "short_region_rate": short_value, "droop_vs_short_region": 1.0 - device_rate / short_value,
           "per_second_rate": per_second, "per_second_min_over_max": min(per_second) / max(per_second) if per_second else None,
           "step_ms": dist(step_ms), "single_launch_ms": dist(launch_ms),
           "note": "rates from HIP events around each step (device time); rate_wall includes the host's sync between steps"}
    with open(args.sustained, "w") as f:
        json.dump(res, f, indent=1)
    return {k: res[k] for k in ("seconds", "rate_device_time", "droop_vs_short_region", "per_second_min_over_max", "step_ms")}


def launch_processes(args, argv):
    """`python bench.py --gpus N` with no launcher around it: start one rank process per GPU ourselves (rank r on device r),
    lined up through a fresh rendezvous directory; rank 0's JSON line is this process's output. A rank that dies - however it
    dies - releases the others (madronalib_amd.rendezvous.run_rank_processes)."""
    from madronalib_amd.rendezvous import run_rank_processes
    rcs, out0 = run_rank_processes([sys.executable, os.path.abspath(__file__)] + argv, args.gpus)
    if any(rcs):
        raise SystemExit(f"bench.py: rank exit codes {rcs}")
    sys.stdout.write(out0)
    sys.stdout.flush()


def launch_threads(args, have):
    """One process, one host thread + engine + stream per device. ctypes releases the GIL inside every C-ABI
    call, and a launch is an asynchronous enqueue, so the threads only contend for microseconds per launch."""
    import threading
    from madronalib_amd.rendezvous import ThreadRendezvous
    group = ThreadRendezvous.group(args.gpus)
    results, errors = [None] * args.gpus, []

    def body(r):
        try:
            results[r] = run_rank(args, r, r % have, args.gpus, group[r])
        except BaseException as ex:   # noqa: BLE001 - a dead rank must release the others
            errors.append((r, ex))
            group[r].abort()
    threads = [threading.Thread(target=body, args=(r,), name=f"mlgpu-rank{r}") for r in range(args.gpus)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise SystemExit(f"bench.py: rank {errors[0][0]} failed: {errors[0][1]!r}")
    print(json.dumps(results[0]), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="cfg3", choices=list(WORKLOADS))
    ap.add_argument("--voices", type=int, default=0, help="voices per GPU (default: the config's)")
    ap.add_argument("--vectors", type=int, default=0, help="DSPVectors per launch (default: the config's)")
    ap.add_argument("--launches", type=int, default=0, help="launches per step (default: the config's)")
    ap.add_argument("--launcher", default="processes", choices=["processes", "threads"],
                    help="how `--gpus N` starts its N ranks when no launcher exported WORLD_SIZE: one process per GPU (default) "
                         "or one host thread per GPU inside this process")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="TEST ONLY: let ranks share devices (rank r on device r mod visible) so the N>1 launch paths can be exercised on "
                         "a box with fewer GPUs; the line says so and is not a scaling measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-counters", action="store_true", help="do not take the headline's FETCH_SIZE / WRITE_SIZE with rocprofv3 in child processes")
    ap.add_argument("--no-extras", action="store_true",
                    help="the default run (cfg3, one GPU) also times configs 2, 4, 5, the survey's patch and the paced 2^20-voice target and adds them "
                         "as flat keys under roofline (~15 s); this switch leaves the headline alone")
    ap.add_argument("--cascade-lanes", type=int, default=None, choices=[-1, 0, 1, 2, 4],
                    help="force the form of SVF-cascade banks (mlgpu_engine_set_cascade_lanes): A/B runs of config 4")
    ap.add_argument("--two-streams", action="store_true",
                    help="synth: EventsToSignals on an engine (HIP stream) of its own, its kernel for block k + 1 under the voice kernel of block k")
    ap.add_argument("--strict-svf", action="store_true",
                    help="mlgpu_engine_set_strict_svf: SVF memories updated with two instructions instead of one fused (generated kernels)")
    ap.add_argument("--sustained", metavar="FILE", default=None,
                    help="after the timed region, run the same steps for --sustained-seconds more and write per-second rates and the "
                         "distribution of step and launch times to FILE (profiles/archive/r03_cfg3_sustained.json)")
    ap.add_argument("--sustained-seconds", type=float, default=30.0)
    ap.add_argument("--print-case", action="store_true", help="print the key of this case in profiles/pmc_workloads.json and exit")
    args = ap.parse_args()
    if args.workload == "strings":
        # per-voice delay times: the transposed ring layout is the one a host should choose (round 5; 0 and 1 for comparison)
        os.environ.setdefault("MLGPU_DELAY_WINDOWS", "2")
    if args.print_case:
        dV, dT, _ = WORKLOADS[args.workload]
        print(workload_key(args.workload, args.voices or dV, args.vectors or dT))
        return
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")

    from madronalib_amd import _lib, rendezvous
    have = _lib.load().mlgpu_device_count()
    if have <= 0:
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")

    if "WORLD_SIZE" in os.environ:
        # a rank of a launched job: torch.distributed.run (the driver's form for N > 1) or our own process launcher
        world = int(os.environ["WORLD_SIZE"])
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
        if world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
        if args.oversubscribe:
            local_rank %= have
        isolated = any(os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "GPU_DEVICE_ORDINAL"))
        if local_rank >= have and have == 1 and isolated:
            local_rank = 0   # the launcher gave every rank its own device through a *_VISIBLE_DEVICES mask (checked below by PCI bus id)
        if local_rank >= have:
            raise SystemExit(f"bench.py: rank {rank} wants device {local_rank} but only {have} GPU(s) are visible")
        rdv = rendezvous.from_environment()
        try:
            out = run_rank(args, rank, local_rank, world, rdv)
        except BaseException:
            if hasattr(rdv, "abort"):
                rdv.abort()
            raise
        if out is not None:
            print(json.dumps(out), flush=True)
        rdv.close()
        return
    if args.gpus > have and not args.oversubscribe:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible: refusing to run fewer ranks than asked for")
    if args.gpus == 1:
        print(json.dumps(run_rank(args, 0, 0, 1, rendezvous.SoloRendezvous())), flush=True)
    elif args.launcher == "threads":
        launch_threads(args, have)
    else:
        launch_processes(args, sys.argv[1:])


if __name__ == "__main__":
    main()
