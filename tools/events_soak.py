#!/usr/bin/env python3
"""EventsToSignals on the device against the reference's own class (oracle/_ref/libdropin_ref.so), random configurations: polyphony
1 .. 16, MIDI / MPE / unison / sustain-pedal performances (tests/test_gpu_events.py: performance), sample rates 8 k .. 192 k, glide and
drift settings, block sizes of 1 .. 16 DSPVectors processed in launches of 1 .. 16, 2 .. 6 instruments per bank, the voice rows AND the
smoothed controller signals - every row of every voice bit for bit. The GPU tests run ten fixed scenarios; this adds volume.
    python tools/events_soak.py [cases] [seed]"""
import os
import sys
import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import madronalib_amd as ml          # noqa: E402
import test_gpu_events as te         # noqa: E402


def dense_performance(rng, frames, polyphony, mpe, sustain):
    """Events in BURSTS - several inside one DSPVector, on its first and last frames, on the same frame - for every protocol, with pedal events
    in all of them: what the scripted performances of the tests (an event every ~130 frames) reach only by accident."""
    evs, held, t = [], [], int(rng.integers(0, 100))
    steps = [0, 0, 1, 2, 3, 7, 17, 31, 62, 63, 64, 65, 66, 127, 128, 129, 300, 700]
    while t < frames - 2:
        r = rng.random()
        chan = int(rng.integers(2, 2 + polyphony + 2)) if mpe else 1
        if r < 0.34 or not held:
            key = int(rng.integers(30, 90))
            evs.append((te.NOTE_ON, chan, key, t, float(np.float32((key - 60) / 12.0)), float(np.float32(rng.uniform(0.1, 1.0)))))
            held.append((chan, key))
        elif r < 0.62:
            c, k = held.pop(int(rng.integers(0, len(held))))
            evs.append((te.NOTE_OFF, c, k, t, 0.0, 0.0))
        elif r < 0.68:
            evs.append((te.BEND, chan if rng.random() < 0.7 else 1, 0, t, float(np.float32(rng.uniform(-1, 1))), 0.0))
        elif r < 0.74:
            evs.append((te.CTRL, chan, int(rng.choice([16, 73, 74, 1, 128])), t, float(np.float32(rng.random())), 0.0))
        elif r < 0.80:
            if mpe or rng.random() < 0.5:
                evs.append((te.CHAN_PRESS, chan if rng.random() < 0.8 else 1, 0, t, float(np.float32(rng.random())), 0.0))
            elif held:
                evs.append((te.NOTE_PRESS, 1, held[-1][1], t, float(np.float32(rng.random())), 0.0))
        elif r < 0.93 and sustain:
            evs.append((te.SUSTAIN, 1, 0, t, float(rng.integers(0, 2)), 0.0))
        elif r < 0.95:
            evs.append((te.CTRL, 1, 123, t, 0.0, 0.0))        # all notes off
            held = []
        if rng.random() < 0.5:
            t += int(rng.choice(steps))
        else:
            t = (t // 64 + int(rng.integers(0, 3))) * 64 + int(rng.choice([0, 0, 1, 62, 63]))   # a vector's edges
    evs.sort(key=lambda e: e[3])   # (stable: events of one frame keep their order of arrival)
    return [e for e in evs if e[3] < frames]


ONLY = int(os.environ.get("MLGPU_SOAK_ONLY", "-1"))   # this configuration alone (the others' random draws are still made), with where it differs


DENSE = bool(os.environ.get("MLGPU_SOAK_DENSE"))       # bursts of events (dense_performance) instead of the tests' scripted performances


def run(cases, seed, eng=None):
    rng = np.random.default_rng(seed)
    eng = eng or ml.Engine(0)
    bad, rows_checked = 0, 0
    for case in range(cases):
        mode = ["midi", "midi", "mpe", "sustain", "unison"][int(rng.integers(0, 5))]
        P = int(rng.integers(1, 17))
        cfg = dict(polyphony=P, sr=float(rng.choice([8000.0, 22050.0, 44100.0, 48000.0, 96000.0, 192000.0])), glide=float(rng.choice([0.0, 0.002, 0.01, 0.05])),
                   drift=float(rng.choice([0.0, 0.3, 1.0])), bend=float(rng.choice([2.0, 7.0, 12.0])), mpe_bend=float(rng.choice([24.0, 48.0])), mod_cc=int(rng.choice([1, 16])))
        if mode == "mpe":
            cfg["mpe"] = 1
        if mode == "unison":
            cfg["unison"] = 1
        block = 64 * int(rng.integers(1, 17))
        n_blocks = int(rng.integers(3, 10))
        vpl = int(rng.integers(1, 17))
        N = int(rng.integers(2, 7))
        kind = "mpe" if mode == "mpe" else ("sustain" if mode == "sustain" else "midi")
        if DENSE:
            inst = [dense_performance(rng, block * n_blocks, P, mode == "mpe", rng.random() < 0.6) for _ in range(N)]
            if rng.random() < 0.3:
                cfg["unison"] = 1
        else:
            inst = [te.performance(kind, int(rng.integers(1 << 30)), block * n_blocks, P) for _ in range(N)]
        watch = [int(c) for c in rng.choice([1, 16, 73, 74, 7, 11], size=int(rng.integers(1, 4)), replace=False)] if rng.random() < 0.5 else None
        if ONLY >= 0 and case != ONLY:
            continue
        try:
            got = te.gpu_run(eng, cfg, inst, block, n_blocks, vectors_per_launch=vpl, watch=watch)
        except ml.MlgpuError as e:
            print(f"case {case}: {cfg} block {block} x {n_blocks} launches of {vpl}: {e}")
            bad += 1
            continue
        ctl = None
        if watch is not None:
            got, ctl = got
        diff = 0
        for k, evs in enumerate(inst):
            if watch is not None:
                want, wctl = te.ref_run_controllers(cfg, evs, block, n_blocks, watch)
                diff += int((ctl[:, k].view(np.uint32) != wctl.view(np.uint32)).sum())
            else:
                want = te.ref_run(cfg, evs, block, n_blocks)
            g = got[:, k * P:(k + 1) * P]
            nan = np.isnan(g) & np.isnan(want)
            dmask = (g.view(np.uint32) != want.view(np.uint32)) & ~nan
            diff += int(dmask.sum())
            if ONLY >= 0 and dmask.any():
                w = np.argwhere(dmask)
                r0, v0, t0 = w[np.argmin(w[:, 2])].tolist()
                print(f"  instrument {k}: {int(dmask.sum())} words; rows {sorted(set(w[:, 0].tolist()))} voices {sorted(set(w[:, 1].tolist()))}; first at frame {t0} (block {t0 // block}, frame {t0 % block} in it): row {te.ROW_NAMES[r0]} voice {v0} device {g[r0, v0, t0]!r} reference {want[r0, v0, t0]!r}")
                print("   events around it:", [e for e in evs if t0 - 1200 <= e[3] <= t0 + 400])
                for r in sorted(set(w[:, 0].tolist())):
                    ts = sorted(set([max(0, t0 - 40), t0 - 24, t0 - 23, t0 - 22, t0 - 21, t0 - 2, t0 - 1, t0, t0 + 1, t0 + 21, t0 + 22, t0 + 64]))
                    ts = [t for t in ts if 0 <= t < g.shape[2]]
                    print(f"   row {te.ROW_NAMES[r]} voice {v0} at {ts}: device {[float(g[r, v0, t]) for t in ts]} reference {[float(want[r, v0, t]) for t in ts]}")
                for r in sorted(set(w[:, 0].tolist())):
                    ww = w[w[:, 0] == r]
                    print(f"   row {te.ROW_NAMES[r]}: frames {ww[:, 2].min()} .. {ww[:, 2].max()}, {len(ww)} words")
            rows_checked += 8 * P
        if diff:
            bad += 1
            print(f"case {case}: mode {mode} {cfg} block {block} x {n_blocks}, launches of {vpl} vectors, {N} instruments, watch {watch}: {diff} words differ")
    print(f"{cases} configurations (seed {seed}), {rows_checked} voice rows against the reference class: {bad} with a difference")
    return bad


def run_fused(cases, seed, eng=None):
    """The same random configurations (MIDI, unison, sustain), the pitch and gate rows made INSIDE the voice graph's kernel (e2s_ctl_kernel's
    control records + mlev::CtlVoice) against e2s_kernel writing the rows and the graph reading them: the 16-node synth voice's audio -
    every other case summed per instrument inside the kernel -, bit for bit."""
    rng = np.random.default_rng(seed)
    eng = eng or ml.Engine(0)
    bad = 0
    for case in range(cases):
        mode = ["midi", "midi", "sustain", "unison"][int(rng.integers(0, 4))]
        P = int(rng.choice([1, 2, 3, 4, 8, 16]))
        cfg = dict(polyphony=P, sr=float(rng.choice([8000.0, 44100.0, 48000.0, 192000.0])), glide=float(rng.choice([0.0, 0.002, 0.01, 0.05])),
                   drift=float(rng.choice([0.0, 0.3, 1.0])), bend=float(rng.choice([2.0, 7.0])), mod_cc=int(rng.choice([1, 16])))
        if mode == "unison":
            cfg["unison"] = 1
        block = 64 * int(rng.integers(1, 13))
        n_blocks = int(rng.integers(3, 8))
        vpl = int(rng.integers(1, 13))
        N = 64 // P * int(rng.integers(1, 3)) if rng.random() < 0.7 else int(rng.integers(2, 7))
        voice_sum = bool(case % 2) and (N * P) % 64 == 0 and P in (2, 4, 8, 16)
        if DENSE:
            inst = [dense_performance(rng, block * n_blocks, P, False, mode == "sustain" or rng.random() < 0.3) for _ in range(N)]
        else:
            inst = [te.performance("sustain" if mode == "sustain" else "midi", int(rng.integers(1 << 30)), block * n_blocks, P) for _ in range(N)]
        if ONLY >= 0 and case != ONLY:
            continue
        try:
            a, b = te._two_kernel_and_fused(eng, cfg, inst, block, n_blocks, vectors_per_launch=vpl, voice_sum=voice_sum)
        except ml.MlgpuError as e:
            print(f"case {case}: {cfg} {N} instruments, block {block} x {n_blocks}, launches of {vpl}, voice sum {voice_sum}: {e}")
            bad += 1
            continue
        nan = np.isnan(a) & np.isnan(b)
        diff = int(((a.view(np.uint32) != b.view(np.uint32)) & ~nan).sum())
        if diff:
            bad += 1
            w = np.argwhere((a.view(np.uint32) != b.view(np.uint32)) & ~nan)
            print(f"case {case}: mode {mode} {cfg} {N} instruments, block {block} x {n_blocks}, launches of {vpl}, voice sum {voice_sum}: {diff} words differ, first [row, frame] {w[:3].tolist()}")
    print(f"{cases} configurations (seed {seed}), event rows inside the voice kernel against two kernels: {bad} with a difference")
    return bad


if __name__ == "__main__":
    if os.environ.get("MLGPU_SOAK_FUSED"):
        sys.exit(1 if run_fused(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
