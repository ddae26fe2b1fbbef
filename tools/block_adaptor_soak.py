#!/usr/bin/env python3
"""mlgpu_process_buffer (the engine's SignalProcessBuffer) against the reference's own SignalProcessBuffer (oracle/_ref/libdropin_ref.so: spb_ref_run)
on random sequences of host block sizes (1 .. max_frames frames, max_frames 64 .. 1024, any value), both outputs bit for bit - including the
sequences that overdrive the reference's rings (its own glitch, which the synchronous mode reproduces) - and, where the rings are not overdriven,
the pipelined mode against the synchronous one delayed by its latency.     python tools/block_adaptor_soak.py [cases] [first seed]"""
import os
import sys
import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import madronalib_amd as ml                                # noqa: E402
from madronalib_amd.constants import Layout, Op, Proc      # noqa: E402
from inputs import lcg_noise                               # noqa: E402
import test_gpu_processbuffer as tp                        # noqa: E402

DESC = [dict(name="x", type="input"), dict(name="half", type="const", value=0.5),
        dict(name="lp", type="proc", kind=Proc.LOPASS, inputs=["x"]), dict(name="y1", type="op", kind=Op.MULTIPLY, inputs=["x", "half"])]


def device(eng, max_frames, blocks, x, pipelined):
    g = ml.Graph(eng, 1, DESC, ["lp", "y1"])
    g.set_coeffs("lp", ml.Lopass.makeCoeffs(0.05, 0.9))
    pb = ml.ProcessBuffer(eng, 1, 2, max_frames)
    if pipelined:
        pb.set_pipelined(True)

    def fn(n_vectors, d_in, d_out):
        g.process(n_vectors, d_in, d_out, Layout.VOICE_MAJOR, Layout.VOICE_MAJOR)
    outs, pos = [[], []], 0
    for b in blocks:
        o = pb.process([x[pos:pos + b]], b, fn)
        outs[0].append(o[0]), outs[1].append(o[1])
        pos += b
    lat = pb.latency_frames()
    pb.close()
    g.close()
    return np.concatenate(outs[0]), np.concatenate(outs[1]), lat


def run(cases, first, eng=None):
    eng = eng or ml.Engine(0)
    bad, over = 0, 0
    for seed in range(first, first + cases):
        rng = np.random.default_rng(70000 + seed)
        max_frames = int(rng.choice([64, 100, 128, 200, 256, 500, 512, 777, 1024])) if rng.random() < 0.8 else int(rng.integers(64, 1025))
        blocks = []
        for _ in range(int(rng.integers(6, 30))):
            r = rng.random()
            blocks.append(max_frames if r < 0.15 else (64 * int(rng.integers(1, max(2, max_frames // 64 + 1))) if r < 0.4 else int(rng.integers(1, max_frames + 1))))
        blocks = [min(b, max_frames) for b in blocks]
        total = sum(blocks)
        x = lcg_noise(np.array([seed], np.uint32), total)[0]
        want0, want1 = tp._spb_ref(max_frames, blocks, x)
        s0, s1, _ = device(eng, max_frames, blocks, x, False)
        d = int((s0.view(np.uint32) != want0.view(np.uint32)).sum()) + int((s1.view(np.uint32) != want1.view(np.uint32)).sum())
        what = "synchronous mode against the reference"
        if not d and not tp._overdrives_the_reference_rings(max_frames, blocks):
            p0, p1, lat = device(eng, max_frames, blocks, x, True)
            n = total - lat
            if n > 0:
                d = int((p0[lat:].view(np.uint32) != s0[:n].view(np.uint32)).sum()) + int((p1[lat:].view(np.uint32) != s1[:n].view(np.uint32)).sum()) + int((p0[:lat] != 0).sum())
                what = "pipelined mode against the synchronous one"
        else:
            over += 1
        if d:
            bad += 1
            print(f"seed {seed}: max_frames {max_frames} blocks {blocks}: {d} words differ ({what})")
    print(f"{cases} block sequences (seeds {first} .. {first + cases - 1}; {over} of them overdrive the reference's rings): {bad} with a difference")
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
