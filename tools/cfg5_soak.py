#!/usr/bin/env python3
"""Developer tool (GPU + the compiled reference as the checker): BASELINE configs[4] as bench.py runs it, EVERY voice, over many
launches with carried state - the outputs of chosen launches (the last one among them) compared bit for bit with the same voice
written with the reference's own objects and run from the start on the host threads, slab of voices by slab. The GPU tests check
every voice on launches 0 and 1 and a strided subset later; this is the long run behind them.

    python tools/cfg5_soak.py cfg5|cfg5full [launches=32] [seconds=300]

Prints one line per slab and a total; exit status 1 on any differing word. The host side is what takes the time (the reference runs
launches x 16 DSPVectors for every voice); `seconds` bounds it - the slabs not reached are reported as such."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import madronalib_amd as ml  # noqa: E402
from madronalib_amd import patches  # noqa: E402
from madronalib_amd.sharding import cfg5_gate_quad, cfg5_voice_params  # noqa: E402
from cpu_checkers import fast_checker, host_threads  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 300.0
full = which == "cfg5full"
V, T = 262144, 16
keep = sorted({L // 4 - 1, L // 2 - 1, L - 1} - {-1})
fast = fast_checker()
if fast is None:
    raise SystemExit("the compiled reference (oracle/_ref) is not built")

eng = ml.Engine(0)
desc, outs = patches.synth16(full=full)
g = ml.Graph(eng, V, desc, outs)
g.clear()
params, coeffs, seeds = cfg5_voice_params(0, V, V, ml, full=full)
for k, v in params.items():
    g.set_param(k, v if np.ndim(v) else float(v))
for k, c in coeffs.items():
    g.set_coeffs(k, [np.ascontiguousarray(r) for r in c])
g.set_state("noise", 0, seeds)
gate_q = cfg5_gate_quad(0, V, T)
d_gate = eng.to_device(gate_q)
d_out = eng.alloc(4 * V * T * 64)
got = {}
t0 = time.time()
for launch in range(L):
    g.process(T, [d_gate], [d_out])
    if launch in keep:
        got[launch] = d_out.download(np.float32).reshape(T * 16, V, 4)
eng.sync()
print(f"{which}: {L} launches of {T} DSPVectors x {V} voices on the GPU in {time.time() - t0:.1f} s (with {len(keep)} downloads); kept launches {keep}", flush=True)
g.close()

run = fast.synth16full_run if full else fast.synth16_run
threads = host_threads()
slab = max(1024, min(8192, 8192 * 32 // L))   # (the reference's output for a slab: slab x launches x 4 KiB)
slab = 1 << (slab.bit_length() - 1)            # a divisor of the bank's voices (a power of two): 40 launches gave 6 553 and a ragged last slab
bad = done = 0
t0 = time.time()
for a in range(0, V, slab):
    if time.time() - t0 > budget:
        break
    b = a + slab
    gate = np.ascontiguousarray(gate_q[:, a:b, :].transpose(1, 0, 2).reshape(slab, T * 64))
    p = {k: (np.asarray(v)[a:b] if np.ndim(v) else v) for k, v in params.items()}
    c = {k: np.ascontiguousarray(np.asarray(cc)[:, a:b]) for k, cc in coeffs.items()}
    want = run(p, c, seeds[a:b], np.tile(gate, (1, L)), threads)[0].reshape(slab, L, T * 64)
    words = 0
    for launch in keep:
        gq = got[launch][:, a:b, :].transpose(1, 0, 2).reshape(slab, T * 64)
        words += int((gq.view(np.uint32) != want[:, launch].view(np.uint32)).sum())
    bad += words
    done = b
    print(f"  voices {a:6d}..{b - 1:6d}: {words} words differ in launches {keep}   ({time.time() - t0:.0f} s)", flush=True)
print(f"{which}: {done} of {V} voices checked over {L} launches ({L * T * 64} samples each, {threads} host threads, {time.time() - t0:.0f} s): "
      f"{bad} words differ" + ("" if done == V else f"; {V - done} voices not reached within {budget:.0f} s"))
sys.exit(1 if bad else 0)
