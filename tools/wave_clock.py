#!/usr/bin/env python3
"""Developer tool (GPU): when does each wavefront of a fused graph kernel start and end? MLGPU_GRAPH_WAVE_CLOCK makes the generated
kernel stamp s_memrealtime (100 MHz) at entry and exit of every wavefront; this runs a workload's graph a few times, reads the last
launch's table and prints the distribution: launch span, per-wavefront life, when the first / median / last wavefront ends, and the
same per XCD.   python tools/wave_clock.py cfg5|cfg5full [voices]"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
path = os.path.join(tempfile.gettempdir(), "mlgpu_wave_clock.bin")
os.environ["MLGPU_GRAPH_WAVE_CLOCK"] = path
import madronalib_amd as ml  # noqa: E402
from madronalib_amd import patches  # noqa: E402
from madronalib_amd.sharding import cfg5_gate_quad, cfg5_voice_params  # noqa: E402

w = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
V = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
T = 16
full = w == "cfg5full"
eng = ml.Engine(0)
n = V * T * 64
d_gate = eng.to_device(cfg5_gate_quad(0, V, T))
d_out = eng.alloc(4 * n)
desc, outs = patches.synth16(full=full)
g = ml.Graph(eng, V, desc, outs, voices_per_lane=1)
g.clear()
params, coeffs, seeds = cfg5_voice_params(0, V, V, ml, full=full)
for k, x in params.items():
    g.set_param(k, x if np.ndim(x) else float(x))
for k, c in coeffs.items():
    g.set_coeffs(k, [np.ascontiguousarray(r) for r in c])
g.set_state("noise", 0, seeds)
for _ in range(30):
    g.process(T, [d_gate], [d_out])
eng.sync()
eng.timer_start()
g.process(T, [d_gate], [d_out])
ms = eng.timer_stop_ms()
g.close()
t = np.fromfile(path, dtype=np.uint64).reshape(-1, 4)
t0, t1 = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)
xcc = (t[:, 3] & 0xF).astype(int)
hw = t[:, 2]
cu, se, simd, wave_slot = (hw >> 8) & 0xF, (hw >> 13) & 0x7, (hw >> 4) & 0x3, hw & 0xF
base = t0.min()
us = lambda x: (x - base) / 100.0
life = (t1 - t0) / 100.0
print(f"{w}: {len(t)} wavefronts, launch {ms * 1000:.1f} us by HIP events; first start 0, last start {us(t0.max()):.1f} us, first end {us(t1.min()):.1f} us, "
      f"median end {us(np.median(t1)):.1f} us, last end {us(t1.max()):.1f} us")
print(f"  wavefront life: min {life.min():.1f}  p10 {np.percentile(life, 10):.1f}  median {np.median(life):.1f}  p90 {np.percentile(life, 90):.1f}  max {life.max():.1f} us"
      f"   mean life / launch span = {life.mean() / us(t1.max()):.3f}")
for x in sorted(set(xcc)):
    m = xcc == x
    print(f"  XCD {x}: {m.sum():5d} wavefronts  starts {us(t0[m].min()):7.1f}..{us(t0[m].max()):7.1f}  ends {us(t1[m].min()):7.1f}..{us(t1[m].max()):7.1f}  median life {np.median(life[m]):7.1f}")
# how many wavefronts are resident over time (per SIMD there is room for 4 of this kernel)
grid = np.linspace(0, us(t1.max()), 21)
res = [int(((us(t0) <= x) & (us(t1) > x)).sum()) for x in grid]
print("  resident wavefronts at 5 % steps of the span:", res)
slots = len(set(zip(xcc.tolist(), se.tolist(), cu.tolist(), simd.tolist())))
sid = (((xcc * 8 + se.astype(int)) * 16 + cu.astype(int)) * 4 + simd.astype(int))
print(f"  distinct (XCD, SE, CU, SIMD) seen: {slots}; wavefronts per SIMD: min {np.bincount(sid).min()} max {np.bincount(sid).max()}; hardware wave slots used: {sorted(set(wave_slot.astype(int).tolist()))}")
# by voice block: does a wavefront's life depend on where its voices are?
k = len(t) // 8
print("  median life by eighth of the voice range:", [round(float(np.median(life[i * k:(i + 1) * k])), 1) for i in range(8)])
# where in the chip: is a SIMD index, a CU or a shader engine systematically slower? (mean life, us; spread of the per-SIMD means)
print("  mean life by SIMD index:", [round(float(life[simd == s].mean()), 1) for s in range(4)],
      " by wave slot:", [round(float(life[wave_slot == s].mean()), 1) for s in sorted(set(wave_slot.astype(int).tolist()))])
print("  mean life by SE:", [round(float(life[se == s].mean()), 1) for s in sorted(set(se.astype(int).tolist()))])
print("  mean life by CU index:", [round(float(life[cu == c].mean()), 1) for c in sorted(set(cu.astype(int).tolist()))])
per_simd_last = np.zeros(sid.max() + 1)
np.maximum.at(per_simd_last, sid, us(t1))
per_simd_first = np.full(sid.max() + 1, 1e30)
np.minimum.at(per_simd_first, sid, us(t1))
used = np.bincount(sid, minlength=sid.max() + 1) > 0
print(f"  per SIMD: last end min {per_simd_last[used].min():.1f} median {np.median(per_simd_last[used]):.1f} max {per_simd_last[used].max():.1f};"
      f" first end min {per_simd_first[used].min():.1f} median {np.median(per_simd_first[used]):.1f}; "
      f"mean (last - first end) within a SIMD {np.mean(per_simd_last[used] - per_simd_first[used]):.1f} us")
# within a CU: do its four SIMDs end together?
cid = sid // 4
per_cu_last = np.zeros(cid.max() + 1); np.maximum.at(per_cu_last, cid, us(t1))
per_cu_first_simd_last = np.full(cid.max() + 1, 1e30); np.minimum.at(per_cu_first_simd_last, sid // 4, per_simd_last[sid])
usedc = np.bincount(cid, minlength=cid.max() + 1) > 0
print(f"  per CU: last end min {per_cu_last[usedc].min():.1f} median {np.median(per_cu_last[usedc]):.1f} max {per_cu_last[usedc].max():.1f}; "
      f"spread of its SIMDs' last ends (mean) {np.mean(per_cu_last[usedc] - per_cu_first_simd_last[usedc]):.1f} us")
