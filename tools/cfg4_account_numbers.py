#!/usr/bin/env python3
"""The tables of profiles/archive/r03_cfg4_account.md from the files a profile call left (profiles/archive/r03_cascade_lanes.txt and the two lab PMC
files): cycles with and without the HBM streams, the forms at 131 072 channels, the forms at other bank sizes. Markdown on stdout."""
import os
import re
import sys

D = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
txt = open(os.path.join(D, "r03_cascade_lanes.txt")).read()
parts = re.split(r"# mode (\d)[^\n]*\n", txt)
blocks = {}
for i in range(1, len(parts), 2):
    mode, body = int(parts[i]), parts[i + 1]
    V = int(re.search(r"# (\d+) channels", body).group(1))
    rows = {}
    for line in body.splitlines():
        m = re.match(r"(product: 1 lane/channel|lanes LPC=(\d) R=(\d+) w(\d))\s+min [\d.]+ ms.*median ([\d.]+) ms \(\d+ GB/s = ([\d.]+) of", line)
        if m:
            key = "r2" if m.group(1).startswith("product") else f"L{m.group(2)}R{m.group(3)}"
            rows[key] = (float(m.group(5)), float(m.group(6)))
    blocks.setdefault((V, mode), rows)


def pmc(fname, kernel):
    t = open(os.path.join(D, fname)).read()
    seg = t[t.index(kernel):][:1200]
    cyc = float(re.search(r"GRBM_GUI_ACTIVE\s+mean\s+([\d.]+)", seg).group(1)) / 8
    valu = float(re.search(r"SQ_INSTS_VALU\s+mean\s+([\d.]+)", seg).group(1))
    return cyc, valu


K = {"r2": "cascade_kernel<mldev::Chain<>, 16, 8, true>", "L1R8": "cascade_lanes_kernel<16, 8, 1, 8, 2, true>",
     "L2R8": "cascade_lanes_kernel<16, 8, 2, 8, 4, true>", "L4R8": "cascade_lanes_kernel<16, 8, 4, 8, 6, true>"}
a, b = blocks[(131072, 0)], blocks[(131072, 1)]
print("| kernel | cycles, real streams | cycles, no HBM | time, real streams | time, no HBM |\n|---|---|---|---|---|")
for k, name in (("r2", "round 2 `cascade_kernel`"), ("L1R8", "round 3 `cascade_lanes_kernel<…, 1, 8, 2>`")):
    cs, _ = pmc("r03_cascade_lab_pmc_streams.txt", K[k])
    cn, _ = pmc("r03_cascade_lab_pmc_no_hbm.txt", K[k])
    sp = lambda x: f"{x:,.0f}".replace(",", " ")  # noqa: E731
    print(f"| {name} | {sp(cs)} | {sp(cn)} ({(cn / cs - 1) * 100:+.1f} %) | {a[k][0]:.3f} ms | {b[k][0]:.3f} ms ({(b[k][0] / a[k][0] - 1) * 100:+.0f} %) |")
print("\ninput only / output only streamed (modes 2, 3), one lane: %.3f / %.3f ms" % (blocks[(131072, 2)]["L1R8"][0], blocks[(131072, 3)]["L1R8"][0]))
print("\n| form (131 072 channels × 32 DSPVectors) | VALU instr / launch | cycles | ms | of 8 TB/s |\n|---|---|---|---|---|")
for k, name in (("r2", "round 2, one lane per channel"), ("L1R8", "round 3, one lane per channel"), ("L2R8", "round 3, two lanes per channel (DPP)"),
                ("L4R8", "round 3, four lanes per channel (DPP)")):
    cs, valu = pmc("r03_cascade_lab_pmc_streams.txt", K[k])
    print(f"| {name} | {valu / 1e6:.1f} M | {cs / 1e3:.0f} k | {a[k][0]:.3f} | {a[k][1]:.3f} |")
print("\n| channels × 32 DSPVectors | round 2 | 1 lane | 2 lanes | 4 lanes | picked / round 2 |\n|---|---|---|---|---|---|")
for V in (4096, 16384, 32768, 49152, 65536, 262144):
    r = blocks[(V, 0)]
    us = {k: r[k][0] * 1000 for k in ("r2", "L1R8", "L2R8", "L4R8")}
    best = min(("L1R8", "L2R8", "L4R8"), key=lambda k: us[k])
    cells = [f"**{us[k]:.0f}**" if k == best else f"{us[k]:.0f}" for k in ("L1R8", "L2R8", "L4R8")]
    print(f"| {V:,} | {us['r2']:.0f} µs | {' | '.join(cells)} | {us['r2'] / us[best]:.2f} × |".replace(",", " "))
