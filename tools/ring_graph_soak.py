#!/usr/bin/env python3
"""Random CHAINS of 2 .. 6 delay nodes (IntegerDelay / FractionalDelay / PitchbendableDelay mixed, each followed by a mix with the input)
in every ring layout against the plain rows of rounds 2-5 (layout 0 with MLGPU_GRAPH_EARLY_READS=0 MLGPU_GRAPH_ROW_ADDR32=0), outputs
and every state word bit for bit over several launches. Delay times come from an input (stepped, with zeros and times of the ring's
length and beyond), a constant, the node's state, or the previous node's output; whole and ragged banks; write indices equal, or
different from voice to voice. tools/ring_layout_soak.py does single nodes with richer delay-time signals; this one is about several
nodes sharing a kernel (LDS slots per node, the reads of a sample issued together, a PitchbendableDelay's one ring).
    python tools/ring_graph_soak.py [cases] [seed]        MLGPU_SOAK_LAYOUTS=0,1,2,3,4 (default: all that apply to the graph)"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import madronalib_amd as ml                         # noqa: E402
from madronalib_amd.constants import Layout, Op, Proc  # noqa: E402
from inputs import lcg_noise, stepped               # noqa: E402

KNOBS = ("MLGPU_GRAPH_EARLY_READS", "MLGPU_GRAPH_ROW_ADDR32")


def build(rng, case, seed):
    V = int(rng.integers(1, 300)) if case % 2 else 64 * int(rng.integers(1, 5))
    T, launches = int(rng.integers(1, 6)), int(rng.integers(2, 4))
    S = 64 * T * launches
    n = int(rng.integers(2, 7))
    dmax = float([40.0, 100.0, 192.0, 700.0][int(rng.integers(0, 4))])
    ring = 1 << int(np.ceil(np.log2(max(64, int(dmax) + 64))))
    kinds = [[Proc.INTEGER_DELAY, Proc.FRACTIONAL_DELAY, Proc.PITCHBENDABLE_DELAY][int(rng.integers(0, 3))] for _ in range(n)]
    tmodes = [int(rng.integers(0, 4)) for _ in range(n)]   # 0 an input, 1 a constant, 2 none (the state's), 3 made of the previous node's output
    sig = {"x": lcg_noise(np.arange(V, dtype=np.uint32) + np.uint32(seed * 131 + case), S)}
    desc = [dict(name="x", type="input"), dict(name="half", type="const", value=0.5), dict(name="scale", type="const", value=float(dmax) * 0.45)]   # (|signal| stays under 2 - an allpass interpolator overshoots 1 -: a delay time made of it stays within the node's maximum)
    src = "x"
    for i, kind in enumerate(kinds):
        tm = tmodes[i]
        if kind == Proc.PITCHBENDABLE_DELAY and tm == 2:
            tm = 0                                         # (its call takes a delay time)
        ins = [src]
        if tm == 0:
            short = rng.random() < 0.4                     # (short ones: the history rows and the 16 / 32-sample boundaries of layouts 2 and 4)
            d = stepped(V, S, seed * 17 + i + case * 7, 0.0, min(dmax, 60.0) if short else dmax + 0.9, 1, 150)
            d[:, ::97] = 0.0                               # the read lands on the write ...
            if rng.random() < 0.5:
                d[:, 5::131] = np.float32(ring)            # ... also by the ring's whole length, and beyond it (layout 0's rows only promise
                d[:, 7::173] = np.float32(ring + 3)        #     the reference's bits within the node's maximum: see `within`)
                within = False
            else:
                within = True
            sig[f"dt{i}"] = d
            desc.append(dict(name=f"dt{i}", type="input"))
            ins.append(f"dt{i}")
        elif tm == 1:
            desc.append(dict(name=f"dt{i}", type="const", value=float(rng.integers(0, int(dmax)))))
            ins.append(f"dt{i}")
            within = True
        elif tm == 3:
            desc += [dict(name=f"ab{i}", type="op", kind=Op.ABS, inputs=[src]), dict(name=f"dt{i}", type="op", kind=Op.MULTIPLY, inputs=[f"ab{i}", "scale"])]
            ins.append(f"dt{i}")
            within = True                                  # under the maximum (see `scale`; seed 2 case 92 of the first version was not: a read
                                                           # 255 samples back in a 256-sample ring lands AHEAD of the writer inside the chunk being
                                                           # written, which layout 1 serves from its write window - beyond the maximum, outside the contract)
        else:
            within = True
        desc.append(dict(name=f"d{i}", type="proc", kind=kind, inputs=ins, max_delay=dmax, _within=within))
        desc += [dict(name=f"m{i}", type="op", kind=Op.ADD, inputs=[f"d{i}", "x"]), dict(name=f"s{i}", type="op", kind=Op.MULTIPLY, inputs=[f"m{i}", "half"])]
        src = f"s{i}"
    all_within = all(d.get("_within", True) for d in desc)
    for d in desc:
        d.pop("_within", None)
    wmode = int(rng.integers(0, 3))
    return dict(V=V, T=T, launches=launches, n=n, dmax=dmax, ring=ring, kinds=kinds, desc=desc, out=[src, "d0"], sig=sig, wmode=wmode, within=all_within)


def evaluate(eng, c, layout, case, knobs=None):
    saved = {k: os.environ.get(k) for k in KNOBS}
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(knobs or {})
    try:
        g = ml.Graph(eng, c["V"], c["desc"], c["out"], delay_windows=layout)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    V, T, ring = c["V"], c["T"], c["ring"]
    for i, kind in enumerate(c["kinds"]):
        if c["wmode"] == 1:
            g.set_state(f"d{i}", 0, np.full(V, (case * 11 + i * 8) % ring, np.uint32))
            if kind == Proc.PITCHBENDABLE_DELAY:
                g.set_state(f"d{i}", 5, np.full(V, (case * 11 + i * 8) % ring, np.uint32))
        elif c["wmode"] == 2:
            w = ((np.arange(V, dtype=np.uint32) * 5 + case * 3 + i) % ring).astype(np.uint32)
            g.set_state(f"d{i}", 0, w)
            if kind == Proc.PITCHBENDABLE_DELAY:
                g.set_state(f"d{i}", 5, w if case % 2 else ((w + 3) % ring).astype(np.uint32))
        if kind == Proc.INTEGER_DELAY:
            g.set_state(f"d{i}", 1, np.full(V, (7 * i + case) % max(1, int(c["dmax"])), np.uint32))
    outs = []
    for k in range(c["launches"]):
        part = {name: np.ascontiguousarray(a[:, k * 64 * T:(k + 1) * 64 * T]) for name, a in c["sig"].items()}
        outs.append(np.stack(g.process_host(T, part, Layout.QUAD)))
    states = []
    for i in range(c["n"]):
        states += [g.get_state(f"d{i}", j) for j in range(g.num_state(f"d{i}"))]
    eff = g.delay_layout
    g.close()
    return np.concatenate(outs, 2), np.stack(states), eff


def run(cases, seed, eng=None, layouts=None):
    rng = np.random.default_rng(seed)
    eng = eng or ml.Engine(0)
    want = layouts or [int(x) for x in os.environ.get("MLGPU_SOAK_LAYOUTS", "0,1,2,3,4").split(",")]
    bad, ran = 0, {}
    only = int(os.environ.get("MLGPU_SOAK_ONLY", "-1"))   # this chain alone (the others' random draws are still made), with where it differs
    for case in range(cases):
        c = build(rng, case, seed)
        if only >= 0 and case != only:
            continue
        rings = sum(2 if k == Proc.PITCHBENDABLE_DELAY else 1 for k in c["kinds"])
        base = evaluate(eng, c, 0, case, {"MLGPU_GRAPH_EARLY_READS": "0", "MLGPU_GRAPH_ROW_ADDR32": "0"})
        for layout in want:
            if layout == 2 and (rings > 4 or (c["V"] % 64 and False)):
                continue
            if layout in (1, 2, 4) and not c["within"]:
                continue   # (the windowed layouts promise the rows' bits for delay times within the node's maximum: include/mlgpu.h)
            if layout == 3 and not c["within"]:
                continue
            try:
                got = evaluate(eng, c, layout, case)
            except ml.MlgpuError as e:
                if e.status == ml.Status.ERR_UNSUPPORTED:   # (more LDS than the layout has for this graph: said at compile)
                    continue
                raise
            ran[layout] = ran.get(layout, 0) + 1
            nan = np.isnan(base[0]) & np.isnan(got[0])
            diff = int(((base[0].view(np.uint32) != got[0].view(np.uint32)) & ~nan).sum()) + int((base[1] != got[1]).sum())
            if diff:
                bad += 1
                where = np.argwhere((base[0].view(np.uint32) != got[0].view(np.uint32)) & ~nan)[:4].tolist()
                print(f"case {case}: layout {layout} (in effect {got[2]}): V {c['V']} T {c['T']} launches {c['launches']} kinds {[int(k) for k in c['kinds']]} max delay {c['dmax']} "
                      f"write-index mode {c['wmode']}: {diff} words differ, first at [output, voice, sample] {where}")
                if only >= 0:
                    for o, v, n in np.argwhere((base[0].view(np.uint32) != got[0].view(np.uint32)) & ~nan)[:12].tolist():
                        print(f"  output {o} voice {v} sample {n}: plain rows {base[0][o, v, n]!r} layout {layout} {got[0][o, v, n]!r}")
                    v = where[0][1]
                    n0 = where[0][2]
                    for name, a in c["sig"].items():
                        if name != "x":
                            print(f"  {name}[voice {v}, samples {max(0, n0 - 24)} .. {n0 + 2}]: {[round(float(t), 2) for t in a[v, max(0, n0 - 24):n0 + 3]]}")
                    print("  nodes:", [(d["name"], int(d["kind"]), d["inputs"]) for d in c["desc"] if d["type"] == "proc"], "consts:", [(d["name"], d["value"]) for d in c["desc"] if d["type"] == "const"])
                    print("  state words that differ [word, voice]:", np.argwhere(base[1] != got[1])[:12].tolist())
    print(f"{cases} chains (seed {seed}); graphs run per layout {dict(sorted(ran.items()))}: {bad} with a difference from the plain rows")
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
