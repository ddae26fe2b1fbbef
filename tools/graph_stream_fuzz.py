#!/usr/bin/env python3
"""Random graphs WITH delay lines and one-vector feedback - tests/test_gpu_graph_fuzz.py's random generators / filters / operators, plus
1 .. 4 delay nodes of all three kinds (delay times per voice, made of an audio node, or the node's state) in a random ring layout, plus
0 .. 2 feedback nodes - on the device against the oracle's vector-by-vector evaluator (tests/graph_oracle.py: evaluate_stream), every
output of two launches bit for bit. The delay tests pin the delay nodes in small fixed graphs; this puts them next to everything else
the generator emits (node order, the reads issued ahead, LDS of several features in one kernel).
    python tools/graph_stream_fuzz.py [cases] [first seed]"""
import os
import sys
import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import madronalib_amd as ml                                   # noqa: E402
from madronalib_amd.constants import Layout, Op, Proc          # noqa: E402
from cpu_checkers import Oracle                               # noqa: E402
from graph_oracle import evaluate_stream, new_stream_state    # noqa: E402
from inputs import lcg_noise                                  # noqa: E402
import test_gpu_graph_fuzz as gf                              # noqa: E402


# the wider pool (every second graph): every operator but the hardware-approximate ones (rcpps / rsqrtps: a tolerance, not bits) and the
# integer-input ones; more processors of the families that take one input and per-voice coefficients
# NaN: "any NaN equals any NaN" is the contract - which of its operands' NaNs an instruction hands on, and with which sign,
# is the hardware's business (x86 keeps the first operand's, negative for the all-ones mask of a comparison; the GPU makes a positive one).
# An operator that READS a NaN's sign (sign, signBit) turns that into +1 / -1: seeds 3749 and 3835 of the first wide run, both a comparison
# mask multiplied as a float and fed to sign() through a feedback node. So here: masks go to select() only, and sign / signBit see no NaN
# (they are replaced by abs in the wide graphs, where sqrt / log / divide can make one).
COMPARES = (Op.EQUAL, Op.NOT_EQUAL, Op.GREATER_THAN, Op.GREATER_THAN_OR_EQUAL, Op.LESS_THAN, Op.LESS_THAN_OR_EQUAL)
# ... and so does an approximation that takes a float's bits apart: logApprox / log2Approx / powApprox of a NaN are finite numbers made of its
# sign and payload (seed 9963 of the second wide run: powApprox(NaN, 0.05) = 9 529.8 for 7fc00000 and 8.3e7 for ffc00000 - on the device, in
# the oracle and in the reference alike, given the same NaN). They stay out of the wide graphs too.
NAN_BITS_READERS = (Op.LOG_APPROX, Op.LOG2_APPROX, Op.POW_APPROX)
WIDE_UNARY = [k for k in Op.UNARY if k not in Op.HW_APPROX and k not in Op.INT_INPUT and k not in (Op.SIGN, Op.SIGN_BIT) and k not in NAN_BITS_READERS]
WIDE_BINARY = [k for k in Op.BINARY if k not in Op.HW_APPROX and k not in Op.INT_INPUT and k not in COMPARES and k not in NAN_BITS_READERS]
WIDE_TERNARY = [k for k in Op.TERNARY if k not in (Op.SELECT_INT, Op.SELECT, Op.WITHIN)]
WIDE_PROCS = [Proc.LO_SHELF, Proc.HI_SHELF, Proc.BELL, Proc.GAIN, Proc.ADSR, Proc.SAMPLE_ACCURATE_LINEAR_GLIDE, Proc.IMPULSE_GEN, Proc.PULSE_GEN, Proc.TEST_SINE_GEN]


def widen(rng, orc, V, desc, params, coeffs):
    from inputs import proc_default_coeffs
    audio = [d["name"] for d in desc if d["type"] in ("input", "proc", "op")]
    for i in range(int(rng.integers(3, 10))):
        name = f"w{i}"
        r = rng.random()
        if r < 0.3:
            desc.append(dict(name=name, type="op", kind=int(rng.choice(WIDE_UNARY)), inputs=[str(rng.choice(audio))]))
        elif r < 0.55:
            desc.append(dict(name=name, type="op", kind=int(rng.choice(WIDE_BINARY)), inputs=[str(rng.choice(audio)), str(rng.choice(audio + ["half"]))]))
        elif r < 0.68:
            desc.append(dict(name=name, type="op", kind=int(rng.choice(WIDE_TERNARY)), inputs=[str(rng.choice(audio)), str(rng.choice(audio + ["half"])), str(rng.choice(audio + ["small"]))]))
        elif r < 0.78:   # a comparison's mask, used as a mask: select(a, b, a' < b')
            desc.append(dict(name=name + "m", type="op", kind=int(rng.choice(COMPARES)), inputs=[str(rng.choice(audio)), str(rng.choice(audio + ["half", "small"]))]))
            desc.append(dict(name=name, type="op", kind=Op.SELECT, inputs=[str(rng.choice(audio)), str(rng.choice(audio + ["half"])), name + "m"]))
        else:
            kind = int(rng.choice(WIDE_PROCS))
            if kind in (Proc.IMPULSE_GEN, Proc.PULSE_GEN, Proc.TEST_SINE_GEN):
                ins = ["f"]
            else:
                ins = [str(rng.choice(audio))]
            desc.append(dict(name=name, type="proc", kind=kind, inputs=ins))
            co = proc_default_coeffs(orc, kind, V, seed=int(rng.integers(0, 1000)))
            if co is not None and np.size(co):
                coeffs[name] = np.ascontiguousarray(co, np.float32)
        audio.append(name)
    return audio


def build(rng, orc, V, wide=False):
    desc, outs, params, coeffs = gf.random_graph(rng, orc, V)
    if wide:
        widen(rng, orc, V, desc, params, coeffs)
        for d in desc:
            if d["type"] == "op" and d["kind"] in (Op.SIGN, Op.SIGN_BIT):
                d["kind"] = Op.ABS
            if d["type"] == "op" and d["kind"] in (Op.LOG_APPROX, Op.LOG2_APPROX):
                d["kind"] = Op.LOG2
    audio = [d["name"] for d in desc if d["type"] in ("input", "proc", "op")]
    dmax = float(rng.choice([40.0, 100.0, 700.0]))
    desc.append(dict(name="dmaxc", type="const", value=dmax * 0.98))
    desc.append(dict(name="dscale", type="const", value=dmax))
    rings = 0
    for i in range(int(rng.integers(1, 5))):
        kind = [Proc.INTEGER_DELAY, Proc.FRACTIONAL_DELAY, Proc.PITCHBENDABLE_DELAY][int(rng.integers(0, 3))]
        src = str(rng.choice(audio))
        tm = int(rng.integers(0, 3))       # 0: a per-voice parameter, 1: made of an audio node, 2: none (the state's)
        if kind == Proc.PITCHBENDABLE_DELAY and tm == 2:
            tm = 0
        ins = [src]
        if tm == 0:
            params[f"dl{i}"] = rng.uniform(0.0, dmax, V).astype(np.float32)
            desc.append(dict(name=f"dl{i}", type="param"))
            ins.append(f"dl{i}")
        elif tm == 1:
            a = str(rng.choice(audio))
            desc += [dict(name=f"da{i}", type="op", kind=Op.ABS, inputs=[a]), dict(name=f"ds{i}", type="op", kind=Op.MULTIPLY, inputs=[f"da{i}", "dscale"]),
                     dict(name=f"dl{i}", type="op", kind=Op.MIN, inputs=[f"ds{i}", "dmaxc"])]
            ins.append(f"dl{i}")
        desc.append(dict(name=f"d{i}", type="proc", kind=kind, inputs=ins, max_delay=dmax))
        # the delayed signal goes back into the pool through a mix, so later nodes (and outputs) use it
        desc.append(dict(name=f"dm{i}", type="op", kind=Op.ADD, inputs=[f"d{i}", str(rng.choice(audio))]))
        desc.append(dict(name=f"dh{i}", type="op", kind=Op.MULTIPLY, inputs=[f"dm{i}", "half"]))
        audio.append(f"dh{i}")
        rings += 2 if kind == Proc.PITCHBENDABLE_DELAY else 1
    # feedback: a node early in the list that reads, one DSPVector late, a node from the end of it
    for j in range(int(rng.integers(0, 3))):
        source = str(rng.choice(audio[-3:]))
        fb = dict(name=f"fb{j}", type="feedback", source=source)
        user = dict(name=f"fu{j}", type="op", kind=Op.MULTIPLY, inputs=[f"fb{j}", "small"])
        # insert right after the input and mix it into the first audio-rate consumer chain through a new node others may pick
        pos = next(k for k, d in enumerate(desc) if d["name"] == "small") + 1
        desc[pos:pos] = [fb, user]
        # let one existing op read it
        for d in desc[pos + 2:]:
            if d["type"] == "op" and d.get("inputs") and d["inputs"][0] in audio and rng.random() < 0.3 and d["name"] != source:
                mix = dict(name=f"fx{j}", type="op", kind=Op.ADD, inputs=[d["inputs"][0], f"fu{j}"])
                k = desc.index(d)
                desc.insert(k, mix)
                d["inputs"] = [f"fx{j}"] + d["inputs"][1:]
                break
    outs = list(dict.fromkeys([audio[-1], outs[0]]))
    return desc, outs, params, coeffs, rings


def run(cases, first, eng=None):
    eng = eng or ml.Engine(0)
    orc = Oracle()
    bad = 0
    layouts_run = {}
    for seed in range(first, first + cases):
        rng = np.random.default_rng(5000 + seed)
        V, T = int(rng.choice([64, 70, 256, 300])), int(rng.integers(2, 6))
        desc, outs, params, coeffs, rings = build(rng, orc, V, wide=bool(seed % 2))
        layout = int(rng.choice([0, 0, 1, 2, 3, 4]))
        if layout == 2 and rings > 4:
            layout = 3
        # the wide graphs blow up (pow, exp, shelving filters fed infinities): there the SVF accumulators as the reference's two operations
        # (deviation (a): `fma(2, t, ic)` differs from `ic + 2 t` exactly when 2 t overflows - seed 20885 of the third campaign,
        # a HiShelf fed pow(0, negative) = inf: -inf on the device, NaN in the oracle; bit for bit with the switch on)
        eng.set_strict_svf(bool(seed % 2))
        try:
            g = ml.Graph(eng, V, desc, outs, delay_windows=layout)
        except ml.MlgpuError as e:
            if e.status != ml.Status.ERR_UNSUPPORTED:
                raise
            layout = 0
            g = ml.Graph(eng, V, desc, outs, delay_windows=0)
        layouts_run[g.delay_layout] = layouts_run.get(g.delay_layout, 0) + 1
        g.clear()
        for k, v in params.items():
            g.set_param(k, v)
        for k, c in coeffs.items():
            g.set_coeffs(k, [np.ascontiguousarray(r) for r in c])
        st = new_stream_state(orc, desc, V)
        for n in desc:
            if n["type"] == "proc" and n["kind"] not in Proc.DELAYS:
                st[n["name"]] = orc.chain_clear([n["kind"]], V)
        x = lcg_noise(np.arange(V, dtype=np.uint32) + np.uint32(seed), 64 * T * 2) * np.float32(0.5)
        diff = 0
        for call in range(2):
            sig = {"x": np.ascontiguousarray(x[:, call * 64 * T:(call + 1) * 64 * T])}
            got = g.process_host(T, sig, Layout.QUAD)
            want = evaluate_stream(orc, desc, outs, V, T, sig, params, coeffs, st)
            for o, a, b in zip(outs, got, want):
                nan = np.isnan(a) & np.isnan(b)
                m = (a.view(np.uint32) != b.view(np.uint32)) & ~nan
                if m.any():
                    w = np.argwhere(m)
                    diff += int(m.sum())
                    print(f"seed {seed} call {call} output {o}: {int(m.sum())} words differ, first [voice, sample] {w[0].tolist()}: device {a[tuple(w[0])]!r} oracle {b[tuple(w[0])]!r}")
        g.close()
        eng.set_strict_svf(False)
        if diff:
            bad += 1
            print(f"seed {seed}: V {V} T {T} ring layout asked {layout}, {rings} rings\\n  " + "\\n  ".join(str({k: (int(v) if isinstance(v, (np.integer,)) else v) for k, v in d.items()}) for d in desc))
    print(f"{cases} graphs (seeds {first} .. {first + cases - 1}); ring layouts in effect {dict(sorted(layouts_run.items()))}: {bad} with a difference from the oracle's evaluator")
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
