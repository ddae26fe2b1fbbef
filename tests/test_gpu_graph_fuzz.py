"""Random graphs: processors and ops wired at random (a DAG in topological order), evaluated by the fused GPU kernel and by the
node-by-node oracle evaluator; every output and every processor's final state must agree bit for bit. Catches code-generation
slips that the hand-written patches do not reach (naming, rates, launch boundaries, ragged voice counts)."""
import os

import numpy as np
import pytest

from graph_oracle import evaluate
from inputs import assert_bits_equal, lcg_noise, proc_default_coeffs
from madronalib_amd.constants import Layout, Op, Proc

UNARY = [Op.ABS, Op.SIGN, Op.SIN_APPROX, Op.COS_APPROX, Op.EXP2_APPROX, Op.FRACTIONAL_PART]
BINARY = [Op.ADD, Op.SUBTRACT, Op.MULTIPLY, Op.MIN, Op.MAX]
FILTERS = [Proc.LOPASS, Proc.HIPASS, Proc.BANDPASS, Proc.ONE_POLE, Proc.DC_BLOCKER, Proc.INTEGRATOR, Proc.DIFFERENTIATOR, Proc.ALLPASS1]
GENS = [Proc.SINE_GEN, Proc.SAW_GEN, Proc.PHASOR_GEN, Proc.TICK_GEN]


@pytest.fixture(scope="module")
def eng():
    import madronalib_amd as ml
    e = ml.Engine(0)
    yield e
    e.close()


def random_graph(rng, oracle, V):
    desc = [dict(name="x", type="input"), dict(name="f", type="param"), dict(name="half", type="const", value=0.5),
            dict(name="small", type="const", value=0.01)]
    audio = ["x"]                      # audio-rate nodes available as inputs
    params = {"f": (rng.uniform(0.001, 0.02, V)).astype(np.float32)}
    coeffs = {}
    n_nodes = int(rng.integers(4, 12))
    for i in range(n_nodes):
        name = f"n{i}"
        r = rng.random()
        if r < 0.05:
            desc.append(dict(name=name, type="proc", kind=Proc.NOISE_GEN, inputs=[]))
        elif r < 0.2:
            kind = int(rng.choice(GENS))
            src = "f" if rng.random() < 0.6 else None
            if src is None:            # an audio-rate frequency: |signal| * 0.01 keeps it in range
                desc.append(dict(name=name + "a", type="op", kind=Op.ABS, inputs=[str(rng.choice(audio))]))
                desc.append(dict(name=name + "s", type="op", kind=Op.MULTIPLY, inputs=[name + "a", "small"]))
                src = name + "s"
            desc.append(dict(name=name, type="proc", kind=kind, inputs=[src]))
        elif r < 0.5:
            kind = int(rng.choice(FILTERS))
            desc.append(dict(name=name, type="proc", kind=kind, inputs=[str(rng.choice(audio))]))
            co = proc_default_coeffs(oracle, kind, V, seed=int(rng.integers(0, 1000)))
            if co is not None and np.size(co):
                coeffs[name] = np.ascontiguousarray(co, np.float32)
        elif r < 0.7:
            desc.append(dict(name=name, type="op", kind=int(rng.choice(UNARY)), inputs=[str(rng.choice(audio))]))
        else:
            a, b = str(rng.choice(audio)), str(rng.choice(audio + ["half"]))
            desc.append(dict(name=name, type="op", kind=int(rng.choice(BINARY)), inputs=[a, b]))
        audio.append(name)
    outs = list(dict.fromkeys([audio[-1], str(rng.choice(audio[1:]))]))
    return desc, outs, params, coeffs


@pytest.mark.gpu
# (a longer campaign: MLGPU_FUZZ_FIRST=40 MLGPU_FUZZ_SEEDS=400 python -m pytest tests/test_gpu_graph_fuzz.py -m gpu)
@pytest.mark.parametrize("seed", range(int(os.environ.get("MLGPU_FUZZ_FIRST", "0")), int(os.environ.get("MLGPU_FUZZ_FIRST", "0")) + int(os.environ.get("MLGPU_FUZZ_SEEDS", "40"))))
def test_random_graph_vs_evaluator(eng, oracle, seed):
    import madronalib_amd as ml
    rng = np.random.default_rng(1000 + seed)
    V, T = int(rng.choice([64, 70, 300, 513])), 5
    desc, outs, params, coeffs = random_graph(rng, oracle, V)
    # every third graph reads its constants from the device table instead of the instruction stream: same bits
    g = ml.Graph(eng, V, desc, outs, voices_per_lane=int(rng.choice([0, 1, 2])), autotune=False, live_constants=(seed % 3 == 0))
    g.clear()
    for k, v in params.items():
        g.set_param(k, v)
    for k, c in coeffs.items():
        g.set_coeffs(k, [np.ascontiguousarray(r) for r in c])
    states = {n["name"]: oracle.chain_clear([n["kind"]], V) for n in desc if n["type"] == "proc"}
    x = lcg_noise(np.arange(V, dtype=np.uint32) + np.uint32(seed), 64 * T * 2) * np.float32(0.5)
    for call, layout in enumerate((Layout.QUAD, Layout.VOICE_MAJOR)):
        sig = {"x": np.ascontiguousarray(x[:, call * 64 * T:(call + 1) * 64 * T])}
        got = g.process_host(T, sig, layout)
        want = evaluate(oracle, desc, outs, V, T, sig, params, coeffs, states)
        for o, a, b in zip(outs, got, want):
            assert_bits_equal(a, b, True, f"seed {seed} call {call} output {o}\\n{desc}")
    for n in desc:
        if n["type"] == "proc":
            for i in range(g.num_state(n["name"])):
                assert (g.get_state(n["name"], i) == states[n["name"]][i]).all(), (seed, n["name"], i)


@pytest.mark.gpu
def test_random_graphs_with_delay_lines_and_feedback(eng):
    """A short run of tools/graph_stream_fuzz.py: the random graphs above plus delay nodes of all kinds in random ring layouts plus one-vector
    feedback, against the oracle's vector-by-vector evaluator. (840 graphs of it at the round's end: profiles/r06_graph_stream_fuzz.txt.)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("graph_stream_fuzz", os.path.join(os.path.dirname(__file__), "..", "tools", "graph_stream_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(24, 2000, eng) == 0
