"""Processors and ops BY NAME: the registry (mlgpu_registry_*) in the convention of the reference's dynamic
proc stub - a class registered under a string, its params / inputs / outputs named by strings
(source/procs/MLProcMultiply.cpp:12-18,29-32,44-47) - and graphs described entirely with strings. No device needed: graphs
are built offline and compared through the kernel source they generate."""
import numpy as np
import pytest

import madronalib_amd as ml
from madronalib_amd import _lib, constants, patches
from madronalib_amd.constants import Op, Proc


def test_the_reference_example_name_is_registered():
    reg = ml.registry()
    m = reg["multiply"]                     # ProcRegistryEntry<ProcMultiply> classReg("multiply")
    assert m["node_type"] == 0 and m["kind"] == Op.MULTIPLY and m["inputs"] == ["in1", "in2"] and m["output"] == "out" and m["params"] == []


def test_every_graph_node_kind_has_a_name():
    reg = ml.registry()
    L = _lib.load()
    kinds = {(e["node_type"], e["kind"]) for e in reg.values()}
    for op in Op.UNARY + Op.BINARY + Op.TERNARY:
        if op == Op.EXP_APPROX_OF_SIN_APPROX:      # the fused benchmark pair is not a reference function
            continue
        assert (0, op) in kinds, f"op {op} has no registry name"
    for kind in list(Proc.ALL) + list(Proc.DELAYS) + list(Proc.VECTOR_RATE):
        assert (1, kind) in kinds, f"proc {kind} has no registry name"
    assert len(reg) == len({(e["node_type"], e["kind"]) for e in reg.values()})       # one name per kind
    probe = ml.Graph(ml.OfflineEngine(), 64)
    probe.add("x", "input")
    probe.add("c", "control")      # one float per DSPVector: what Interpolator1 / LinearGlide / TempoLock's float arguments take
    for name, e in reg.items():
        assert name == name.lower() and " " not in name
        if e["node_type"] == 1:
            ins = ["x"] * len(e["inputs"])
            if e["kind"] in (Proc.INTERPOLATOR1, Proc.LINEAR_GLIDE):
                ins = ["c"]
            if e["kind"] == Proc.TEMPO_LOCK:
                ins = ["x", "c", "c"]
            node = probe.add_named(name, "n_" + name, ins)
            assert len(e["params"]) == L.mlgpu_graph_num_coeffs(probe.h, node), name     # coefficient names cover every slot
            assert L.mlgpu_registry_param_index(name.encode(), e["params"][-1].encode()) == len(e["params"]) - 1 if e["params"] else True
        assert 0 <= e["required_inputs"] <= len(e["inputs"])
    assert reg["lopass"]["inputs"] == ["in", "omega", "k"] and reg["lopass"]["required_inputs"] == 1
    assert reg["lopass"]["params"] == ["g0", "g1", "g2"] and reg["adsr"]["inputs"] == ["gate"]
    assert L.mlgpu_registry_lookup(b"no_such_proc", None) == ml.Status.ERR_RANGE


def _strings_only_synth16(full):
    """patches.synth16 again, with nothing but strings: registry names for the node kinds, node names for the wiring."""
    kind_name = {(e["node_type"], e["kind"]): n for n, e in ml.registry().items()}
    desc, outs = patches.synth16(full=full)
    g = ml.Graph(ml.OfflineEngine(), 512)
    for n in desc:
        if n["type"] in ("input", "param"):
            g.add(n["name"], n["type"])
        elif n["type"] == "const":
            g.add(n["name"], "const", value=n["value"])
        else:
            g.add_named(kind_name[(1 if n["type"] == "proc" else 0, n["kind"])], n["name"], n["inputs"])
    for o in outs:
        g.add_output(o)
    return g


@pytest.mark.parametrize("full", [False, True])
def test_a_patch_described_with_strings_is_the_same_kernel(full):
    desc, outs = patches.synth16(full=full)
    by_enum = ml.Graph(ml.OfflineEngine(), 512, desc, outs)
    by_name = _strings_only_synth16(full)
    src_a, code_a = by_enum.emit()
    src_b, code_b = by_name.emit()
    assert src_a == src_b and code_a == code_b and code_a[:4] == b"\x7fELF"


def test_named_errors():
    g = ml.Graph(ml.OfflineEngine(), 64)
    g.add("x", "input")
    with pytest.raises(ml.MlgpuError):
        g.add_named("nonsense", "n", ["x"])
    with pytest.raises(ml.MlgpuError):
        g.add_named("multiply", "m", ["x"])              # two inputs required
    with pytest.raises(ml.MlgpuError):
        g.add_named("multiply", "m", ["x", "nobody"])    # unknown input node
    g.add_named("lopass", "lp", ["x"])
    g.add_named("multiply", "m", ["lp", "x"])
    with pytest.raises(ml.MlgpuError):
        g.set_named_coeff("lp", "q", 0.5)                # Lopass has g0, g1, g2
    with pytest.raises(ml.MlgpuError):
        g.set_named_coeff("m", "g0", 0.5)                # an op has no coefficients
