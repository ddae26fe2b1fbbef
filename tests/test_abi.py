"""CPU-side checks of the C-ABI library: it builds for gfx950, loads, exports every symbol that
include/mlgpu.h declares, its enums match the Python mirror, the host-side coefficient makers
match the oracle, and without a GPU every compute entry fails loudly (no silent fallback)."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from madronalib_amd import _lib, constants

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mlgpu.h")


def _header():
    with open(HEADER) as f:
        return f.read()


def test_library_builds_and_loads():
    L = _lib.load()
    assert os.path.exists(_lib.LIB_PATH)
    assert L.mlgpu_abi_version() == 2


def test_every_declared_symbol_is_exported():
    L = _lib.load()
    src = re.sub(r"/\*.*?\*/", "", _header(), flags=re.S)
    names = set(re.findall(r"\b(mlgpu_[a-z0-9_]+)\s*\(", src))
    assert len(names) > 40
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing


def test_code_object_is_gfx950():
    data = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in data
    assert b"chain_kernel" in data and b"op_kernel" in data


def test_python_enums_match_header():
    h = _header()
    vals = {m.group(1): int(m.group(2)) for m in re.finditer(r"\b(MLGPU_[A-Z0-9_]+)\s*=\s*(\d+)", h)}
    for cls, prefix in ((constants.Op, "MLGPU_OP_"), (constants.Proc, "MLGPU_PROC_"), (constants.Layout, "MLGPU_LAYOUT_"),
                        (constants.RowOp, "MLGPU_ROWOP_"), (constants.Status, "MLGPU_"), (constants.Vop, "MLGPU_VOP_"),
                        (constants.Region, "MLGPU_REGION_"), (constants.Route, "MLGPU_ROUTE_")):
        for k, v in vars(cls).items():
            if k.startswith("_") or not isinstance(v, int):
                continue
            assert vals[prefix + k] == v, (prefix + k, v)
    hdr_procs = {v for k, v in vals.items() if k.startswith("MLGPU_PROC_")}
    assert hdr_procs == set(constants.Proc.ALL) | set(constants.Proc.GRAPH_ONLY)


def test_coefficient_makers_match_oracle(oracle):
    import madronalib_amd as ml
    rng = np.random.default_rng(1)
    for _ in range(100):
        om, k, A = rng.uniform(0.0005, 0.49), rng.uniform(0.01, 3), rng.uniform(0.1, 8)
        assert (ml.Lopass.makeCoeffs(om, k).view(np.uint32) == oracle.make_coeffs("lopass", om, k).view(np.uint32)).all()
        assert (ml.Hipass.makeCoeffs(om, k).view(np.uint32) == oracle.make_coeffs("hipass", om, k).view(np.uint32)).all()
        assert (ml.Bandpass.makeCoeffs(om, k).view(np.uint32) == oracle.make_coeffs("bandpass", om, k).view(np.uint32)).all()
        assert (ml.LoShelf.makeCoeffs(om, k, A).view(np.uint32) == oracle.make_coeffs("loshelf", om, k, A).view(np.uint32)).all()
        assert (ml.HiShelf.makeCoeffs(om, k, A).view(np.uint32) == oracle.make_coeffs("hishelf", om, k, A).view(np.uint32)).all()
        assert (ml.Bell.makeCoeffs(om, k, A).view(np.uint32) == oracle.make_coeffs("bell", om, k, A).view(np.uint32)).all()
        assert (ml.OnePole.makeCoeffs(om).view(np.uint32) == oracle.make_coeffs("onepole", om).view(np.uint32)).all()
        assert ml.DCBlocker.makeCoeffs(om) == oracle.dcblocker_coeffs(om)
        assert ml.dBToGain(A) == oracle.db_to_gain(A)
        a = (rng.uniform(0, .1), rng.uniform(0, .1), rng.random(), rng.uniform(0, .1), 48000.0)
        assert (ml.ADSR.calcCoeffs(*a).view(np.uint32) == oracle.make_coeffs("adsr", *a).view(np.uint32)).all()


def test_no_gpu_means_loud_failure():
    """Without a gfx950 device the product refuses to run instead of falling back to the CPU."""
    import madronalib_amd as ml
    if ml.device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(ml.MlgpuError) as ei:
        ml.Engine(0)
    assert ei.value.status == ml.Status.ERR_NO_DEVICE


def test_product_does_not_reference_oracle():
    """The shipped package must never import, link or load anything under oracle/."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = []
    for base in ("madronalib_amd", "include"):
        for dp, _, files in os.walk(os.path.join(root, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"oracle/|mlorc_|mlref_|libmloracle|libmlref", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def _code_object_notes(code):
    import subprocess
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(code)
        f.flush()
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout


@pytest.mark.parametrize("windows", [False, True])
def test_offline_graph_emit_keeps_processors_in_registers(windows):
    """mlgpu_graph_emit generates and compiles a graph's kernel without a device. The delay-line graph must not touch
    scratch memory (a processor object the optimizer cannot split ends up there: ~15 % slower, seen in round 1) and the
    windowed form must fit two waves per SIMD; an offline graph refuses to compile for a device."""
    import madronalib_amd as ml
    from madronalib_amd import patches
    from madronalib_amd.constants import Proc
    desc = [dict(name="x", type="input"), dict(name="dl", type="param")]
    src = "x"
    for j in range(4):
        sub, src = patches.allpass(f"ap{j}_", src, Proc.PITCHBENDABLE_DELAY, 4096.0 - 64.0, "dl")
        desc += sub
    g = ml.Graph(ml.OfflineEngine(), 1024, desc, [src], delay_windows=windows)
    source, code = g.emit()
    assert "mlgpu_graph_kernel" in source and ("MLGPU_RING_WINDOWS 1" in source) == windows
    assert code[:4] == b"\x7fELF"
    notes = _code_object_notes(code)
    assert "amdgcn-amd-amdhsa--gfx950" in notes
    vgpr = int(re.search(r"\.vgpr_count:\s+(\d+)", notes).group(1)) + int((re.search(r"\.agpr_count:\s+(\d+)", notes) or [0, 0])[1])
    scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", notes).group(1))
    lds = int(re.search(r"\.group_segment_fixed_size:\s+(\d+)", notes).group(1))
    assert vgpr <= 256
    assert lds == (8 * 8 * 256 * 4 if windows else 8 * 4 * 64 * 4)   # rows: a 256-byte landing slot per ring read and wavefront (the early reads)
    # the windowed form is held to two waves per SIMD (256 registers): a few dozen spilled values are the price; a processor
    # object living in scratch would be several hundred bytes
    assert scratch == 0 if not windows else scratch <= 256
    with pytest.raises(ml.MlgpuError) as ei:
        g.compile()
    assert ei.value.status == ml.Status.ERR_INVALID
    g.close()


def test_register_budget_of_generated_kernels(monkeypatch):
    """graph.hip: generateBudgeted — for a bank big enough to fill the chip (65 536 voices and up), a generated kernel that comes
    out above 128 VGPRs (three, two or one wavefront per SIMD: the bank's blocks then run in rounds) is generated again with a
    four-wavefront bound and kept when its scratch is moderate; a kernel that fits anyway is left alone, and so is any kernel of
    a small bank; MLGPU_GRAPH_MIN_WAVES=0 switches the policy off. Decided from the code objects' metadata: no device needed."""
    import madronalib_amd as ml
    from madronalib_amd import patches

    def build(V, **kw):
        desc, outs = patches.synth16(**kw)
        g = ml.Graph(ml.OfflineEngine(), V, desc, outs)
        source, code = g.emit()
        g.close()
        notes = _code_object_notes(code)
        return (re.search(r"__launch_bounds__\([^)]*\)", source).group(0), int(re.search(r"\.vgpr_count:\s+(\d+)", notes).group(1)),
                int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", notes).group(1)))
    monkeypatch.delenv("MLGPU_GRAPH_MIN_WAVES", raising=False)
    # the 16-node voice: with its streamed gate a quad ahead (round 4) it comes out a few registers above 128 - 133 means blocks of
    # 136, three wavefronts per SIMD - and is bounded too, for a handful of words kept in scratch outside the sample loop
    bounds, vgpr, scratch = build(262144)
    assert bounds == "__launch_bounds__(256, 4)" and vgpr <= 128 and scratch <= 64
    bounds, vgpr, scratch = build(262144, full=True)           # the 22-node patch does not: bounded, with a little scratch
    assert bounds == "__launch_bounds__(256, 4)" and vgpr <= 128 and 0 < scratch <= 640
    # EventsToSignals' rows inside: round 4's record-walking form was far above (134 spilled registers under the bound); round 5's
    # control-record form (mlev::CtlVoice, buffer-descriptor addressing) fits the four-wavefront budget with next to no scratch
    bounds, vgpr, scratch = build(262144, pitch_input=True, event_rows=True)
    assert vgpr <= 128 and scratch <= 64
    bounds, vgpr, scratch = build(1024, full=True)             # a small bank does not fill the chip: the compiler's choice stands
    assert bounds == "__launch_bounds__(256)" and vgpr > 128 and scratch == 0
    monkeypatch.setenv("MLGPU_GRAPH_MIN_WAVES", "0")
    bounds, vgpr, scratch = build(262144, full=True)
    assert bounds == "__launch_bounds__(256)" and vgpr > 128 and scratch == 0


def test_offline_emit_of_const_vectors_live_constants_and_regions():
    """Code generation paths that only GPU tests would otherwise reach, compiled here with hiprtc for gfx950: a constant
    DSPVector (a __constant__ table), live constants (read from the argument table, not literals), a rate region, and the
    structural comparison behind mlgpu_graph_update_constants_from (no device needed for that either)."""
    import madronalib_amd as ml
    from madronalib_amd.constants import Op, Proc, Region

    def build(gain, live, wire_other=False):
        g = ml.Graph(ml.OfflineEngine(), 512, live_constants=live)
        g.add("x", "input")
        g.add("gain", "const", value=gain)
        g.add("win", "const_vector", value=np.hanning(64).astype(np.float32))
        g.begin_region(Region.UPSAMPLE_2X, ["x"], ["rx"])
        g.add("tbl", "const_vector", value=np.linspace(0, 1, 64, dtype=np.float32))
        g.add("shaped", "op", Op.MULTIPLY, ["rx", "tbl"])
        g.add("lp", "proc", Proc.LOPASS, ["shaped"])
        g.end_region("lp", "y")
        g.add("yw", "op", Op.MULTIPLY, ["y", "win"])
        g.add("out", "op", Op.MULTIPLY, ["x" if wire_other else "yw", "gain"])
        g.add_output("out")
        return g

    g = build(0.5, True)
    source, code = g.emit()
    assert code[:4] == b"\x7fELF"
    assert source.count("__constant__ unsigned cv") == 2 and "a.consts[0]" in source
    lit_source, lit_code = build(0.5, False).emit()
    assert "a.consts[" not in lit_source and lit_code[:4] == b"\x7fELF"
    # set_const before compile changes the value a later compile would start from; never an error
    g.set_const("gain", 0.25)
    with pytest.raises(ml.MlgpuError):
        g.set_const("x", 1.0)                        # not a const node
    with pytest.raises(ml.MlgpuError):
        g.update_constants_from(build(0.75, True))   # g itself is not compiled: nothing to update


def test_hiprtc_disk_cache_spares_the_second_process(tmp_path):
    """Run-time fused kernels are cached on disk under MLGPU_CACHE_DIR, keyed by source + options + embedded headers +
    hiprtc version: the first process compiles, the next one loads the same code object without calling hiprtc; a
    corrupted entry is ignored and rebuilt; MLGPU_CACHE_DIR=off compiles every time. (No device needed: offline emit.)"""
    prog = r'''
import json, sys, time
import madronalib_amd as ml
from madronalib_amd.constants import Op, Proc
g = ml.Graph(ml.OfflineEngine(), 256)
g.add("x", "input"); g.add("k", "const", value=float(sys.argv[1]))
g.add("lp", "proc", Proc.LOPASS, ["x"]); g.add("y", "op", Op.MULTIPLY, ["lp", "k"]); g.add_output("y")
t0 = time.perf_counter(); src, code = g.emit(); dt = time.perf_counter() - t0
import hashlib
print(json.dumps(dict(stats=ml.jit_stats(), seconds=dt, sha=hashlib.sha256(code).hexdigest(), n=len(code))))
'''

    def run(cache_dir, k="0.5"):
        env = dict(os.environ, MLGPU_CACHE_DIR=str(cache_dir), PYTHONPATH=ROOT)
        r = subprocess.run([sys.executable, "-c", prog, k], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads(r.stdout.strip().splitlines()[-1])
    cold = run(tmp_path)
    assert cold["stats"]["compiles"] == 1 and cold["stats"]["disk_hits"] == 0
    files = [f for f in os.listdir(tmp_path) if f.endswith(".co")]
    assert len(files) == 1
    warm = run(tmp_path)
    assert warm["stats"]["compiles"] == 0 and warm["stats"]["disk_hits"] == 1
    assert warm["sha"] == cold["sha"] and warm["seconds"] < cold["seconds"]
    other = run(tmp_path, "0.25")                    # another constant = another source = another entry
    assert other["stats"]["compiles"] == 1 and len([f for f in os.listdir(tmp_path) if f.endswith(".co")]) == 2
    with open(os.path.join(tmp_path, files[0]), "r+b") as f:   # a torn write: not an ELF any more
        f.write(b"garbage!")
    again = run(tmp_path)
    assert again["stats"]["compiles"] == 1 and again["sha"] == cold["sha"]
    off = run("off")
    assert off["stats"]["compiles"] == 1 and off["stats"]["disk_hits"] == 0
    print(f"hiprtc cold {cold['seconds']:.2f} s, warm {warm['seconds']:.3f} s")

    # An entry is used only when the context (options, device-source fingerprint, hiprtc / runtime versions) and the
    # generated source stored in it are byte for byte the ones asked for - the file name is only a 64-bit hash.
    names = sorted(f for f in os.listdir(tmp_path) if f.endswith(".co"))
    assert len(names) == 2
    a, b = (os.path.join(tmp_path, n) for n in names)
    blob_a = open(a, "rb").read()
    assert blob_a.startswith(b"MLGPUCO2 ") and b"device-sources " in blob_a[:400] and b"\x7fELF" in blob_a
    # (1) a whole, valid entry of ANOTHER kernel under this kernel's name (what a hash collision would look like)
    blob_b = open(b, "rb").read()
    open(a, "wb").write(blob_b)
    open(b, "wb").write(blob_a)
    for k in ("0.5", "0.25"):
        r = run(tmp_path, k)
        assert r["stats"]["compiles"] == 1 and r["stats"]["disk_hits"] == 0, "a foreign entry must not be loaded"
    assert run(tmp_path)["sha"] == cold["sha"]
    # (2) the same entry written by another build of the library or another compiler
    for path in (a, b):
        data = open(path, "rb").read()
        open(path, "wb").write(data.replace(b"device-sources ", b"device-sources f", 1)[:len(data)])
    r = run(tmp_path)
    assert r["stats"]["compiles"] == 1 and r["sha"] == cold["sha"]
    # (3) a directory other users can write to is not trusted with code objects
    shared = tmp_path / "shared"
    shared.mkdir()
    os.chmod(shared, 0o777)
    r = run(shared)
    assert r["stats"]["compiles"] == 1 and not os.listdir(shared)
    # no stray temporaries left behind
    assert not [f for f in os.listdir(tmp_path) if ".co." in f and not f.endswith(".co.lock")]   # <entry>.lock: the cross-process build lock


def test_concurrent_builds_of_one_kernel_compile_once():
    """The host threads of a multi-device program ask for the same kernel at the same time (ml::gpu::DeviceGroup, bench.py
    --launcher threads): one of them runs hiprtc, the others get its code object."""
    prog = r'''
import json, threading
import madronalib_amd as ml
from madronalib_amd.constants import Op, Proc
codes = [None] * 4
def build(i):
    g = ml.Graph(ml.OfflineEngine(), 256)
    g.add("x", "input"); g.add("k", "const", value=0.625)
    g.add("hp", "proc", Proc.HIPASS, ["x"]); g.add("y", "op", Op.MULTIPLY, ["hp", "k"]); g.add_output("y")
    codes[i] = g.emit()[1]
threads = [threading.Thread(target=build, args=(i,)) for i in range(4)]
[t.start() for t in threads]; [t.join() for t in threads]
print(json.dumps(dict(stats=ml.jit_stats(), same=all(c == codes[0] for c in codes), n=len(codes[0]))))
'''
    env = dict(os.environ, MLGPU_CACHE_DIR="off", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["same"] and out["n"] > 1000
    assert out["stats"]["compiles"] == 1 and out["stats"]["memory_hits"] == 3, out["stats"]


def test_concurrent_rank_processes_compile_once(tmp_path):
    """The ranks of a multi-GPU job are PROCESSES that start together with a cold cache and ask for the same kernel: whoever
    takes the entry's lock file first runs hiprtc, the other seven wait in flock() and load what it wrote - one compile per
    machine, not per rank (profiles/archive/r03_multi_gpu_launch_paths.txt shows the same on a GPU box)."""
    prog = r'''
import json, sys, time
import madronalib_amd as ml
from madronalib_amd.constants import Op, Proc
start = float(sys.argv[1])
while time.time() < start:      # line the processes up so that they really do race
    time.sleep(0.001)
g = ml.Graph(ml.OfflineEngine(), 256)
g.add("x", "input"); g.add("k", "const", value=0.8125)
g.add("bp", "proc", Proc.BANDPASS, ["x"]); g.add("y", "op", Op.MULTIPLY, ["bp", "k"]); g.add_output("y")
code = g.emit()[1]
import hashlib
print(json.dumps(dict(stats=ml.jit_stats(), sha=hashlib.sha256(code).hexdigest())))
'''
    import time
    env = dict(os.environ, MLGPU_CACHE_DIR=str(tmp_path), PYTHONPATH=ROOT)
    start = str(time.time() + 3.0)
    procs = [subprocess.Popen([sys.executable, "-c", prog, start], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for _ in range(8)]
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0, se[-2000:]
        outs.append(json.loads(so.strip().splitlines()[-1]))
    assert len({o["sha"] for o in outs}) == 1
    assert sum(o["stats"]["compiles"] for o in outs) == 1, [o["stats"] for o in outs]
    assert sum(o["stats"]["disk_hits"] for o in outs) == 7


def test_offline_emit_of_round5_generator_forms():
    """Generated-kernel forms of round 5, compiled here with hiprtc for gfx950 (no device): delay rings in layout 2 - whole wavefronts
    and a bank whose last wavefront is not full (its spare lanes stay) -, layout 3's choice, an output that is the mixdown of all
    voices, and the register / LDS budgets the measured kernels rely on."""
    import madronalib_amd as ml
    from madronalib_amd import patches
    from madronalib_amd.constants import Op, Proc

    def strings(V, layout):
        desc = [dict(name="x", type="input"), dict(name="g", type="const", value=0.995),
                dict(name="fb", type="feedback", source="damp"),
                dict(name="fbg", type="op", kind=Op.MULTIPLY, inputs=["fb", "g"]),
                dict(name="sum", type="op", kind=Op.ADD, inputs=["x", "fbg"]),
                dict(name="line", type="proc", kind=Proc.FRACTIONAL_DELAY, inputs=["sum"], max_delay=1024.0),
                dict(name="damp", type="proc", kind=Proc.ONE_POLE, inputs=["line"])]
        return ml.Graph(ml.OfflineEngine(), V, desc, ["damp"], delay_windows=layout)

    def facts(code):
        notes = _code_object_notes(code)
        return (int(re.search(r"\.vgpr_count:\s+(\d+)", notes).group(1)), int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", notes).group(1)),
                int(re.search(r"\.group_segment_fixed_size:\s+(\d+)", notes).group(1)))

    g = strings(262144, 2)
    source, code = g.emit()
    vgpr, scratch, lds = facts(code)
    assert "MLGPU_RING_WINDOWS 2" in source and "__launch_bounds__(256, 4)" in source and "vr_0" not in source
    assert vgpr <= 128 and scratch <= 64 and lds == 4 * 40 * 64 * 4          # four wavefronts x (8 + 32) rows x 64 lanes: a CU holds four workgroups
    g.close()
    g = strings(1000, 2)                                                        # 1000 = 15 wavefronts + 40 voices
    source, code = g.emit()
    assert "const size_t v_0 = vr_0 < a.V ? vr_0 : a.V - 1;" in source and "(vr_0 >> 8)" in source and code[:4] == b"\x7fELF"
    g.close()
    # (round 6: layout 3 takes the sector trips - MLGPU_RING_WINDOWS 3 in the source, layout 4 of the API - for three rings and more)
    for V, n_delays, want in ((256, 2, 2), (256, 3, 3), (100, 1, 2), (256, 5, 3)):
        g = ml.Graph(ml.OfflineEngine(), V, delay_windows="best")
        g.add("x", "input")
        src = "x"
        for j in range(n_delays):
            g.add(f"d{j}", "proc", Proc.INTEGER_DELAY, [src], max_delay=300.0)
            src = f"d{j}"
        g.add_output(src)
        source, _ = g.emit()
        assert (f"MLGPU_RING_WINDOWS {want}" in source), (V, n_delays)
        g.close()
    # an output as the mixdown of all voices (mlgpu_graph_set_output_mixdown), next to a plain one, voices not a multiple of 64
    desc, outs = patches.synth16()
    g = ml.Graph(ml.OfflineEngine(), 1000, desc, outs, compile_now=False)
    g.add_output("lp")
    g.set_output_mixdown(0)
    source, code = g.emit()
    assert "mix64_park(mstrip0" in source and "mix64_sum_store(mstrip0" in source and "(vr_0 < a.V) ? y0_0" in source
    assert "__builtin_nontemporal_store(y1_0" in source and "__builtin_nontemporal_store(y0_0" not in source
    vgpr, scratch, lds = facts(code)
    assert lds == 4 * (64 * 20 + 64) * 4 and scratch == 0
    with pytest.raises(ml.MlgpuError):
        g.set_output_group_sum(0, 8)
    g.close()


def test_kernels_travel_to_an_installation_without_hiprtc(monkeypatch):
    """libmlgpu.so does not link hiprtc (round 6): where the compiler is missing, a generated kernel works when its code is there. A
    graph is compiled ahead of time here (no device), every generated kernel of the process is exported as a bundle, the process then
    behaves as an installation without the compiler (MLGPU_HIPRTC=off, disk cache off, memory cache emptied): the same graph fails
    with MLGPU_ERR_UNSUPPORTED and says why - and after the bundle's import it yields the same code object again."""
    import subprocess
    import madronalib_amd as ml
    from madronalib_amd import _lib
    from madronalib_amd.constants import Op, Proc
    L = _lib.load()
    out = subprocess.run(["ldd", os.path.join(os.path.dirname(_lib.__file__), "csrc", "libmlgpu.so")], capture_output=True, text=True).stdout
    assert "hiprtc" not in out, "libmlgpu.so must not link hiprtc: it is looked up at run time"

    def build():
        g = ml.Graph(ml.OfflineEngine(), 512)
        g.add("x", "input")
        g.add("c", "const", value=0.37251)
        g.add("y", "op", Op.MULTIPLY, ["x", "c"])
        g.add("lp", "proc", Proc.ONE_POLE, ["y"])
        g.add_output("lp")
        return g
    assert ml.jit_compiler_available()
    g = build()
    src, code = g.emit()
    g.close()
    bundle = ml.jit_cache_export()
    assert bundle[:8] == b"MLGPUKB1" and len(bundle) > len(code)
    monkeypatch.setenv("MLGPU_HIPRTC", "off")
    monkeypatch.setenv("MLGPU_CACHE_DIR", "off")
    assert not ml.jit_compiler_available()
    assert L.mlgpu_jit_cache_clear_memory() == 0
    g = build()
    with pytest.raises(ml.MlgpuError) as ei:
        g.emit()
    assert ei.value.status == ml.Status.ERR_UNSUPPORTED and "MLGPU_HIPRTC=off" in str(ei.value) and "mlgpu_jit_cache_export" in str(ei.value)
    g.close()
    n = ml.jit_cache_import(bundle)
    assert n >= 1
    g = build()
    src2, code2 = g.emit()
    assert src2 == src and code2 == code
    g.close()
    # a bundle that is not one, and one of another build of the device code
    with pytest.raises(ml.MlgpuError) as ei:
        ml.jit_cache_import(b"not a bundle at all")
    assert ei.value.status == ml.Status.ERR_INVALID
    fp_len = int.from_bytes(bundle[8:16], "little")
    other = bundle[:16] + bytes([bundle[16] ^ 1]) + bundle[17:]
    assert fp_len > 8
    with pytest.raises(ml.MlgpuError) as ei:
        ml.jit_cache_import(other)
    assert ei.value.status == ml.Status.ERR_UNSUPPORTED
