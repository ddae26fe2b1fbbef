"""Pin the plain-C oracle against (a) the committed golden vectors produced by the compiled
reference, (b) the SURVEY Appendix-B anchors, (c) the reference's own test assertions
(Tests/dspOpsTest.cpp:103-104,154,164; Tests/dspGensTest.cpp:31). Needs no reference tree."""
import os

import numpy as np
import pytest

from golden_cases import (ANCHORS, chain_case, chain_case_names, hash31, load_chains, load_multi, load_ops,
                          multi_golden_case)
from inputs import MULTI_CASES, assert_bits_equal, assert_rel_close, is_float_result, multi_inputs_audio
from madronalib_amd.constants import Op, Proc, RowOp, Vop

from golden_cases import load_rows, rows_golden_case  # noqa: E402
from rows_cases import ROWS_CASES, run as run_rows_case  # noqa: E402

HW_REL = 2.0 ** -11


@pytest.mark.parametrize("op", Op.UNARY + Op.BINARY + Op.TERNARY)
def test_ops_golden(oracle, op):
    d = load_ops()
    g = lambda k: d[f"op{op}_{k}"] if f"op{op}_{k}" in d.files else None  # noqa: E731
    got = oracle.op(op, g("a"), g("b"), g("c"))
    want = d[f"op{op}_out"]
    if op in Op.HW_APPROX:
        ok = np.isfinite(want.view(np.float32)) & np.isfinite(got.view(np.float32)) & (want.view(np.float32) != 0)
        assert_rel_close(got.view(np.float32)[ok], want.view(np.float32)[ok], HW_REL, f"op {op}")
    else:
        assert_bits_equal(got, want, is_float_result(op), f"op {op}")


def test_row_reduce_golden(oracle):
    d = load_ops()
    for ro in (RowOp.SUM, RowOp.MEAN, RowOp.MAX, RowOp.MIN):
        assert_bits_equal(oracle.row_reduce(ro, d["rows"]), d[f"rowop{ro}"], True, f"rowop {ro}")


@pytest.mark.parametrize("name", chain_case_names())
def test_chains_golden(oracle, name):
    c = chain_case(load_chains(), name)
    st = c["state0"].copy()
    hw = any(p in Proc.HW_APPROX for p in c["procs"])
    for out_k, st_k in (("out1", "state1"), ("out2", "state2")):
        got = oracle.chain_process(c["procs"], c[out_k].shape[1] // 64, c["coeffs"], st, c["in_signal"], c["in_const"])
        if hw:
            assert_rel_close(got, c[out_k], HW_REL, name)
        else:
            assert_bits_equal(got, c[out_k], True, name + " " + out_k)
        assert_bits_equal(st, c[st_k], False, name + " " + st_k)


@pytest.mark.parametrize("name", MULTI_CASES)
def test_multi_input_forms_golden(oracle, name):
    c = multi_golden_case(load_multi(), name)
    T = c["out"][0].shape[1] // 64
    ins = multi_inputs_audio(c, 2 * T)
    st = c["state0"].copy()
    for call in range(2):
        sl = slice(call * 64 * T, (call + 1) * 64 * T)
        got = oracle.proc_multi(c["kind"], T, c["coeffs"], st, [np.ascontiguousarray(x[:, sl]) for x in ins])
        assert_bits_equal(got, c["out"][call], True, f"{name} out{call + 1}")
        assert_bits_equal(st, c["state"][call], False, f"{name} state{call + 1}")


def test_vector_generators_golden(oracle):
    d = load_multi()
    V, T = d["vop_a"].shape
    for vop in (Vop.COLUMN_INDEX, Vop.RANGE_OPEN, Vop.RANGE_CLOSED, Vop.INTERPOLATE_LINEAR):
        got = oracle.vop(vop, V, T, np.repeat(d["vop_a"], 64, 1), np.repeat(d["vop_b"], 64, 1))
        assert_bits_equal(got, d[f"vop{vop}_out"], True, f"vop {vop}")


def test_restated_libm_sinf_against_host_libm(oracle):
    """The device's sinf (Lopass::makeCoeffsVec, MLDSPFilters.h:104-113) is glibc 2.35's algorithm restated
    (mlorc_libm_sinf is the same restatement on the CPU). Checked here against the host libm over ALL 2^32
    float bit patterns: identical except where glibc's x86-64 FMA ifunc variant rounds differently from the
    plain-double algorithm — 12 arguments, all with 53 < |x| < 120 — and never for |x| <= pi, the only
    range the SVF coefficient code uses (omega <= 0.5 after the clamp)."""
    pi_bits = int(np.float32(3.2).view(np.uint32))
    n, lst = oracle.sinf_check(0, pi_bits)                      # +x in [0, 3.2]
    assert n == 0, [hex(x) for x in lst]
    n, lst = oracle.sinf_check(0x80000000, 0x80000000 + pi_bits)  # -x
    assert n == 0, [hex(x) for x in lst]
    n, lst = oracle.sinf_check(0, 0xFFFFFFFF)
    xs = np.abs(lst.view(np.float32))
    assert n <= 12 and ((xs > 53.0) & (xs < 120.0)).all(), (n, [hex(x) for x in lst])
    # exactly these twelve on a host whose glibc picks the FMA variant, none on one that does not (INTEGRATION.md lists them;
    # tests/test_gpu_deviations.py runs the device on each)
    twelve = [0x4255b0a9, 0x42a35c07, 0x42a35d44, 0x42a97360, 0x42cf5854, 0x42e87a55, 0xc255b0a9, 0xc2a35c07, 0xc2a35d44, 0xc2a97360, 0xc2cf5854, 0xc2e87a55]
    assert n == 0 or sorted(int(x) for x in lst) == sorted(twelve), [hex(x) for x in lst]


def test_fast_sinf_forms_exhaustively(oracle):
    """Lopass(x, omega, k) on the device computes its two sinf per sample with cheaper sequences than glibc's (Horner forms with
    fused multiply-adds, the quadrant from two float comparisons; mldsp_math.hpp: libm_sinf_q0 - the sine polynomial on the argument
    itself - for arguments in [2^-12, kSinfT1 = 0.785...), libm_sinf_pair for [2^-12, pi_f]) - legitimate only because their
    rounded floats equal the host libm's for EVERY argument of those domains. The same sequences in C over every float of both
    domains (the direct form up to the float before kSinfT1: the device uses it that far, not just to glibc's own 0.75); and the
    two thresholds are where glibc's quadrant steps."""
    def bits(x):
        return int(np.float32(x).view(np.uint32))
    lo, direct_end, pi_f = bits(2.0 ** -12), 0x3F490FDB, bits(np.float32(np.pi))     # direct_end = kSinfT1
    assert pi_f == 0x40490FDB and bits(np.float32(np.pi) * np.float32(0.5) * np.float32(2.0)) == pi_f   # the largest argument: 2 (pi_f * 0.5)
    n, first = oracle.sinf_fast_check(0, lo, direct_end - 1)
    assert n == 0, hex(first)
    n, first = oracle.sinf_fast_check(1, lo, pi_f)
    assert n == 0, hex(first)
    for thr, q in ((0x3F490FDB, 1), (0x4016CBE4, 2)):       # kSinfT1, kSinfT2
        below, at = np.uint32(thr - 1).view(np.float32), np.uint32(thr).view(np.float32)
        assert oracle.sinf_quadrant(below) == q - 1 and oracle.sinf_quadrant(at) == q
    src = open(os.path.join(os.path.dirname(__file__), "..", "madronalib_amd", "csrc", "mldsp_math.hpp")).read()
    assert "kSinfT1 = 0x1.921fb6p-1f, kSinfT2 = 0x1.2d97c8p+1f, kSinfMax = 0x1.921fb6p+1f" in src
    assert float.fromhex("0x1.921fb6p-1") == float(np.uint32(0x3F490FDB).view(np.float32))
    assert float.fromhex("0x1.2d97c8p+1") == float(np.uint32(0x4016CBE4).view(np.float32))
    assert float.fromhex("0x1.921fb6p+1") == float(np.uint32(pi_f).view(np.float32))


@pytest.mark.parametrize("name", list(ROWS_CASES))
def test_row_plumbing_and_routing_golden(oracle, name):
    ins, want = rows_golden_case(load_rows(), name)
    assert_bits_equal(run_rows_case(oracle, name, ins), want, True, name)


def test_impulse_table_golden(oracle):
    assert_bits_equal(oracle.impulse_table(), load_chains()["impulse_table"], True, "ImpulseGen table")


# ---- SURVEY Appendix B anchors ----------------------------------------------------------

def test_anchor_cfg1(oracle):
    a = ANCHORS["cfg1"]
    co = oracle.make_coeffs("lopass", *a["lopass"]).reshape(3, 1).copy()
    st = oracle.chain_clear(a["procs"], 1)
    assert st[0, 0] == 0xC0000000
    y = oracle.chain_process(a["procs"], 1, co, st, None, np.array([a["freq"]], np.float32))[0]
    assert list(y[:4]) == a["y0_3"] and y[63] == a["y63"]


def test_anchor_cfg3(oracle):
    a = ANCHORS["cfg3"]
    co = np.concatenate([oracle.make_coeffs("bandpass", *a["bandpass"]), [a["gain"]]]).astype(np.float32).reshape(4, 1).copy()
    st = oracle.chain_clear(a["procs"], 1)
    z = oracle.chain_process(a["procs"], 1, co, st, None, np.array([a["freq"]], np.float32))[0]
    assert list(z[:4]) == a["z0_3"] and z[63] == a["z63"]


def test_anchor_saw_noise_onepole_cfg4(oracle):
    st = oracle.chain_clear([Proc.SAW_GEN], 1)
    w = oracle.chain_process([Proc.SAW_GEN], 3, np.zeros((0, 1), np.float32), st, None, np.array([440.0 / 48000.0], np.float32))
    wb = w.view(np.uint32)[0, 128:]
    assert (wb[0], wb[63]) == (ANCHORS["saw_vec3"]["w0"], ANCHORS["saw_vec3"]["w63"])

    st = oracle.chain_clear([Proc.NOISE_GEN], 1)
    n = oracle.chain_process([Proc.NOISE_GEN], 1, np.zeros((0, 1), np.float32), st).view(np.uint32)[0]
    assert (n[0], n[63]) == (ANCHORS["noise_vec0"]["n0"], ANCHORS["noise_vec0"]["n63"])

    a = ANCHORS["onepole"]
    c = oracle.make_coeffs("onepole", a["omega"])
    assert tuple(c.view(np.uint32)) == (a["a0"], a["b1"])
    imp = np.zeros((1, 64), np.float32)
    imp[0, 0] = 1.0
    st = oracle.chain_clear([Proc.ONE_POLE], 1)
    o = oracle.chain_process([Proc.ONE_POLE], 1, c.reshape(2, 1).copy(), st, imp).view(np.uint32)[0]
    assert (o[1], o[63]) == (a["o1"], a["o63"])

    procs = [Proc.NOISE_GEN] + [Proc.LOPASS] * 8
    co = np.concatenate([oracle.make_coeffs("lopass", float(np.float32(0.02) * np.float32(i + 1)), 0.7) for i in range(8)]).reshape(24, 1).copy()
    st = oracle.chain_clear(procs, 1)
    y = oracle.chain_process(procs, 4, co, st).view(np.uint32)[0, 192:]
    a = ANCHORS["cfg4_vec4"]
    assert (y[0], y[31], y[63]) == (a["y0"], a["y31"], a["y63"])


def test_anchor_ramp(oracle):
    a = oracle.range_closed(-np.pi, np.pi)
    for op, bits in ANCHORS["ramp_elem5"].items():
        assert oracle.op(op, a)[5] == bits, op
    for op, h in ANCHORS["ramp_hash"].items():
        assert hash31(oracle.op(op, a)) == h, op
    a2 = (a * a + np.float32(0.1)).astype(np.float32)
    for op, h in ANCHORS["ramp_hash_log"].items():
        assert hash31(oracle.op(op, a2)) == h, op


# ---- the reference's own assertions -------------------------------------------------------

def test_reference_precision_thresholds(oracle):
    """Tests/dspOpsTest.cpp:85-105: max|libm - precise| < 2e-6, max|libm - approx| < 2e-4 on
    rangeClosed(-pi, pi). (For log the reference's max() drops the NaNs of x<=0; we compare x>0.)"""
    a = oracle.range_closed(-np.pi, np.pi)
    for precise, approx, native in ((Op.SIN, Op.SIN_APPROX, np.sin), (Op.COS, Op.COS_APPROX, np.cos),
                                    (Op.EXP, Op.EXP_APPROX, np.exp)):
        nat = native(a.astype(np.float64))
        assert np.abs(oracle.op(precise, a).view(np.float32) - nat).max() < 2e-6
        assert np.abs(oracle.op(approx, a).view(np.float32) - nat).max() < 2e-4
    pos = a[a > 0]
    pos = np.resize(pos, 64)
    assert np.abs(oracle.op(Op.LOG, pos).view(np.float32) - np.log(pos.astype(np.float64))).max() < 2e-6
    assert np.abs(oracle.op(Op.LOG_APPROX, pos).view(np.float32) - np.log(pos.astype(np.float64))).max() < 2e-4
    assert np.isnan(oracle.op(Op.LOG, a[a <= 0][:1].repeat(64)).view(np.float32)).all()


def test_reference_lerp_and_fractional_part(oracle):
    """Tests/dspOpsTest.cpp:148-165."""
    idx = np.arange(64, dtype=np.float32)
    r = oracle.op(Op.LERP, idx, np.zeros(64, np.float32), np.full(64, 0.5, np.float32)).view(np.float32)
    assert r[63] == 31.5
    p = oracle.op(Op.FRACTIONAL_PART, np.full(64, 1.25, np.float32)).view(np.float32)
    n = oracle.op(Op.FRACTIONAL_PART, np.full(64, -1.25, np.float32)).view(np.float32)
    assert p[63] == -n[63]


def test_reference_sinegen_cycle(oracle):
    """Tests/dspGensTest.cpp:24-31: SineGen after clear(), one cycle at 1/64, ends within -120 dB of 0."""
    st = oracle.chain_clear([Proc.SINE_GEN], 1)
    v = oracle.chain_process([Proc.SINE_GEN], 1, np.zeros((0, 1), np.float32), st, None, np.array([1.0 / 64], np.float32))[0]
    assert abs(v[63]) < 10.0 ** (-120.0 / 20.0)


@pytest.mark.parametrize("which", ["up", "down"])
def test_rate_functions_golden(oracle, which):
    """tests/golden/regions.npz: outputs of the reference's Upsample2xFunction / Downsample2xFunction objects."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "regions.npz"))
    got = oracle.rate_function_run(which == "up", g["freq"], g["co"], g["x"], g["m"])
    assert (got.view(np.uint32) == g[which].view(np.uint32)).all()
    assert np.abs(g[which]).max() > 0.1


def test_synth16_golden(oracle):
    """tests/golden/synth16.npz: BASELINE configs[4]'s voice run with the reference's objects; the evaluator reproduces it."""
    import os
    from graph_oracle import evaluate
    from madronalib_amd import patches
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "synth16.npz"))
    V, T = g["gate"].shape[0], g["gate"].shape[1] // 64
    params = {k[2:]: (g[k] if g[k].ndim else float(g[k])) for k in g.files if k.startswith("p_")}
    coeffs = {k[2:]: g[k] for k in g.files if k.startswith("c_")}
    desc, outs = patches.synth16()
    states = {n["name"]: oracle.chain_clear([n["kind"]], V) for n in desc if n["type"] == "proc"}
    states["noise"][0] = g["seeds"]
    (got,) = evaluate(oracle, desc, outs, V, T, {"gate": g["gate"]}, params, coeffs, states)
    assert (got.view(np.uint32) == g["out"].view(np.uint32)).all()


def test_synth16full_golden(oracle):
    """tests/golden/synth16full.npz: the patch SURVEY 8d lists (filter envelope, per-sample cutoff through exp2Approx,
    Lopass(x, omega, k) with its two libm sinf per sample) run with the reference's own objects; the node-by-node evaluator
    over the plain-C oracle reproduces it bit for bit."""
    import os
    from graph_oracle import evaluate
    from madronalib_amd import patches
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "synth16full.npz"))
    V, T = g["gate"].shape[0], g["gate"].shape[1] // 64
    params = {k[2:]: (g[k] if g[k].ndim else float(g[k])) for k in g.files if k.startswith("p_")}
    coeffs = {k[2:]: g[k] for k in g.files if k.startswith("c_")}
    desc, outs = patches.synth16(full=True)
    assert sum(n["type"] in ("proc", "op") for n in desc) == 22
    states = {n["name"]: oracle.chain_clear([n["kind"]], V) for n in desc if n["type"] == "proc"}
    states["noise"][0] = g["seeds"]
    (got,) = evaluate(oracle, desc, outs, V, T, {"gate": g["gate"]}, params, coeffs, states)
    assert (got.view(np.uint32) == g["out"].view(np.uint32)).all()
    assert np.abs(g["out"]).max() > 0.5


@pytest.mark.parametrize("V,shards", [(128, 2), (4096, 2), (8192, 2), (8192, 8), (64 * 64 * 6, 3), (64 * 72, 3), (64 * 64 * 64 // 2, 2)])
def test_mixdown_shards_equal_one_bank(oracle, V, shards):
    """A voice bank split over several engines (mlgpu_bank_process_mixdown_shard + mlgpu_mixdown_finish): every shard hands over its
    rows of the mixdown tree at the highest level its voice count is whole at, the host finishes the same tree - the oracle's
    restatement of both halves gives the bits of the unsplit mixdown, and the library's host function (no device needed) the same."""
    import madronalib_amd as ml
    from inputs import lcg_noise
    T = 1
    sig = lcg_noise(np.arange(V, dtype=np.uint32) + 17, 64 * T)
    want = oracle.mixdown(sig)
    per = V // shards
    assert ml.mixdown_shard_level(per) >= 1 and ml.mixdown_shard_rows(per) == per // 64 ** ml.mixdown_shard_level(per)
    rows = np.concatenate([oracle.mixdown_shard(sig[k * per:(k + 1) * per]) for k in range(shards)], 0)
    assert rows.shape[0] == shards * ml.mixdown_shard_rows(per)
    got = oracle.mixdown_rows(rows)
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
    lib = ml.mixdown_finish(rows)
    assert (lib.view(np.uint32) == want.view(np.uint32)).all()
    assert ml.mixdown_shard_level(100) == 0 and ml.mixdown_shard_rows(100) == 0


def test_mixdown_finish_many_rows_and_flush(oracle):
    """mlgpu_mixdown_finish over more than 64 and more than 4096 rows (the passes through its scratch), and in flush mode."""
    import madronalib_amd as ml
    rng = np.random.default_rng(4)
    for n in (1, 63, 64, 65, 4096, 4097, 5000):
        rows = rng.standard_normal((n, 64)).astype(np.float32)
        got = ml.mixdown_finish(rows)
        assert (got.view(np.uint32) == oracle.mixdown_rows(rows).view(np.uint32)).all(), n
    tiny = np.full((3, 64), 1e-39, np.float32)          # denormals: kept in the default mode, +0 in flush mode
    assert (ml.mixdown_finish(tiny) == np.float32(3e-39)).all()
    assert (ml.mixdown_finish(tiny, flush_denormals=True).view(np.uint32) == 0).all()
