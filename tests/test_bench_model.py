"""bench.py's accounting helpers, on the CPU: the issue-time model behind roofline.valu.busy_frac and the core count."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402


def record(**kw):
    r = {"valu_wave_insts_per_launch": 1000.0, "SQ_INSTS_VALU_ADD_F32": 400.0, "SQ_INSTS_VALU_MUL_F32": 300.0, "SQ_INSTS_VALU_FMA_F32": 100.0,
         "SQ_INSTS_VALU_CVT": 50.0, "SQ_INSTS_VALU_TRANS_F32": 10.0, "SQ_INSTS_VALU_INT32": 40.0}
    r.update(kw)
    return r


def test_busy_fraction_prices_the_classes():
    ms = 1e-6  # span = 1024 SIMDs x 1 ns
    lo, hi = bench.valu_busy(record(), ms)["busy_frac"]
    n = bench.ISSUE_NS
    known = 800 * n["plain"] + 50 * n["slow"] + 10 * n["trans"] + 40 * n["plain"]
    assert lo == pytest.approx((known + 100 * n["plain"]) / 1024.0)
    assert hi == pytest.approx((known + 100 * n["slow"]) / 1024.0)


def test_packed_fp32_is_priced_at_the_slow_rate():
    ms = 1e-6
    plain = bench.valu_busy(record(), ms)["busy_frac"]
    packed = bench.valu_busy(record(packed_f32_share=1.0), ms)
    assert packed["packed_f32_share"] == 1.0
    n = bench.ISSUE_NS
    assert packed["busy_frac"][0] - plain[0] == pytest.approx(800 * (n["slow"] - n["plain"]) / 1024.0)
    # two wavefronts per SIMD: the high figure moves to the slower two-wavefront rates, the low one stays
    two = bench.valu_busy(record(packed_f32_share=1.0), ms, waves_per_simd=2.0)["busy_frac"]
    assert two[0] == packed["busy_frac"][0] and two[1] > packed["busy_frac"][1]


def test_no_class_counters_no_busy_fraction():
    assert bench.valu_busy({"valu_wave_insts_per_launch": 10.0}, 1.0) is None
    assert bench.valu_busy({}, 1.0) is None


def test_usable_cores_is_within_the_affinity_mask():
    n = bench.usable_cores()
    assert 1 <= n <= len(os.sched_getaffinity(0))


def test_committed_counter_records_name_their_build():
    recs = json.load(open(os.path.join(ROOT, "profiles", "pmc_workloads.json")))["workloads"]
    for case in ("cfg2:65536x1", "cfg2:4194304x1", "cfg3:262144x30", "cfg4:131072x32", "cfg5:262144x16", "cfg5full:262144x16"):
        assert case in recs and recs[case].get("device_source_hash"), case
        assert recs[case].get("fetch_bytes_per_launch_raw") is not None and recs[case].get("valu_wave_insts_per_launch")


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"), reason="no llvm-objdump")
def test_packed_share_of_the_shipped_cascade_kernel():
    import kernel_mix
    lib = os.path.join(ROOT, "madronalib_amd", "csrc", "libmlgpu.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    share = kernel_mix.packed_share("void mldev::cascade_lanes_kernel<16, 8, 1, 8, 2, true>(ChainArgs)", lib)
    assert share is not None and 0.75 < share < 0.95          # 39 packed + 2 plain arithmetic instructions per tick, plus the edges
    assert kernel_mix.packed_share("void mldev::chain_kernel<mldev::Chain<2, 18, 48>, false>(ChainArgs)", lib) == 0.0
    assert kernel_mix.packed_share("mlgpu_graph_kernel", lib) is None   # compiled at run time: not in the library


def test_slot_model_brackets_between_all_plain_paired_and_none():
    """With the launch's shader cycles (GRBM_GUI_ACTIVE / 8) the classes are priced in 4-cycle issue slots: a bracket from "every plain
    instruction issued beside another" to "none", never above 1."""
    b = bench.valu_busy(record(), 123.0, cycles=10.0)["busy_frac"]
    slots = 1024 * 10.0 / 4.0
    fixed = 50 * 1.0 + 10 * 2.0            # conversions a slot, transcendentals two
    plain = 800 + 40                       # FP32 add / mul / fma and integer
    assert b[0] == pytest.approx((fixed + (plain + 100) * 0.5) / slots)
    assert b[1] == pytest.approx(min(1.0, (fixed + plain * 1.0 + 100 * 1.0) / slots))
    assert bench.valu_busy(record(), 123.0, cycles=0.5)["busy_frac"][1] == 1.0
