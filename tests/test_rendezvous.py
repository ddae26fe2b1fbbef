"""The N>1 plumbing of bench.py without a GPU: the rendezvous objects (madronalib_amd/rendezvous.py) that line up
ranks — processes through files, threads through a barrier, torch.distributed.run ranks through gloo — and the
launcher's refusal to run fewer ranks than asked for. The path has no data collective (voices share nothing), so this
is ALL the multi-GPU coordination there is; sharding itself is covered by tests/test_sharding_gloo.py (CPU) and
tests/cpp/multi_engine_test.cpp (GPU)."""
import json
import multiprocessing as mp
import os
import socket
import subprocess
import sys
import tempfile
import threading

import numpy as np
import pytest

from madronalib_amd import rendezvous
from madronalib_amd.sharding import partition

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _file_rank(directory, rank, world, q):
    r = rendezvous.FileRendezvous(directory, rank, world, timeout_s=60)
    r.barrier()
    got = r.gather({"rank": rank, "span": list(partition(1000, world, rank))})
    slowest = r.max(0.01 * (rank + 1))
    r.barrier()
    q.put((rank, got, slowest))


def test_file_rendezvous_three_processes():
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with tempfile.TemporaryDirectory(prefix="mlgpu_rdv_test_") as d:
        ps = [ctx.Process(target=_file_rank, args=(d, r, world, q)) for r in range(world)]
        for p in ps:
            p.start()
        res = sorted(q.get(timeout=120) for _ in range(world))
        for p in ps:
            p.join(60)
            assert p.exitcode == 0
    for rank, got, slowest in res:
        assert [g["rank"] for g in got] == list(range(world))          # gather is rank-ordered on every rank
        assert got[0]["span"][0] == 0 and got[-1]["span"][1] == 1000
        assert slowest == pytest.approx(0.03)


def test_file_rendezvous_abort_releases_waiters():
    with tempfile.TemporaryDirectory(prefix="mlgpu_rdv_test_") as d:
        a = rendezvous.FileRendezvous(d, 0, 2, timeout_s=30)
        b = rendezvous.FileRendezvous(d, 1, 2, timeout_s=30)
        err = []

        def wait():
            try:
                a.barrier()
            except RuntimeError as ex:
                err.append(str(ex))
        t = threading.Thread(target=wait)
        t.start()
        b.abort()
        t.join(30)
        assert err and "aborted" in err[0]


def test_file_rendezvous_eight_processes():
    """The width of one node: 8 rank processes through one directory, many collectives back to back."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with tempfile.TemporaryDirectory(prefix="mlgpu_rdv_test_") as d:
        ps = [ctx.Process(target=_file_rank, args=(d, r, world, q)) for r in range(world)]
        for p in ps:
            p.start()
        res = sorted(q.get(timeout=180) for _ in range(world))
        for p in ps:
            p.join(60)
            assert p.exitcode == 0
    for rank, got, slowest in res:
        assert [g["rank"] for g in got] == list(range(world))
        spans = [g["span"] for g in got]
        assert spans[0][0] == 0 and spans[-1][1] == 1000 and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert slowest == pytest.approx(0.08)


_RANK_PROG = r"""
import os, signal, sys, time
sys.path.insert(0, {root!r})
from madronalib_amd import rendezvous
r = rendezvous.from_environment()
rank = r.rank
r.barrier()
if rank == 5 and {mode!r} == "exception":
    raise SystemExit(3)                      # a refusal / an exception: the rank ends with a status, nothing else
if rank == 5 and {mode!r} == "segfault":
    os.kill(os.getpid(), signal.SIGSEGV)     # dies hard: no abort file, no exit handler
got = r.gather(rank)
r.barrier()
if rank == 0:
    print("ranks", got)
"""


@pytest.mark.parametrize("mode", ["ok", "exception", "segfault"])
def test_launcher_releases_every_rank_when_one_dies(mode):
    """bench.py's own launcher (run_rank_processes) with 8 ranks: all fine -> rank 0's line comes back; rank 5 fails after the
    first barrier, by exit status or by a segfault that writes nothing -> the seven ranks waiting in the next collective
    leave within seconds (the launcher writes the abort file), instead of sitting out the rendezvous timeout."""
    import time
    prog = _RANK_PROG.format(root=ROOT, mode=mode)
    t0 = time.monotonic()
    rcs, out0 = rendezvous.run_rank_processes([sys.executable, "-c", prog], 8, grace_s=20.0)
    dt = time.monotonic() - t0
    if mode == "ok":
        assert rcs == [0] * 8 and "ranks [0, 1, 2, 3, 4, 5, 6, 7]" in out0
    else:
        assert rcs[5] != 0
        assert all(rc != 0 for rc in rcs), rcs          # nobody reports success for a run that lost a rank
        assert dt < 60, f"ranks were not released ({dt:.0f} s)"


def test_thread_rendezvous():
    world = 4
    group = rendezvous.ThreadRendezvous.group(world)
    out = [None] * world

    def body(r):
        group[r].barrier()
        g1 = group[r].gather({"rank": r})
        g2 = group[r].gather({"rank": 10 * r})        # back-to-back gathers do not overwrite each other
        out[r] = (g1, g2, group[r].max(float(r)))
    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(30)
    for r in range(world):
        g1, g2, m = out[r]
        assert [x["rank"] for x in g1] == [0, 1, 2, 3] and [x["rank"] for x in g2] == [0, 10, 20, 30] and m == 3.0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gloo_rank(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.pop("MLGPU_RDV_DIR", None)
    r = rendezvous.from_environment()
    assert isinstance(r, rendezvous.GlooRendezvous)
    r.barrier()
    q.put((rank, r.gather({"rank": rank}), r.max(1.0 + rank)))
    r.close()


def test_gloo_rendezvous_two_ranks():
    """The form the driver uses (torch.distributed.run exports RANK / WORLD_SIZE / MASTER_*): gloo on CPU, no RCCL."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_gloo_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, got, slowest in res:
        assert [g["rank"] for g in got] == [0, 1] and slowest == 2.0


def test_from_environment_picks_files_when_a_directory_is_exported(monkeypatch, tmp_path):
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "1")
    monkeypatch.setenv("MLGPU_RDV_DIR", str(tmp_path))
    r = rendezvous.from_environment()
    assert isinstance(r, rendezvous.FileRendezvous) and r.rank == 1 and r.world == 2
    monkeypatch.setenv("WORLD_SIZE", "1")
    assert isinstance(rendezvous.from_environment(), rendezvous.SoloRendezvous)


def test_bench_refuses_to_run_without_enough_gpus():
    """`python bench.py --gpus N` never silently runs fewer ranks: here (no GPU) every N fails with a message and rc != 0;
    the same check compares N with the visible device count on a GPU box (gpurun log in profiles/archive/r02_multi_gpu.txt)."""
    from madronalib_amd import _lib
    if _lib.load().mlgpu_device_count() > 0:
        pytest.skip("a GPU is visible")
    for n in ("1", "2", "8"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", n, "--steps", "1", "--warmup", "0"],
                           capture_output=True, text=True, timeout=120)
        assert r.returncode != 0
        assert "needs a GPU" in r.stderr + r.stdout
    # a launcher that exported a different world size than --gpus is an error too, not a warning
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0


def test_pmc_record_is_keyed_by_workload_and_size():
    """roofline.traffic comes from the PMC passes of exactly the measured case, never from a kernel-name match."""
    sys.path.insert(0, ROOT)
    import bench
    key = bench.workload_key("cfg3", 262144, 30)
    assert key == "cfg3:262144x30"
    os.environ["MLGPU_DELAY_WINDOWS"] = "1"
    try:
        assert bench.workload_key("strings", 262144, 16) == "strings:delay_windows=1:262144x16"
    finally:
        del os.environ["MLGPU_DELAY_WINDOWS"]
    with open(os.path.join(ROOT, "profiles", "pmc_workloads.json")) as f:
        table = json.load(f)["workloads"]
    for k, rec in table.items():
        assert rec["hbm_bytes_per_launch"] > 0 and ":" in k
    assert bench.pmc_record("strings:262144x16") is None or bench.pmc_record("strings:262144x16") != bench.pmc_record("cfg5:262144x16")
    assert bench.pmc_record("no-such-workload:1x1") is None
    assert np.isfinite(bench.VALU_PEAK_LANE_INST)
