"""N>1 path on CPU: world_size-2 gloo processes shard a voice bank exactly like bench.py does on
GPUs (contiguous voice ranges, per-rank parameters from GLOBAL voice indices, no data-path
collective) and the union of the shards equals the unsharded computation. The per-shard compute
here is the CPU oracle (a checker standing in for the GPU kernel, which needs a GPU)."""
import os
import socket

import numpy as np
import pytest

from madronalib_amd.sharding import cfg3_voice_params, max_over_ranks, partition


def test_partition_covers_everything():
    for total in (1, 7, 64, 1000, 262144, 2097152):
        for world in (1, 2, 3, 8):
            spans = [partition(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        partition(10, 2, 2)


def test_partition_matches_baseline_config5():
    # 2 097 152 voices over 8 GPUs = 262 144 per GPU (BASELINE configs[4])
    assert [partition(2097152, 8, r) for r in (0, 7)] == [(0, 262144), (1835008, 2097152)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, T, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cpu_checkers import Oracle
    from madronalib_amd.constants import Proc
    orc = Oracle()
    lo, hi = partition(total, world, rank)
    freq, co = cfg3_voice_params(lo, hi, total, lambda om, k: orc.make_coeffs("bandpass", om, k))
    procs = [Proc.SAW_GEN, Proc.BANDPASS, Proc.GAIN]
    coeffs = np.ascontiguousarray(np.concatenate([co, np.full((1, hi - lo), 0.25, np.float32)], 0))
    st = orc.chain_clear(procs, hi - lo)
    dist.barrier()
    out = orc.chain_process(procs, T, coeffs, st, None, freq)
    dist.barrier()
    slowest = max_over_ranks(0.001 * (rank + 1), dist)   # timing reduction path
    ret[rank] = (lo, hi, out, st, slowest)
    dist.destroy_process_group()


def test_two_rank_sharding_equals_unsharded(oracle):
    import torch.multiprocessing as mp
    from madronalib_amd.constants import Proc
    total, T, world = 777, 3, 2   # ragged: ranks get 389 and 388 voices
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, T, ret), nprocs=world, join=True)
    assert sorted(ret.keys()) == [0, 1]
    # unsharded reference computation
    freq, co = cfg3_voice_params(0, total, total, lambda om, k: oracle.make_coeffs("bandpass", om, k))
    procs = [Proc.SAW_GEN, Proc.BANDPASS, Proc.GAIN]
    coeffs = np.ascontiguousarray(np.concatenate([co, np.full((1, total), 0.25, np.float32)], 0))
    st = oracle.chain_clear(procs, total)
    want = oracle.chain_process(procs, T, coeffs, st, None, freq)
    got = np.concatenate([ret[r][2] for r in range(world)], 0)
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
    assert (np.concatenate([ret[r][3] for r in range(world)], 1) == st).all()
    assert ret[0][1] == ret[1][0] == 389
    assert ret[0][4] == ret[1][4] == pytest.approx(0.002)   # MAX over ranks


def test_cfg5_parameters_do_not_depend_on_the_sharding():
    """BASELINE configs[4] (2 097 152 voices over 8 GPUs): every per-voice parameter, coefficient, seed and gate sample is a
    function of the GLOBAL voice index, so the union of the ranks' shards is the unsharded patch."""
    import madronalib_amd as ml
    from madronalib_amd.sharding import cfg5_gate_quad, cfg5_voice_params
    total, world = 4099, 8   # ragged
    full_p, full_c, full_s = cfg5_voice_params(0, total, total, ml)
    full_g = cfg5_gate_quad(0, total, 2)
    for rank in range(world):
        lo, hi = partition(total, world, rank)
        p, c, s = cfg5_voice_params(lo, hi, total, ml)
        for k in full_p:
            if np.ndim(full_p[k]):
                assert (p[k] == full_p[k][lo:hi]).all(), k
            else:
                assert p[k] == full_p[k]
        for k in full_c:
            assert (c[k].view(np.uint32) == full_c[k][:, lo:hi].view(np.uint32)).all(), k
        assert (s == full_s[lo:hi]).all()
        assert (cfg5_gate_quad(lo, hi, 2) == full_g[:, lo:hi, :]).all()


def _mix_worker(rank, world, port, total, T, blocks, ret):
    """One rank of a sharded bank whose host wants ONE channel: its shard's voices (the oracle standing in for the voice kernel) -> the
    rows of the mixdown tree at the shard's hand-over level (the oracle's restatement of mlgpu_bank_process_mixdown_shard) -> rank 0
    through RowExchange -> mlgpu_mixdown_finish (the library's host function: no device)."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import madronalib_amd as ml
    from cpu_checkers import Oracle
    from madronalib_amd.constants import Proc
    from madronalib_amd.rendezvous import GlooRendezvous
    from madronalib_amd.sharding import RowExchange
    orc = Oracle()
    lo, hi = partition(total, world, rank)
    freq, co = cfg3_voice_params(lo, hi, total, lambda om, k: orc.make_coeffs("bandpass", om, k))
    procs = [Proc.SAW_GEN, Proc.BANDPASS, Proc.GAIN]
    coeffs = np.ascontiguousarray(np.concatenate([co, np.full((1, hi - lo), 0.25, np.float32)], 0))
    st = orc.chain_clear(procs, hi - lo)
    rdv = GlooRendezvous(rank, world)      # (the process group is up already: it only wraps it)
    ex = RowExchange(rdv, rank, world, ml.mixdown_shard_rows(hi - lo), T)
    mixes = []
    for b in range(blocks):
        voices = orc.chain_process(procs, T, coeffs, st, None, freq)
        ex.put(b, orc.mixdown_shard(voices))
        if rank == 0:
            mixes.append(ml.mixdown_finish(ex.collect(b, T)))
    rdv.barrier()
    ex.close()
    ret[rank] = np.concatenate(mixes) if rank == 0 else None
    dist.destroy_process_group()


def test_two_rank_voice_sum_equals_one_bank(oracle):
    """The real-time block across the GPUs of a node ("per-GPU reduction then host add"), here with world_size-2 gloo
    processes on CPU: 2 x 4096 voices summed to one channel through per-shard tree rows + the host's finish give the bits of the
    unsplit 8192-voice mixdown, four blocks in a row through both slots of the exchange."""
    import torch.multiprocessing as mp
    from madronalib_amd.constants import Proc
    total, T, world, blocks = 8192, 1, 2, 4
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_mix_worker, args=(world, _free_port(), total, T, blocks, ret), nprocs=world, join=True)
    freq, co = cfg3_voice_params(0, total, total, lambda om, k: oracle.make_coeffs("bandpass", om, k))
    procs = [Proc.SAW_GEN, Proc.BANDPASS, Proc.GAIN]
    coeffs = np.ascontiguousarray(np.concatenate([co, np.full((1, total), 0.25, np.float32)], 0))
    st = oracle.chain_clear(procs, total)
    want = np.concatenate([oracle.mixdown(oracle.chain_process(procs, T, coeffs, st, None, freq)) for _ in range(blocks)])
    assert (ret[0].view(np.uint32) == want.view(np.uint32)).all()
    assert np.abs(want).max() > 0
