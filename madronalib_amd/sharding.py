"""Multi-GPU partitioning of a voice bank: independent contiguous voice ranges, one per rank.

The hot path shards embarrassingly (reference: every Bank row / Synth voice owns private state,
MLDSPFunctional.h:324,334; source/app/MLSynth.h:49-57), so there is NO data-path collective:
rank g creates its own engine on its own GPU and processes voices [lo, hi). torch.distributed
(RCCL on GPUs, gloo in the CPU tests) is used only to line ranks up and to reduce the timing.
"""
import numpy as np


def partition(total_voices, world, rank):
    """Contiguous voice range [lo, hi) owned by `rank`; sizes differ by at most one voice."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    per, rem = divmod(int(total_voices), int(world))
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)


def cfg3_voice_params(lo, hi, total, make_bandpass_coeffs):
    """BASELINE configs[2] per-voice parameters for GLOBAL voices [lo, hi) of `total`
    (SURVEY §8d): freq[v] = 55*2^(5 v/V)/48000, Bandpass(omega=min(0.45, 4 freq), k=0.5).
    Returns (freq [n], coeffs [3][n]). `make_bandpass_coeffs(omega, k)` -> 3 floats (host libm)."""
    v = np.arange(lo, hi, dtype=np.float64)
    freq = (55.0 * 2.0 ** (5.0 * v / total) / 48000.0).astype(np.float32)
    om = np.minimum(0.45, 4.0 * freq.astype(np.float64)).astype(np.float32)
    uniq, inv = np.unique(om, return_inverse=True)
    table = np.stack([np.asarray(make_bandpass_coeffs(float(o), 0.5), np.float32) for o in uniq]) if len(uniq) else np.zeros((0, 3), np.float32)
    return freq, np.ascontiguousarray(table[inv].T.reshape(3, -1))


def cfg5_voice_params(lo, hi, total, ml, full=False):
    """BASELINE configs[4] (synth16 patch, madronalib_amd/patches.py) per-voice parameters for GLOBAL voices
    [lo, hi) of `total`. Every value is a pure function of the global voice index, so any sharding of the
    voice range gives the same patch per voice. Coefficients come from small tables (host libm makers are
    called once per distinct value). Returns (params {name: [n] or scalar}, coeffs {node: [NC][n]}, seeds [n])."""
    v = np.arange(lo, hi, dtype=np.int64)
    u = v.astype(np.float64) / max(1, int(total))
    params = dict(pitch=(5.0 * u - 2.0).astype(np.float32), baseFreq=np.float32(110.0 / 48000.0),
                  width=(0.1 + 0.8 * ((v * 37) % 101) / 100.0).astype(np.float32),
                  lfoFreq=((0.1 + 7.9 * ((v * 53) % 97) / 96.0) / 48000.0).astype(np.float32),
                  noiseLevel=(0.3 * ((v * 29) % 89) / 88.0).astype(np.float32))

    def table(maker, n, fn):
        t = np.stack([np.atleast_1d(np.asarray(maker(*fn(i / (n - 1.0))), np.float32)) for i in range(n)])
        return t

    def pick(t, mult, n):
        return np.ascontiguousarray(t[(v * mult) % n].T)
    coeffs = dict(
        lp=pick(table(ml.Lopass.makeCoeffs, 64, lambda x: (0.01 + 0.29 * x, 0.3 + 1.2 * (1.0 - x))), 7, 64),
        hp=pick(table(ml.Hipass.makeCoeffs, 61, lambda x: (0.0005 + 0.0095 * x, 0.7 + 0.8 * x)), 11, 61),
        smooth=pick(table(ml.OnePole.makeCoeffs, 59, lambda x: (0.1 + 0.3 * x,)), 13, 59),
        dc=pick(table(ml.DCBlocker.makeCoeffs, 53, lambda x: (0.01 + 0.09 * x,)), 17, 53),
        env=pick(table(ml.ADSR.calcCoeffs, 47, lambda x: (0.0005 + 0.0095 * x, 0.002 + 0.018 * (1.0 - x), 0.2 + 0.7 * x,
                                                          0.002 + 0.028 * x, 48000.0)), 19, 47))
    seeds = (v.astype(np.uint64) * 2654435761 % (1 << 32)).astype(np.uint32)
    if full:   # patches.synth16(full=True): the filter envelope and the per-sample cutoff
        params.update(cutoffOct=(-1.0 + 3.0 * ((v * 23) % 83) / 82.0).astype(np.float32), envAmount=(0.5 + 2.5 * ((v * 31) % 79) / 78.0).astype(np.float32),
                      cutoffBase=np.float32(400.0 / 48000.0), resonance=(0.3 + 1.2 * ((v * 43) % 73) / 72.0).astype(np.float32))
        coeffs["fenv"] = pick(table(ml.ADSR.calcCoeffs, 41, lambda x: (0.001 + 0.02 * x, 0.01 + 0.1 * (1.0 - x), 0.1 + 0.6 * x,
                                                                         0.01 + 0.2 * x, 48000.0)), 29, 41)
        del coeffs["lp"]   # the Lopass gets omega and k per sample: it has no stored coefficients
    return params, coeffs, seeds


def cfg5_gate_quad(lo, hi, n_vectors):
    """A deterministic gate signal for GLOBAL voices [lo, hi) in the QUAD layout [16 T][n][4]: voice v is on
    (amplitude 0.2..1.0) for half of a per-voice period of 600..4600 samples, phase-shifted per voice."""
    v = np.arange(lo, hi, dtype=np.int64)[None, :, None]
    s = (np.arange(16 * n_vectors, dtype=np.int64)[:, None, None] * 4 + np.arange(4, dtype=np.int64)[None, None, :])
    half = 300 + (v * 131) % 2000
    on = (((s + (v * 977) % 4096) // half) & 1) == 0
    amp = (0.2 + 0.8 * ((v * 41) % 64) / 63.0).astype(np.float32)
    return np.ascontiguousarray(np.where(on, amp, np.float32(0.0)).astype(np.float32))


def max_over_ranks(seconds, dist=None, device="cpu"):
    """Barrier-bracketed wall time of the slowest rank (what bench.py reports)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    import torch
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _yield():
    """Inside a spin loop: let another thread of this process have the interpreter (ranks may be threads: `bench.py --launcher threads`)."""
    import time
    time.sleep(0)


class RowExchange:
    """The one exchange a sharded bank has when a host wants ONE channel from it (SURVEY 8e "optional later: per-GPU reduction then host
    add"): every rank hands rank 0 the rows of the mixdown tree its shard is whole at (mlgpu_bank_process_mixdown_shard: 256 bytes per
    row and DSPVector, one row per 262 144 voices), rank 0 finishes the tree (mlgpu_mixdown_finish). No collective library: the rows
    go through a POSIX shared-memory segment of the node - ranks may be processes (torch.distributed.run, bench.py's own launcher)
    or threads -, two slots so that a rank can write block k + 1 while rank 0 still reads block k, sequence counters instead of locks.
    `rdv`: the ranks' rendezvous (gather is used once, for the segment's name)."""

    def __init__(self, rdv, rank, world, rows_per_rank, max_vectors):
        from multiprocessing import shared_memory
        self.rank, self.world, self.rows, self.S = int(rank), int(world), int(rows_per_rank), 64 * int(max_vectors)
        self.hdr = 2 * self.world + 2                                   # uint64: written[slot][rank], consumed[slot]
        nbytes = 8 * self.hdr + 4 * 2 * self.world * self.rows * self.S
        self.shm = shared_memory.SharedMemory(create=True, size=nbytes) if self.rank == 0 else None
        names = rdv.gather(self.shm.name if self.rank == 0 else None)
        if self.rank != 0:
            name = next(n for n in names if n)
            self.shm = shared_memory.SharedMemory(name=name)
            try:   # (the segment is rank 0's to unlink: keep this process's resource tracker from doing it at exit and warning about it)
                from multiprocessing import resource_tracker
                resource_tracker.unregister(self.shm._name, "shared_memory")
            except Exception:
                pass
        self.seq = np.ndarray((self.hdr,), np.uint64, self.shm.buf, 0)
        self.data = np.ndarray((2, self.world * self.rows, self.S), np.float32, self.shm.buf, 8 * self.hdr)
        if self.rank == 0:
            self.seq[:] = 0
        rdv.barrier()

    def put(self, block, rows):
        """This rank's rows [rows_per_rank][64 T] of block `block` (0, 1, 2 ...)."""
        slot, T64 = block & 1, rows.shape[1]
        while block >= 2 and int(self.seq[2 * self.world + slot]) < block - 1:      # rank 0 has not read this slot's block k - 2 yet
            _yield()
        self.data[slot, self.rank * self.rows:(self.rank + 1) * self.rows, :T64] = rows
        self.seq[slot * self.world + self.rank] = block + 1

    def collect(self, block, n_vectors):
        """Rank 0: all ranks' rows of block `block`, [world * rows_per_rank][64 n_vectors], in voice order (a copy: the slot is free again)."""
        slot = block & 1
        want = block + 1
        while any(int(self.seq[slot * self.world + r]) < want for r in range(self.world)):
            _yield()
        out = np.array(self.data[slot, :, :64 * n_vectors])
        self.seq[2 * self.world + slot] = want
        return out

    def close(self):
        self.seq = self.data = None
        try:
            self.shm.close()
            if self.rank == 0:
                self.shm.unlink()
        except Exception:
            pass
