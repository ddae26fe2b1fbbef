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


def max_over_ranks(seconds, dist=None, device="cpu"):
    """Barrier-bracketed wall time of the slowest rank (what bench.py reports)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    import torch
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
