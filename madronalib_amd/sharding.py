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
