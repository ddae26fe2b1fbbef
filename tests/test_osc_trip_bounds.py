"""The claim behind the graph kernels' oscillator trips (madronalib_amd/csrc/mldsp_procs.hpp: trip_u), checked on the CPU in numpy with
the device's own float32 arithmetic: for a per-voice frequency 0 < dt <= 1 / (2 N), a trip of N consecutive samples has at most one
sample in the zone after a step (t < dt) and at most one in the zone before it (t > 1 - dt) - for the oscillator's phase and for
PulseGen's shifted phase fract(p - w + 1) - UNLESS one of the trip's phases lies within kTripTiny / kTripNearOne (kTripTinyShifted /
kTripNearOneShifted) of 0 or 1, which is what sends a trip to the per-sample form. Free-running phases by the hundred million, and
phases placed on the knife edges (a few units of 2^-32 after / before a wrap, around the width's grid point): every trip with two
samples in one zone must be a recognised one. The thresholds are read from the header, so the test follows the code."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def thresholds():
    src = open(os.path.join(ROOT, "madronalib_amd", "csrc", "mldsp_procs.hpp")).read()
    m = re.search(r"kTripTiny = 0x1p-(\d+)f, kTripNearOne = 1\.0f - 0x1p-(\d+)f;", src)
    s = re.search(r"kTripTinyShifted = 0x1p-(\d+)f, kTripNearOneShifted = 1\.0f - 0x1p-(\d+)f;", src)
    f = re.search(r"kTripMaxFreq\(int n\) \{ return ([0-9.]+)f / \(float\)n; \}", src)
    assert m and s and f, "the trip constants of mldsp_procs.hpp changed their spelling: update this test"
    t = lambda e: np.float32(2.0 ** -int(e))
    return t(m.group(1)), np.float32(1.0) - t(m.group(2)), t(s.group(1)), np.float32(1.0) - t(s.group(2)), float(f.group(1))


def phases(om, istep, n):
    """the device's PhasorGen for n samples: omega32 += istep; p = float32(int32(omega32 >> 1)) * 2^-31"""
    k = np.arange(1, n + 1, dtype=np.uint64)
    ph = (om[:, None] + k[None, :] * istep[:, None]) & np.uint64(0xFFFFFFFF)
    return (ph >> np.uint64(1)).astype(np.float32) * np.float32(2.0 ** -31)


def shifted(p, w):
    d = (p - w[:, None]).astype(np.float32) + np.float32(1.0)
    return (d - np.floor(d)).astype(np.float32)   # v_fract_f32 on [0, 2]


def check(N, freq, width, om, samples, tiny, near_one, tiny_s, near_one_s):
    istep = np.rint(freq.astype(np.float64) * 2.0 ** 32).astype(np.uint64)   # cvtps2dq of dt * 2^32 (exact product)
    p = phases(om.astype(np.uint64), istep, samples)
    d = shifted(p, width)
    dt = freq[:, None]
    omdt = (np.float32(1.0) - freq)[:, None]
    stats = dict(trips=0, doubles=0, suspects=0, missed=0)
    for t, lo_t, hi_t in ((p, tiny, near_one), (d, tiny_s, near_one_s)):
        V = t.shape[0]
        tr = t.reshape(V, samples // N, N)
        lo = (tr < dt[:, :, None]).sum(2)
        hi = (tr > omdt[:, :, None]).sum(2)
        both = ((tr < dt[:, :, None]) & (tr > omdt[:, :, None])).any(2)
        double = (lo > 1) | (hi > 1) | both
        suspect = (tr.min(2) < lo_t) | (tr.max(2) > hi_t)
        stats["trips"] += double.size
        stats["doubles"] += int(double.sum())
        stats["suspects"] += int(suspect.sum())
        stats["missed"] += int((double & ~suspect).sum())
    return stats


@pytest.mark.parametrize("N", [4, 8, 16])
def test_two_samples_in_one_zone_only_in_recognised_trips(N):
    tiny, near_one, tiny_s, near_one_s, fmax = thresholds()
    limit = np.float32(fmax / N)
    rng = np.random.default_rng(100 + N)
    total = dict(trips=0, doubles=0, suspects=0, missed=0)
    # (1) free-running: random phases, frequencies log-uniform up to the limit (and AT the limit), random and special widths
    V, S = 1 << 16, 64 * N
    for rep in range(3):
        freq = (1e-7 * ((float(limit) / 1e-7) ** rng.random(V))).astype(np.float32)
        freq[::97] = limit
        freq = np.minimum(freq, limit)
        width = rng.uniform(0.0, 1.0, V).astype(np.float32)
        width[::11] = rng.choice(np.array([0.0, 1.0, 0.5, 0.25, 0.75], np.float32), len(width[::11]))
        width[5::13] = freq[5::13]
        width[7::13] = np.float32(1.0) - freq[7::13]
        om = rng.integers(0, 2 ** 32, V, dtype=np.uint64)
        st = check(N, freq, width, om, S, tiny, near_one, tiny_s, near_one_s)
        for k in total:
            total[k] += st[k]
    # (2) on the edges: a sample 0 .. 63 units after a wrap, 0 .. 1023 units before one, and within +-640 units of the width
    V = 1 << 15
    freq = (1e-5 * ((float(limit) / 1e-5) ** rng.random(V))).astype(np.float32)
    width = rng.uniform(0.0, 1.0, V).astype(np.float32)
    istep = np.rint(freq.astype(np.float64) * 2.0 ** 32).astype(np.uint64)
    v = np.arange(V, dtype=np.uint64)
    k = v % np.uint64(3 * N) + np.uint64(1)
    big = np.uint64(2 ** 32 * 64)
    wq = np.rint(width.astype(np.float64) * 2.0 ** 32).astype(np.uint64)
    om = np.where(v % 3 == 0, big - k * istep + (v // 3) % 64,
                  np.where(v % 3 == 1, big - k * istep - ((v // 3) % 1024), big + wq - k * istep + ((v // 3) % 1280) - 640)) & np.uint64(0xFFFFFFFF)
    st = check(N, freq, width, om, 4 * N * 4, tiny, near_one, tiny_s, near_one_s)
    for kk in total:
        total[kk] += st[kk]
    print(f"N={N}: {total}")
    assert total["missed"] == 0, total
    assert total["doubles"] > 200, total                      # the knife edges do occur in this volume ...
    assert total["suspects"] < 0.02 * total["trips"], total   # ... and the per-sample form stays the exception
