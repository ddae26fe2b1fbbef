"""Seeded inputs and comparison helpers shared by the parity tests."""
import numpy as np

from madronalib_amd.constants import Op, Proc


def lcg_noise(seeds, n):
    """The reference NoiseGen stream (MLDSPGens.h:115,127-128) for each seed: [len(seeds)][n] in [-1,1)."""
    seed = np.asarray(seeds, np.uint32).copy()
    out = np.empty((seed.size, n), np.float32)
    for i in range(n):
        seed = (seed * np.uint32(0x0019660D) + np.uint32(0x3C6EF35F)).astype(np.uint32)
        bits = ((seed >> np.uint32(9)) & np.uint32(0x007FFFFF)) | np.uint32(0x3F800000)
        out[:, i] = bits.view(np.float32) * np.float32(2.0) - np.float32(3.0)
    return out


def ramp_pi(V, S=64):
    """SURVEY §8d config-2 ramp: x = -pi + 2pi*((v*S+n) mod 4096)/4095."""
    idx = (np.arange(V * S, dtype=np.int64) % 4096).astype(np.float32)
    return (np.float32(-np.pi) + np.float32(2 * np.pi) * idx / np.float32(4095.0)).astype(np.float32).reshape(V, S)


SPECIALS = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 2.0, 1e-38, -1e-38, 1e-45, 3.4e38, -3.4e38,
                     np.inf, -np.inf, np.nan, 88.5, -88.5, 100.0, -100.0, 8192.0, 1e9, -1e9, 3e9, -3e9,
                     2147483520.0, 2147483648.0, -2147483648.0, 0.70710678, 1.5, 2.5, -1.5, -2.5,
                     1.1754944e-38, 6.2831855, 3.1415927, -3.1415927, 0.78539816, 1e-20, 1e-19, 127.99, -126.5],
                    dtype=np.float32)


def general_floats(n, seed=0):
    """Mixed-magnitude floats incl. specials; length n (multiple of 64)."""
    rng = np.random.default_rng(seed)
    a = rng.standard_normal(n).astype(np.float32)
    scale = np.float32(10.0) ** rng.integers(-6, 7, n).astype(np.float32)
    a = (a * scale).astype(np.float32)
    k = min(len(SPECIALS), n)
    a[:k] = SPECIALS[:k]
    # a few raw bit patterns (denormals etc.)
    raw = rng.integers(0, 2**32, max(1, n // 16), dtype=np.uint64).astype(np.uint32).view(np.float32)
    a[k:k + raw.size] = raw[: max(0, min(raw.size, n - k))]
    return a


def op_inputs(op, n=64 * 64, seed=0):
    """(a, b, c) operands appropriate for `op` (None where unused)."""
    rng = np.random.default_rng(seed + int(op))
    a = general_floats(n, seed + 1)
    b = general_floats(n, seed + 2)[::-1].copy()
    c = general_floats(n, seed + 3)
    if op in (Op.SIN_APPROX, Op.COS_APPROX, Op.EXP_APPROX_OF_SIN_APPROX):
        a[n // 2:] = rng.uniform(-np.pi, np.pi, n - n // 2).astype(np.float32)
    if op in (Op.LOG, Op.LOG2, Op.LOG_APPROX, Op.LOG2_APPROX, Op.SQRT, Op.SQRT_APPROX, Op.POW, Op.POW_APPROX):
        a[n // 2:] = np.abs(a[n // 2:]) + np.float32(1e-30)
    if op in (Op.POW, Op.POW_APPROX):
        b[n // 2:] = rng.uniform(-4, 4, n - n // 2).astype(np.float32)
    if op in (Op.PHASOR_TO_SINE, Op.PHASOR_TO_SAW, Op.PHASOR_TO_PULSE):
        # second half: what an oscillator feeds them - a phasor on [0, 1), frequencies from tiny to Nyquist (and a few absurd
        # ones: 0, negative, beyond 2^64, denormal), pulse widths over the whole range incl. the two edges
        h = n - n // 2
        a[n // 2:] = rng.random(h).astype(np.float32)
        a[n // 2:n // 2 + 8] = np.array([0.0, 1.0 - 2.0 ** -24, 0.5, 1e-9, 2.0 ** -31, 0.999, 1e-3, 0.25], np.float32)
        b[n // 2:] = (10.0 ** rng.uniform(-6, -0.31, h)).astype(np.float32)
        b[n // 2 + 8:n // 2 + 16] = np.array([0.0, -0.01, 1e20, 3e-42, 0.5, 0.499, 1e-30, 2.0 ** -64], np.float32)
        c[n // 2:] = rng.random(h).astype(np.float32)
        c[n // 2 + 16:n // 2 + 24] = np.array([0.0, 1.0, 1e-4, 0.9999, 0.5, 0.5, 2.0, -1.0], np.float32)
    if op in Op.INT_INPUT or op == Op.SELECT_INT:
        a = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
        b = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    if op in (Op.SELECT, Op.SELECT_INT):
        c = np.where(rng.random(n) < 0.5, np.uint32(0xFFFFFFFF), np.uint32(0)).astype(np.uint32)
        c[:8] = rng.integers(0, 2**32, 8, dtype=np.uint64).astype(np.uint32)  # partial masks are bitwise
    if op in Op.UNARY:
        return a, None, None
    if op in Op.BINARY:
        return a, b, None
    return a, b, c


def is_float_result(op):
    return op not in (Op.ROUND_FLOAT_TO_INT, Op.TRUNCATE_FLOAT_TO_INT, Op.ADD_INT32, Op.SUBTRACT_INT32,
                      Op.EQUAL, Op.NOT_EQUAL, Op.GREATER_THAN, Op.GREATER_THAN_OR_EQUAL, Op.LESS_THAN,
                      Op.LESS_THAN_OR_EQUAL, Op.WITHIN, Op.SELECT_INT, Op.SELECT)


def assert_bits_equal(got, want, float_result=True, what=""):
    """Bit-exact comparison; for float results any NaN matches any NaN (payloads are not portable)."""
    g = np.ascontiguousarray(got).view(np.uint32).ravel()
    w = np.ascontiguousarray(want).view(np.uint32).ravel()
    assert g.shape == w.shape, (g.shape, w.shape)
    neq = g != w
    if float_result:
        gn = np.isnan(g.view(np.float32))
        wn = np.isnan(w.view(np.float32))
        neq &= ~(gn & wn)
    if neq.any():
        i = int(np.flatnonzero(neq)[0])
        raise AssertionError(f"{what}: {int(neq.sum())}/{g.size} words differ; first at {i}: "
                             f"got 0x{g[i]:08x} ({g.view(np.float32)[i]!r}) want 0x{w[i]:08x} ({w.view(np.float32)[i]!r})")


def assert_rel_close(got, want, rel, what=""):
    """|got-want| <= rel*|want| elementwise; NaN/inf must agree."""
    g = np.ascontiguousarray(got).view(np.float32).ravel().astype(np.float64)
    w = np.ascontiguousarray(want).view(np.float32).ravel().astype(np.float64)
    fin = np.isfinite(w) & np.isfinite(g)
    assert (np.isnan(g) == np.isnan(w)).all(), what + ": NaN pattern differs"
    inf = np.isinf(w)
    assert (g[inf] == w[inf]).all(), what + ": inf pattern differs"
    err = np.abs(g[fin] - w[fin])
    tol = rel * np.abs(w[fin]) + 1e-44
    bad = err > tol
    worst = float((err / (np.abs(w[fin]) + 1e-300)).max()) if err.size else 0.0
    key = " ".join(what.split(" ")[:2])   # the label's first two words: one running maximum per kind of comparison
    REL_MEASURED[key] = max(worst, REL_MEASURED.get(key, 0.0))
    assert not bad.any(), f"{what}: {int(bad.sum())} beyond rel {rel}; worst {worst:.3e}"


# the largest relative difference every toleranced comparison of this session has seen, by label (printed by conftest.py at the end:
# the tolerance is a bound, this is the measurement)
REL_MEASURED = {}


# ---- chain test cases -------------------------------------------------------------------

def proc_default_coeffs(mk, kind, V, seed=0):
    """Per-voice coefficient arrays [NC][V] for one processor, from the checker's own makeCoeffs."""
    rng = np.random.default_rng(seed + kind)
    P = Proc
    if kind in (P.LOPASS, P.BANDPASS):
        name = "lopass" if kind == P.LOPASS else "bandpass"
        return np.stack([mk.make_coeffs(name, rng.uniform(0.001, 0.45), rng.uniform(0.05, 2.0)) for _ in range(V)], 1)
    if kind == P.HIPASS:
        return np.stack([mk.make_coeffs("hipass", rng.uniform(0.001, 0.45), rng.uniform(0.05, 2.0)) for _ in range(V)], 1)
    if kind == P.LO_SHELF:
        return np.stack([mk.make_coeffs("loshelf", rng.uniform(0.001, 0.4), rng.uniform(0.3, 2.0), rng.uniform(0.25, 4.0)) for _ in range(V)], 1)
    if kind == P.HI_SHELF:
        return np.stack([mk.make_coeffs("hishelf", rng.uniform(0.001, 0.4), rng.uniform(0.3, 2.0), rng.uniform(0.25, 4.0)) for _ in range(V)], 1)
    if kind == P.BELL:
        return np.stack([mk.make_coeffs("bell", rng.uniform(0.001, 0.4), rng.uniform(0.3, 2.0), rng.uniform(0.25, 4.0)) for _ in range(V)], 1)
    if kind in (P.ONE_POLE, P.RMS):
        return np.stack([mk.make_coeffs("onepole", rng.uniform(0.0005, 0.3)) for _ in range(V)], 1)
    if kind == P.PEAK:
        c = np.stack([mk.make_coeffs("onepole", rng.uniform(0.0005, 0.3)) for _ in range(V)], 1)
        hold = rng.integers(0, 400, V).astype(np.uint32).view(np.float32)[None, :]
        return np.concatenate([c, hold], 0)
    if kind == P.DC_BLOCKER:
        return np.array([[mk.dcblocker_coeffs(rng.uniform(0.01, 0.2)) for _ in range(V)]], np.float32)
    if kind == P.INTEGRATOR:
        return rng.uniform(0.0, 0.01, (1, V)).astype(np.float32)
    if kind == P.ADSR:
        return np.stack([mk.make_coeffs("adsr", rng.uniform(0.0001, 0.01), rng.uniform(0.001, 0.01),
                                        rng.uniform(0.1, 0.9), rng.uniform(0.001, 0.01), 48000.0) for _ in range(V)], 1)
    if kind == P.GAIN:
        return rng.uniform(-1.5, 1.5, (1, V)).astype(np.float32)
    if kind == P.PULSE_GEN:
        return rng.uniform(0.05, 0.95, (1, V)).astype(np.float32)
    if kind == P.ALLPASS1:
        return np.array([[mk.allpass1_coeffs(rng.uniform(0.618, 1.618)) for _ in range(V)]], np.float32)
    if kind == P.SAMPLE_ACCURATE_LINEAR_GLIDE:
        return np.stack([mk.make_coeffs("sample_glide", rng.uniform(0.0, 300.0)) for _ in range(V)], 1)
    if kind == P.LINEAR_GLIDE:
        return np.stack([mk.make_coeffs("linear_glide", rng.uniform(0.0, 700.0)) for _ in range(V)], 1)
    return np.zeros((0, V), np.float32)


def chain_coeffs(mk, procs, V, seed=0):
    parts = [proc_default_coeffs(mk, int(p), V, seed + 17 * i) for i, p in enumerate(procs)]
    return np.ascontiguousarray(np.concatenate(parts, 0).astype(np.float32)) if parts else np.zeros((0, V), np.float32)


def gate_signal(V, S, seed=0):
    """ADSR input: per-voice gate on/off pattern with random amplitudes."""
    rng = np.random.default_rng(seed)
    x = np.zeros((V, S), np.float32)
    for v in range(V):
        t = 0
        on = False
        while t < S:
            L = int(rng.integers(20, 400))
            if on:
                x[v, t:t + L] = np.float32(rng.uniform(0.2, 1.0))
            on = not on
            t += L
    return x


def chain_input(procs, V, T, seed=0):
    """(in_signal or None, in_const or None) suitable for the head processor of the chain."""
    rng = np.random.default_rng(seed)
    head = int(procs[0])
    S = 64 * T
    if head == Proc.NOISE_GEN:
        return None, None
    if head in Proc.GENERATORS:
        # cycles per sample: log-spaced 20 Hz..8 kHz at 48 kHz, < 0.5
        f = (20.0 * (400.0 ** rng.random(V)) / 48000.0).astype(np.float32)
        return None, f
    if head == Proc.ADSR:
        return gate_signal(V, S, seed), None
    if head == Proc.SAMPLE_ACCURATE_LINEAR_GLIDE:
        return stepped(V, S, seed, -1.0, 1.0, 30, 500), None
    return lcg_noise(np.arange(V, dtype=np.uint32) + np.uint32(seed * 1000), S), None


def stepped(V, S, seed, lo, hi, min_len, max_len):
    """Per-voice piecewise-constant signal [V][S]: a new uniform value in [lo, hi) every min_len..max_len samples
    (sometimes repeating the previous value, which must NOT restart a glide)."""
    rng = np.random.default_rng(seed)
    x = np.zeros((V, S), np.float32)
    for v in range(V):
        t, val = 0, np.float32(0.0)
        while t < S:
            L = int(rng.integers(min_len, max_len + 1))
            if rng.random() > 0.2:
                val = np.float32(rng.uniform(lo, hi))
            x[v, t:t + L] = val
            t += L
    return x


MULTI_CASES = ("pulse2", "lopass_mod", "loshelf_vc", "hishelf_vc", "interp1", "linear_glide", "linear_glide_long", "tempo_lock")


def multi_case(mk, name, V, T, seed=0):
    """One processor in one of its multi-input / vector-rate forms.
    Returns dict(kind, coeffs [NC][V], inputs [(rate, array)], ...): rate 'audio' -> [V][64T], 'control' -> [V][T]."""
    rng = np.random.default_rng(seed + 101)
    S = 64 * T
    P = Proc

    def interp(c0, c1):  # [V][T] endpoints per vector -> audio-rate interpolateDSPVectorLinear rows
        from madronalib_amd.constants import Vop
        return mk.vop(Vop.INTERPOLATE_LINEAR, V, T, np.repeat(c0, 64, 1), np.repeat(c1, 64, 1))

    if name == "pulse2":
        f = (20.0 * (400.0 ** rng.random(V)) / 48000.0).astype(np.float32)
        freq = (f[:, None] * (1.0 + 0.3 * np.sin(np.arange(S)[None, :] * 0.01 * (1 + np.arange(V)[:, None] % 5)))).astype(np.float32)
        width = (0.5 + 0.4 * lcg_noise(np.arange(V, dtype=np.uint32) + 7, S)).astype(np.float32)
        return dict(kind=P.PULSE_GEN, coeffs=np.full((1, V), 0.5, np.float32), inputs=[("audio", freq), ("audio", width)])
    if name == "lopass_mod":
        x = lcg_noise(np.arange(V, dtype=np.uint32) + 31, S)
        # omega sweeps through the <= 0.5 clamp (never negative: that filter is unstable); k through the >= 0.01 clamp
        omega = (0.31 + 0.29 * np.sin(np.arange(S)[None, :] * 0.003 * (1 + np.arange(V)[:, None] % 7))).astype(np.float32)
        k = (0.8 + 0.9 * np.sin(np.arange(S)[None, :] * 0.0017 * (2 + np.arange(V)[:, None] % 3))).astype(np.float32)
        return dict(kind=P.LOPASS, coeffs=np.zeros((3, V), np.float32), inputs=[("audio", x), ("audio", omega), ("audio", k)])
    if name in ("loshelf_vc", "hishelf_vc"):
        kind = P.LO_SHELF if name == "loshelf_vc" else P.HI_SHELF
        mkname = "loshelf" if kind == P.LO_SHELF else "hishelf"
        nc = 5 if kind == P.LO_SHELF else 6
        # vcoeffs(p0, p1) = interpolateCoeffsLinear(makeCoeffs(p0), makeCoeffs(p1)) per vector (MLDSPFilters.h:283-286)
        ends = np.zeros((nc, V, T + 1), np.float32)
        for v in range(V):
            for t in range(T + 1):
                ends[:, v, t] = mk.make_coeffs(mkname, rng.uniform(0.01, 0.3), rng.uniform(0.4, 1.5), rng.uniform(0.5, 2.5))
        x = lcg_noise(np.arange(V, dtype=np.uint32) + 57, S)
        rows = [interp(np.ascontiguousarray(ends[c, :, :-1]), np.ascontiguousarray(ends[c, :, 1:])) for c in range(nc)]
        return dict(kind=kind, coeffs=np.zeros((nc, V), np.float32), inputs=[("audio", x)] + [("audio", r) for r in rows])
    if name == "interp1":
        c = stepped(V, T, seed + 3, -2.0, 2.0, 1, 4)
        return dict(kind=P.INTERPOLATOR1, coeffs=np.zeros((0, V), np.float32), inputs=[("control", c)])
    if name in ("linear_glide", "linear_glide_long"):
        c = stepped(V, T, seed + 5, -1.0, 1.0, 1, 12) if name == "linear_glide" else stepped(V, T, seed + 6, 0.0, 8.0, 8, 40)
        co = proc_default_coeffs(mk, P.LINEAR_GLIDE, V, seed) if name == "linear_glide" else \
            np.stack([mk.make_coeffs("linear_glide", 64.0 * (3 + v % 30)) for v in range(V)], 1)
        return dict(kind=P.LINEAR_GLIDE, coeffs=co, inputs=[("control", c)])
    if name == "tempo_lock":
        # the input clock: stopped (-1) for a few vectors, then a phasor of per-voice rate; ratios that lock (2, 1/2, 3),
        # one that does not (1.37), changing now and then
        x = np.full((V, S), -1.0, np.float32)
        for v in range(V):
            start = 64 * int(rng.integers(0, 4))
            rate = np.float32(rng.uniform(0.0002, 0.002))
            ph = np.float32(rng.random())
            for i in range(start, S):
                x[v, i] = ph
                ph = np.float32(ph + rate)
                if ph > 1.0:
                    ph = np.float32(ph - 1.0)
            if v % 5 == 4:
                x[v, 64 * (T // 2):64 * (T // 2 + 2)] = -1.0   # the clock stops and restarts
        ratios = np.array([2.0, 0.5, 3.0, 1.37, 1.0, 0.25], np.float32)
        dydx = ratios[(np.arange(V)[:, None] + np.arange(T)[None, :] // 7) % len(ratios)].astype(np.float32)
        isr = np.full((V, T), np.float32(1.0 / 48000.0), np.float32)
        return dict(kind=P.TEMPO_LOCK, coeffs=np.zeros((0, V), np.float32), inputs=[("audio", x), ("control", dydx), ("control", isr)])
    raise KeyError(name)


def multi_inputs_audio(case, T):
    """The case's inputs as audio-rate arrays [V][64T] (controls repeated 64 times per vector), for the CPU checkers."""
    return [np.ascontiguousarray(a if r == "audio" else np.repeat(a, 64, 1), np.float32) for r, a in case["inputs"]]


DELAY_CASES = ("integer_const", "integer_var", "frac_const", "frac_var", "frac_ticks", "pitchbend")


def delay_case(mk, name, V, T, seed=0):
    """A delay-line processor case: dict(kind, max_delay, state0 [NS][V] uint32, inputs [arrays [V][64T]])."""
    rng = np.random.default_rng(seed + 303)
    S = 64 * T
    P = Proc
    x = lcg_noise(np.arange(V, dtype=np.uint32) + 900 + seed, S)
    max_delay = 192.0   # ring of 256 samples: valid delays 0 .. 192
    slow = (100.0 + 80.0 * np.sin(np.arange(S)[None, :] * 0.004 * (1 + np.arange(V)[:, None] % 5))).astype(np.float32)
    if name == "integer_const":
        st = np.zeros((2, V), np.uint32)
        st[1] = rng.integers(0, 193, V).astype(np.uint32)      # incl. 0 (reads the sample just written) and the maximum
        st[1, :3] = [0, 1, 192]
        st[0] = rng.integers(0, 256, V).astype(np.uint32)      # arbitrary write positions: wrap-around inside a vector
        return dict(kind=P.INTEGER_DELAY, max_delay=max_delay, state0=st, inputs=[x])
    if name == "integer_var":
        st = np.zeros((2, V), np.uint32)
        d = stepped(V, S, seed + 1, 0.0, 192.9, 5, 300)
        return dict(kind=P.INTEGER_DELAY, max_delay=max_delay, state0=st, inputs=[x, d])
    if name == "frac_const":
        st = np.zeros((5, V), np.uint32)
        for v in range(V):
            st[3:5, v] = mk.fractional_delay_state(rng.uniform(0.0, 190.0) if v > 3 else [0.0, 0.3, 1.0, 64.617][v]).view(np.uint32)
        return dict(kind=P.FRACTIONAL_DELAY, max_delay=max_delay, state0=st, inputs=[x])
    if name == "frac_var":
        return dict(kind=P.FRACTIONAL_DELAY, max_delay=max_delay, state0=np.zeros((5, V), np.uint32), inputs=[x, slow])
    if name == "frac_ticks":
        ticks = np.where(rng.random((V, S)) < 0.05, np.uint32(0xFFFFFFFF), np.uint32(0)).astype(np.uint32).view(np.float32)
        return dict(kind=P.FRACTIONAL_DELAY, max_delay=max_delay, state0=np.zeros((5, V), np.uint32), inputs=[x, slow, ticks])
    if name == "pitchbend":
        return dict(kind=P.PITCHBENDABLE_DELAY, max_delay=max_delay, state0=np.zeros((10, V), np.uint32), inputs=[x, slow])
    raise KeyError(name)


def region_case(V, T, seed=0):
    """Inputs of the rate-region cases (Upsample2xFunction / Downsample2xFunction around one stateful fn): noise x, a slow
    modulator m, per-voice oscillator frequency, Lopass coefficients (omega 0.2, k 0.8 from the caller's makeCoeffs)."""
    x = lcg_noise(np.arange(V, dtype=np.uint32) + np.uint32(3 + seed), 64 * T)
    m = (0.5 + 0.5 * np.sin(np.arange(64 * T) * 0.01)[None, :] * np.linspace(0.2, 1.0, V)[:, None]).astype(np.float32)
    freq = (55.0 * 2.0 ** (6.0 * np.arange(V) / max(1, V)) / 48000.0).astype(np.float32)
    return x, m, freq


def hostile_gate(V, S, seed):
    """ADSR gates the segment logic was not written for: negative and -0 levels, NaN, +-inf, denormals, levels that change
    without passing through zero, single-sample blips - between ordinary on/off stretches."""
    rng = np.random.default_rng(seed)
    odd = np.array([-0.0, -0.5, np.nan, np.inf, -np.inf, 1e-40, -1e-40, 3.0e38, 1.0e-38], np.float32)
    x = np.zeros((V, S), np.float32)
    for v in range(V):
        t = 0
        while t < S:
            L = int(rng.integers(1, 300))
            r = rng.random()
            if r < 0.35:
                val = np.float32(0.0)
            elif r < 0.75:
                val = np.float32(rng.uniform(0.05, 1.5))
            else:
                val = odd[rng.integers(0, len(odd))]
            x[v, t:t + L] = val
            t += L
    return x


def hostile_adsr_state(clean, rng):
    """ADSR state [8][V] the envelope itself never produces: an off segment with a nonzero y, k or target, segments past off, NaN
    thresholds (words: y, y1, x1, threshold, target, k, amp, segment); every fifth voice keeps the clear() state."""
    st = clean.copy()
    V = st.shape[1]
    vals = np.array([0.0, -0.0, 0.3, -0.3, 1.0, 1.1, 1e-40, np.nan, np.inf, 2.0], np.float32)
    for w in range(7):
        st[w] = vals[rng.integers(0, len(vals), V)].view(np.uint32)
    st[7] = rng.integers(0, 7, V).astype(np.uint32)
    st[:, ::5] = clean[:, ::5]
    return st
