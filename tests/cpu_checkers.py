"""ctypes front-ends for the two CPU checkers (TEST INFRASTRUCTURE ONLY).

  Oracle()  -> oracle/libmloracle.so   plain-C restatement (oracle/ml_oracle.c), prefix mlorc_
  Ref()     -> oracle/_ref/libmlref.so the compiled reference (oracle/ref_wrapper.cpp), prefix mlref_

Both expose the same calls so a test can run either one. Signals are VOICE_MAJOR
[V][64*T] float32; coeffs/state are SoA [slot][V] exactly like the GPU bank.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

c_f32p = ctypes.POINTER(ctypes.c_float)
c_u32p = ctypes.POINTER(ctypes.c_uint32)
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_f64p = ctypes.POINTER(ctypes.c_double)


def _ptr(a, t):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(t)


def build_oracle():
    """Compile oracle/libmloracle.so (gcc only; works on the GPU box too)."""
    so = os.path.join(ORACLE_DIR, "libmloracle.so")
    src = os.path.join(ORACLE_DIR, "ml_oracle.c")
    hdr = os.path.join(ROOT, "include", "mlgpu.h")
    if (not os.path.exists(so)) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libmloracle.so"], stdout=subprocess.DEVNULL)
    return so


def build_ref():
    """Compile oracle/_ref/libmlref.so when /root/reference is present; else use a prebuilt one."""
    so = os.path.join(ORACLE_DIR, "_ref", "libmlref.so")
    if os.path.isdir("/root/reference/source/DSP"):
        src = os.path.join(ORACLE_DIR, "ref_wrapper.cpp")
        if (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "_ref/libmlref.so"], stdout=subprocess.DEVNULL)
    return so if os.path.exists(so) else None


class _Checker:
    prefix = None

    def __init__(self, path):
        self.lib = ctypes.CDLL(path)
        self.path = path
        L, p = self.lib, self.prefix
        sz = ctypes.c_size_t
        f = ctypes.c_float

        def fn(name, res, args):
            h = getattr(L, p + name)
            h.restype = res
            h.argtypes = args
            return h

        self._op_apply = fn("op_apply", ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, sz])
        self._op_rows1 = fn("op_apply_rows1", ctypes.c_int, [ctypes.c_int, c_f32p, c_f32p, c_f32p, sz])
        self._row_reduce = fn("row_reduce", ctypes.c_int, [ctypes.c_int, c_f32p, c_f32p, sz])
        self._nc = fn("proc_num_coeffs", ctypes.c_int, [ctypes.c_int])
        self._ns = fn("proc_num_state", ctypes.c_int, [ctypes.c_int])
        self._clear = fn("chain_clear", ctypes.c_int, [c_i32p, ctypes.c_int, sz, c_u32p])
        self._default = fn("chain_default_state", ctypes.c_int, [c_i32p, ctypes.c_int, sz, c_u32p])
        self._process = fn("chain_process", ctypes.c_int, [c_i32p, ctypes.c_int, sz, sz, c_f32p, c_u32p, c_f32p, c_f32p, c_f32p, ctypes.c_int])
        self._mk = {
            "lopass": (fn("lopass_make_coeffs", None, [f, f, c_f32p]), 2, 3),
            "hipass": (fn("hipass_make_coeffs", None, [f, f, c_f32p]), 2, 4),
            "bandpass": (fn("bandpass_make_coeffs", None, [f, f, c_f32p]), 2, 3),
            "loshelf": (fn("loshelf_make_coeffs", None, [f, f, f, c_f32p]), 3, 5),
            "hishelf": (fn("hishelf_make_coeffs", None, [f, f, f, c_f32p]), 3, 6),
            "bell": (fn("bell_make_coeffs", None, [f, f, f, c_f32p]), 3, 4),
            "onepole": (fn("onepole_make_coeffs", None, [f, c_f32p]), 1, 2),
            "adsr": (fn("adsr_calc_coeffs", None, [f, f, f, f, f, c_f32p]), 5, 4),
        }
        self._dcb = fn("dcblocker_make_coeffs", f, [f])
        self._db2g = fn("db_to_gain", f, [f])
        self._imp = fn("impulse_table", None, [c_f32p])
        self._multi = fn("proc_process_multi", ctypes.c_int, [ctypes.c_int, sz, sz, c_f32p, c_u32p, ctypes.POINTER(c_f32p), ctypes.c_int, c_f32p])
        self._vop = fn("vop", ctypes.c_int, [ctypes.c_int, sz, sz, c_f32p, c_f32p, c_f32p])
        self._mk["linear_glide"] = (fn("linear_glide_make_coeffs", None, [f, c_f32p]), 1, 2)
        self._mk["sample_glide"] = (fn("sample_accurate_linear_glide_make_coeffs", None, [f, c_f32p]), 1, 2)
        self._set_ftz = fn("set_flush_denormals", ctypes.c_int, [ctypes.c_int])
        self._rc = fn("range_closed", None, [f, f, c_f32p])
        self._ro = fn("range_open", None, [f, f, c_f32p])

    # ---- floating-point mode ----
    def flush_denormals(self, on=True):
        """Context manager: run the calls inside under ml::UsingFlushDenormalsToZero (MXCSR DAZ | FZ on this thread,
        MLDSPUtils.h:51-96). Keep the body to checker calls: numpy arithmetic on this thread sees the mode too."""
        checker = self

        class _Scope:
            def __enter__(self_inner):
                self_inner.prev = checker._set_ftz(1 if on else 0)
                return checker

            def __exit__(self_inner, *exc):
                checker._set_ftz(self_inner.prev)
                return False
        return _Scope()

    # ---- elementwise ----
    def op(self, op, a, b=None, c=None):
        a = np.ascontiguousarray(a)
        out = np.empty(a.shape, np.uint32)
        args = [None if x is None else np.ascontiguousarray(x) for x in (a, b, c)]
        r = self._op_apply(int(op), *[None if x is None else x.ctypes.data_as(ctypes.c_void_p) for x in args],
                           out.ctypes.data_as(ctypes.c_void_p), a.size)
        assert r == 0, r
        return out

    def op_rows1(self, op, a, b64):
        a = np.ascontiguousarray(a, np.float32)
        b64 = np.ascontiguousarray(b64, np.float32)
        out = np.empty_like(a)
        r = self._op_rows1(int(op), _ptr(a, c_f32p), _ptr(b64, c_f32p), _ptr(out, c_f32p), a.size // 64)
        assert r == 0
        return out

    def row_reduce(self, rowop, rows):
        rows = np.ascontiguousarray(rows, np.float32)
        out = np.empty(rows.size // 64, np.float32)
        r = self._row_reduce(int(rowop), _ptr(rows, c_f32p), _ptr(out, c_f32p), rows.size // 64)
        assert r == 0
        return out

    # ---- chains ----
    def num_coeffs(self, kind):
        return self._nc(int(kind))

    def num_state(self, kind):
        return self._ns(int(kind))

    def chain_sizes(self, procs):
        return sum(self.num_coeffs(p) for p in procs), sum(self.num_state(p) for p in procs)

    def chain_clear(self, procs, V):
        procs = np.asarray(procs, np.int32)
        _, ns = self.chain_sizes(procs)
        st = np.zeros((ns, V), np.uint32)
        assert self._clear(_ptr(procs, c_i32p), len(procs), V, _ptr(st, c_u32p)) == 0
        return st

    def chain_default_state(self, procs, V):
        procs = np.asarray(procs, np.int32)
        _, ns = self.chain_sizes(procs)
        st = np.zeros((ns, V), np.uint32)
        assert self._default(_ptr(procs, c_i32p), len(procs), V, _ptr(st, c_u32p)) == 0
        return st

    def chain_process(self, procs, T, coeffs, state, in_signal=None, in_const=None, n_threads=1, want_out=True):
        """state is updated in place. Returns out [V][64T] (float32)."""
        procs = np.ascontiguousarray(procs, np.int32)
        V = state.shape[1] if state.ndim == 2 and state.shape[0] > 0 else (
            coeffs.shape[1] if coeffs is not None and coeffs.ndim == 2 and coeffs.shape[0] > 0 else
            (in_const.shape[0] if in_const is not None else in_signal.shape[0]))
        coeffs = np.ascontiguousarray(coeffs, np.float32) if coeffs is not None else np.zeros((0, V), np.float32)
        assert state.dtype == np.uint32 and state.flags["C_CONTIGUOUS"]
        if in_signal is not None:
            in_signal = np.ascontiguousarray(in_signal, np.float32)
            assert in_signal.shape == (V, 64 * T)
        if in_const is not None:
            in_const = np.ascontiguousarray(in_const, np.float32)
        out = np.empty((V, 64 * T), np.float32) if want_out else None
        r = self._process(_ptr(procs, c_i32p), len(procs), V, T, _ptr(coeffs, c_f32p), _ptr(state, c_u32p),
                          _ptr(in_signal, c_f32p), _ptr(in_const, c_f32p), _ptr(out, c_f32p), n_threads)
        assert r == 0, r
        return out

    def proc_multi(self, kind, T, coeffs, state, inputs):
        """One processor in one of its multi-input forms (or a vector-rate ramp). inputs: list of [V][64T] float32
        (a control-rate value repeated 64 times per vector). state [NS][V] is updated in place."""
        V = state.shape[1]
        coeffs = np.ascontiguousarray(coeffs, np.float32) if coeffs is not None else np.zeros((0, V), np.float32)
        ins = [np.ascontiguousarray(x, np.float32) for x in inputs]
        for x in ins:
            assert x.shape == (V, 64 * T), x.shape
        arr = (c_f32p * max(1, len(ins)))(*[_ptr(x, c_f32p) for x in ins])
        out = np.empty((V, 64 * T), np.float32)
        r = self._multi(int(kind), V, T, _ptr(coeffs, c_f32p), _ptr(state, c_u32p), arr, len(ins), _ptr(out, c_f32p))
        assert r == 0, r
        return out

    def delay_process(self, kind, T, state, mem, inputs):
        """One delay-line processor. state [NS][V] and mem [V][rings][len] are updated in place; inputs [V][64T]."""
        fnc = getattr(self.lib, self.prefix + "delay_process")
        fnc.restype = ctypes.c_int
        sz = ctypes.c_size_t
        fnc.argtypes = [ctypes.c_int, sz, sz, c_u32p, c_f32p, sz, ctypes.POINTER(c_f32p), ctypes.c_int, c_f32p]
        V = state.shape[1]
        assert mem.dtype == np.float32 and mem.flags["C_CONTIGUOUS"] and mem.shape[0] == V
        ins = [np.ascontiguousarray(x, np.float32) for x in inputs]
        arr = (c_f32p * len(ins))(*[_ptr(x, c_f32p) for x in ins])
        out = np.empty((V, 64 * T), np.float32)
        r = fnc(int(kind), V, T, _ptr(state, c_u32p), _ptr(mem, c_f32p), mem.shape[-1], arr, len(ins), _ptr(out, c_f32p))
        assert r == 0, r
        return out

    def allpass1_coeffs(self, d):
        fnc = getattr(self.lib, self.prefix + "allpass1_make_coeffs")
        fnc.restype, fnc.argtypes = ctypes.c_float, [ctypes.c_float]
        return np.float32(fnc(float(d)))

    def fractional_delay_state(self, d):
        fnc = getattr(self.lib, self.prefix + "fractional_delay_make_state")
        fnc.restype, fnc.argtypes = None, [ctypes.c_float, c_f32p]
        o = np.zeros(2, np.float32)
        fnc(float(d), _ptr(o, c_f32p))
        return o

    def resample(self, octaves, up, state, x):
        """Downsampler (up=False) / Upsampler (up=True) cascade. state [octaves*9][V] updated in place; x [V][64*Tin]."""
        fnc = getattr(self.lib, self.prefix + "resample")
        fnc.restype = ctypes.c_int
        fnc.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, c_f32p, c_f32p, c_f32p]
        x = np.ascontiguousarray(x, np.float32)
        V, Sin = x.shape
        R = 1 << octaves
        out = np.empty((V, Sin * R if up else Sin // R), np.float32)
        assert state.dtype == np.float32 and state.flags["C_CONTIGUOUS"]
        assert fnc(octaves, 1 if up else 0, V, Sin // 64, _ptr(state, c_f32p), _ptr(x, c_f32p), _ptr(out, c_f32p)) == 0
        return out

    def rate_function_run(self, up, freq, lopass_coeffs, x, m):
        """Upsample2xFunction<2> / Downsample2xFunction<2> around fn = Lopass((clamp(x*3,-1,1) + SawGen(freq)) * m), * 0.5.
        freq [V], x / m [V][64*T]; every voice starts from default-constructed objects."""
        fnc = getattr(self.lib, self.prefix + "rate_function_run")
        fnc.restype = ctypes.c_int
        fnc.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p]
        x = np.ascontiguousarray(x, np.float32)
        m = np.ascontiguousarray(m, np.float32)
        fr = np.ascontiguousarray(freq, np.float32)
        co = np.ascontiguousarray(lopass_coeffs, np.float32)
        out = np.empty_like(x)
        assert fnc(1 if up else 0, x.shape[0], x.shape[1] // 64, _ptr(fr, c_f32p), _ptr(co, c_f32p), _ptr(x, c_f32p), _ptr(m, c_f32p), _ptr(out, c_f32p)) == 0
        return out

    def vop(self, vop, V, T, a=None, b=None):
        out = np.empty((V, 64 * T), np.float32)
        a = None if a is None else np.ascontiguousarray(a, np.float32)
        b = None if b is None else np.ascontiguousarray(b, np.float32)
        assert self._vop(int(vop), V, T, _ptr(a, c_f32p), _ptr(b, c_f32p), _ptr(out, c_f32p)) == 0
        return out

    # ---- coefficient makers ----
    def make_coeffs(self, name, *params):
        fnc, nin, nout = self._mk[name]
        assert len(params) == nin
        o = np.zeros(nout, np.float32)
        fnc(*[float(x) for x in params], _ptr(o, c_f32p))
        return o

    def dcblocker_coeffs(self, omega):
        return np.float32(self._dcb(float(omega)))

    def db_to_gain(self, dB):
        return np.float32(self._db2g(float(dB)))

    def impulse_table(self):
        o = np.zeros(17, np.float32)
        self._imp(_ptr(o, c_f32p))
        return o

    def range_closed(self, a, b):
        o = np.zeros(64, np.float32)
        self._rc(float(a), float(b), _ptr(o, c_f32p))
        return o

    def range_open(self, a, b):
        o = np.zeros(64, np.float32)
        self._ro(float(a), float(b), _ptr(o, c_f32p))
        return o


class Oracle(_Checker):
    prefix = "mlorc_"

    def __init__(self):
        super().__init__(build_oracle())
        L = self.lib
        L.mlorc_chain_time.restype = ctypes.c_double
        L.mlorc_chain_time.argtypes = [c_i32p, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, c_f32p, c_u32p,
                                       c_f32p, c_f32p, c_f32p, ctypes.c_int]

    # ---- row plumbing / routing: same signatures as madronalib_amd.Engine's host-convenience methods ----
    def op_f32(self, op, a, b=None, c=None):
        return self.op(op, a, b, c).view(np.float32)

    def rows_map(self, rule, p0, p1, sample_rotate, src, src_rows, dst_rows, dst_offset, dst_step, count, groups, dst=None):
        fnc = self.lib.mlorc_rows_map
        fnc.restype = ctypes.c_int
        sz, lg = ctypes.c_size_t, ctypes.c_long
        fnc.argtypes = [ctypes.c_int, lg, lg, ctypes.c_int, c_f32p, sz, c_f32p, sz, sz, sz, sz, sz]
        src = np.ascontiguousarray(src, np.float32).reshape(-1, 64)
        if dst is None:
            dst = np.zeros((groups * dst_rows, 64), np.float32)
        dst = np.ascontiguousarray(dst, np.float32)
        r = fnc(int(rule), int(p0), int(p1), int(sample_rotate), _ptr(src, c_f32p), src_rows, _ptr(dst, c_f32p), dst_rows,
                dst_offset, dst_step, count, groups)
        assert r == 0, r
        return dst

    def rows_add(self, rows, rows_per_group, groups):
        fnc = self.lib.mlorc_rows_add
        fnc.argtypes = [c_f32p, ctypes.c_size_t, c_f32p, ctypes.c_size_t]
        rows = np.ascontiguousarray(rows, np.float32)
        out = np.empty((groups, 64), np.float32)
        assert fnc(_ptr(rows, c_f32p), rows_per_group, _ptr(out, c_f32p), groups) == 0
        return out

    def rows_normalize(self, rows):
        fnc = self.lib.mlorc_rows_normalize
        fnc.argtypes = [c_f32p, c_f32p, ctypes.c_size_t]
        rows = np.ascontiguousarray(rows, np.float32).reshape(-1, 64)
        out = np.empty_like(rows)
        assert fnc(_ptr(rows, c_f32p), _ptr(out, c_f32p), rows.shape[0]) == 0
        return out

    def rows_index(self, rows_per_group, groups):
        fnc = self.lib.mlorc_rows_index
        fnc.argtypes = [c_f32p, ctypes.c_size_t, ctypes.c_size_t]
        out = np.empty((groups * rows_per_group, 64), np.float32)
        assert fnc(_ptr(out, c_f32p), rows_per_group, groups) == 0
        return out

    def multiplex(self, selector, inputs, linear=False):
        fnc = self.lib.mlorc_multiplex
        fnc.argtypes = [c_f32p, ctypes.c_size_t, ctypes.POINTER(c_f32p), ctypes.c_int, c_f32p, ctypes.c_size_t, ctypes.c_int]
        sel = np.ascontiguousarray(selector, np.float32)
        ins = [np.ascontiguousarray(x, np.float32) for x in inputs]
        out = np.empty_like(ins[0])
        arr = (c_f32p * len(ins))(*[_ptr(x, c_f32p) for x in ins])
        assert fnc(_ptr(sel, c_f32p), sel.size, arr, len(ins), _ptr(out, c_f32p), out.size, 1 if linear else 0) == 0
        return out

    def demultiplex(self, selector, x, n_outputs, linear=False):
        fnc = self.lib.mlorc_demultiplex
        fnc.argtypes = [c_f32p, ctypes.c_size_t, c_f32p, ctypes.POINTER(c_f32p), ctypes.c_int, ctypes.c_size_t, ctypes.c_int]
        sel = np.ascontiguousarray(selector, np.float32)
        x = np.ascontiguousarray(x, np.float32)
        outs = [np.empty_like(x) for _ in range(n_outputs)]
        arr = (c_f32p * n_outputs)(*[_ptr(o, c_f32p) for o in outs])
        assert fnc(_ptr(sel, c_f32p), sel.size, _ptr(x, c_f32p), arr, n_outputs, x.size, 1 if linear else 0) == 0
        return outs

    def mixdown(self, sig, gains=None):
        fnc = self.lib.mlorc_mixdown
        fnc.argtypes = [c_f32p, ctypes.c_size_t, ctypes.c_size_t, c_f32p, c_f32p]
        sig = np.ascontiguousarray(sig, np.float32)
        V, S = sig.shape
        g = None if gains is None else np.ascontiguousarray(gains, np.float32)
        out = np.empty(S, np.float32)
        assert fnc(_ptr(sig, c_f32p), V, S // 64, _ptr(g, c_f32p), _ptr(out, c_f32p)) == 0
        return out

    def mixdown_shard(self, sig, gains=None):
        """One shard's hand-over rows [rows][64T] of the mixdown tree (mlgpu_bank_process_mixdown_shard): sig [Vs][64T], Vs % 64 == 0."""
        self.lib.mlorc_mixdown_shard_rows.restype = ctypes.c_size_t
        self.lib.mlorc_mixdown_shard_rows.argtypes = [ctypes.c_size_t]
        fnc = self.lib.mlorc_mixdown_shard
        fnc.argtypes = [c_f32p, ctypes.c_size_t, ctypes.c_size_t, c_f32p, c_f32p]
        sig = np.ascontiguousarray(sig, np.float32)
        V, S = sig.shape
        g = None if gains is None else np.ascontiguousarray(gains, np.float32)
        rows = np.empty((self.lib.mlorc_mixdown_shard_rows(V), S), np.float32)
        assert fnc(_ptr(sig, c_f32p), V, S // 64, _ptr(g, c_f32p), _ptr(rows, c_f32p)) == 0
        return rows

    def mixdown_rows(self, rows):
        """The host's finish of the tree over all shards' rows [n][64T] (mlgpu_mixdown_finish)."""
        fnc = self.lib.mlorc_mixdown_rows
        fnc.argtypes = [c_f32p, ctypes.c_size_t, ctypes.c_size_t, c_f32p]
        rows = np.ascontiguousarray(rows, np.float32)
        out = np.empty(rows.shape[1], np.float32)
        assert fnc(_ptr(rows, c_f32p), rows.shape[0], rows.shape[1] // 64, _ptr(out, c_f32p)) == 0
        return out

    def libm_sinf(self, x):
        """The restated glibc sinf (oracle/ml_oracle.c) on an array."""
        self.lib.mlorc_libm_sinf.restype = ctypes.c_float
        self.lib.mlorc_libm_sinf.argtypes = [ctypes.c_float]
        return np.array([self.lib.mlorc_libm_sinf(float(v)) for v in np.asarray(x, np.float32).ravel()], np.float32)

    def sinf_fast_check(self, which, lo, hi, n_threads=8):
        """(mismatch count, first offending bit pattern) of the device's cheaper sinf forms (0: direct, 1: [2^-12, pi]) restated
        in C against the host libm over bit patterns [lo, hi]."""
        fnc = self.lib.mlorc_sinf_fast_check
        fnc.restype = ctypes.c_uint64
        fnc.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(ctypes.c_uint32)]
        first = ctypes.c_uint32(0)
        n = fnc(int(which), int(lo), int(hi), int(n_threads), ctypes.byref(first))
        return int(n), int(first.value)

    def sinf_quadrant(self, y):
        self.lib.mlorc_sinf_quadrant.restype = ctypes.c_int
        self.lib.mlorc_sinf_quadrant.argtypes = [ctypes.c_float]
        return int(self.lib.mlorc_sinf_quadrant(float(y)))

    def sinf_check(self, lo, hi, n_threads=8):
        """(mismatch count, offending bit patterns) of restated vs host-libm sinf over bit patterns [lo, hi]."""
        fnc = self.lib.mlorc_sinf_check
        fnc.restype = ctypes.c_uint64
        fnc.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, c_u32p, ctypes.c_int]
        lst = np.zeros(64, np.uint32)
        n = fnc(int(lo), int(hi), int(n_threads), _ptr(lst, c_u32p), 64)
        return int(n), lst[:min(int(n), 64)]

    def chain_time(self, procs, T, coeffs, state, in_signal=None, in_const=None, n_threads=1):
        procs = np.ascontiguousarray(procs, np.int32)
        V = state.shape[1]
        coeffs = np.ascontiguousarray(coeffs, np.float32)
        return self.lib.mlorc_chain_time(_ptr(procs, c_i32p), len(procs), V, T, _ptr(coeffs, c_f32p),
                                         _ptr(state, c_u32p), _ptr(in_signal, c_f32p), _ptr(in_const, c_f32p),
                                         None, n_threads)


class Ref(_Checker):
    prefix = "mlref_"

    def __init__(self, path=None):
        path = path or build_ref()
        if path is None:
            raise FileNotFoundError("oracle/_ref/libmlref.so not built and /root/reference absent")
        super().__init__(path)
        L = self.lib
        sz = ctypes.c_size_t
        L.mlref_bench_saw_bandpass_gain.restype = ctypes.c_double
        L.mlref_bench_saw_bandpass_gain.argtypes = [sz, sz, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_float,
                                                    ctypes.c_int, c_f32p, c_f64p]
        L.mlref_bench_lopass_cascade8.restype = ctypes.c_double
        L.mlref_bench_lopass_cascade8.argtypes = [sz, sz, c_f32p, ctypes.c_int, c_f64p]
        L.mlref_bench_op.restype = ctypes.c_double
        L.mlref_bench_op.argtypes = [ctypes.c_int, c_f32p, c_f32p, sz, ctypes.c_int, ctypes.c_int]

    # ---- composites with one-vector feedback: the reference classes run as user code would (one voice) ----
    def allpass_run(self, which, gain, max_delay, d, delay_sig, x):
        fnc = self.lib.mlref_allpass_run
        fnc.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_float, ctypes.c_float, ctypes.c_float, c_f32p, c_f32p, c_f32p]
        x = np.ascontiguousarray(x, np.float32)
        ds = None if delay_sig is None else np.ascontiguousarray(delay_sig, np.float32)
        out = np.empty_like(x)
        assert fnc(which, x.size // 64, gain, max_delay, d, _ptr(ds, c_f32p), _ptr(x, c_f32p), _ptr(out, c_f32p)) == 0
        return out

    def fdn4_run(self, times, omegas, gains, max_delay, x):
        fnc = self.lib.mlref_fdn4_run
        fnc.argtypes = [ctypes.c_size_t, c_f32p, c_f32p, c_f32p, ctypes.c_float, c_f32p, c_f32p, c_f32p]
        x = np.ascontiguousarray(x, np.float32)
        a = [np.ascontiguousarray(v, np.float32) for v in (times, omegas, gains)]
        oL, oR = np.empty_like(x), np.empty_like(x)
        assert fnc(x.size // 64, *[_ptr(v, c_f32p) for v in a], max_delay, _ptr(x, c_f32p), _ptr(oL, c_f32p), _ptr(oR, c_f32p)) == 0
        return oL, oR

    def feedback_delay_run(self, feedback_gain, max_delay, lopass_coeffs, delay_sig, x):
        fnc = self.lib.mlref_feedback_delay_run
        fnc.argtypes = [ctypes.c_size_t, ctypes.c_float, ctypes.c_float, c_f32p, c_f32p, c_f32p, c_f32p]
        x = np.ascontiguousarray(x, np.float32)
        co = np.ascontiguousarray(lopass_coeffs, np.float32)
        ds = np.ascontiguousarray(delay_sig, np.float32)
        out = np.empty_like(x)
        assert fnc(x.size // 64, feedback_gain, max_delay, _ptr(co, c_f32p), _ptr(ds, c_f32p), _ptr(x, c_f32p), _ptr(out, c_f32p)) == 0
        return out

    def synth16_run(self, params, coeffs, seeds, gate, n_threads=1):
        """BASELINE configs[4]: the synth16 voice written with the reference's objects (oracle/ref_wrapper.cpp
        mlref_synth16_run). params: dict as patches.synth16 names them; coeffs: {lp, hp, smooth, dc, env}: [n][V]; gate
        [V][64 T]. Returns (out [V][64 T], seconds)."""
        fnc = self.lib.mlref_synth16_run
        fnc.restype = ctypes.c_double
        c_u32p = ctypes.POINTER(ctypes.c_uint32)
        fnc.argtypes = [ctypes.c_size_t, ctypes.c_size_t, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_u32p, c_f32p, c_f32p, ctypes.c_int]
        gate = np.ascontiguousarray(gate, np.float32)
        V, T = gate.shape[0], gate.shape[1] // 64
        P = np.ascontiguousarray(np.stack([np.broadcast_to(np.asarray(params[k], np.float32), (V,))
                                           for k in ("pitch", "baseFreq", "width", "lfoFreq", "noiseLevel")]), np.float32)
        C = {k: np.ascontiguousarray(coeffs[k], np.float32) for k in ("lp", "hp", "smooth", "dc", "env")}
        seeds = np.ascontiguousarray(seeds, np.uint32)
        out = np.empty_like(gate)
        sec = fnc(V, T, _ptr(P, c_f32p), _ptr(C["lp"], c_f32p), _ptr(C["hp"], c_f32p), _ptr(C["smooth"], c_f32p), _ptr(C["dc"], c_f32p),
                  _ptr(C["env"], c_f32p), seeds.ctypes.data_as(c_u32p), _ptr(gate, c_f32p), _ptr(out, c_f32p), int(n_threads))
        return out, sec

    def synth16full_run(self, params, coeffs, seeds, gate, n_threads=1):
        """patches.synth16(full=True) written with the reference's objects (mlref_synth16full_run): the patch SURVEY 8d lists,
        with the filter envelope and Lopass(x, omega, k). coeffs: {hp, smooth, dc, env, fenv}. Returns (out, seconds)."""
        fnc = self.lib.mlref_synth16full_run
        fnc.restype = ctypes.c_double
        c_u32p = ctypes.POINTER(ctypes.c_uint32)
        fnc.argtypes = [ctypes.c_size_t, ctypes.c_size_t] + [c_f32p] * 6 + [c_u32p, c_f32p, c_f32p, ctypes.c_int]
        gate = np.ascontiguousarray(gate, np.float32)
        V, T = gate.shape[0], gate.shape[1] // 64
        names = ("pitch", "baseFreq", "width", "lfoFreq", "noiseLevel", "cutoffOct", "envAmount", "cutoffBase", "resonance")
        P = np.ascontiguousarray(np.stack([np.broadcast_to(np.asarray(params[k], np.float32), (V,)) for k in names]), np.float32)
        C = {k: np.ascontiguousarray(coeffs[k], np.float32) for k in ("hp", "smooth", "dc", "env", "fenv")}
        seeds = np.ascontiguousarray(seeds, np.uint32)
        out = np.empty_like(gate)
        sec = fnc(V, T, _ptr(P, c_f32p), _ptr(C["hp"], c_f32p), _ptr(C["smooth"], c_f32p), _ptr(C["dc"], c_f32p), _ptr(C["env"], c_f32p),
                  _ptr(C["fenv"], c_f32p), seeds.ctypes.data_as(c_u32p), _ptr(gate, c_f32p), _ptr(out, c_f32p), int(n_threads))
        return out, sec

    def rate_allpass_run(self, up, gain, max_delay, delay, one_pole_coeffs, x):
        """Upsample2xFunction<1> / Downsample2xFunction<1> around fn = OnePole(Allpass<IntegerDelay>(x)) (mlref_rate_allpass_run)."""
        fnc = self.lib.mlref_rate_allpass_run
        fnc.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_float, ctypes.c_float, ctypes.c_float, c_f32p, c_f32p, c_f32p]
        x = np.ascontiguousarray(x, np.float32)
        co = np.ascontiguousarray(one_pole_coeffs, np.float32)
        out = np.empty_like(x)
        assert fnc(1 if up else 0, x.shape[0], x.shape[1] // 64, gain, max_delay, delay, _ptr(co, c_f32p), _ptr(x, c_f32p), _ptr(out, c_f32p)) == 0
        return out

    def rate_nested_run(self, outer_up, inner_up, lopass_coeffs, one_pole_coeffs, x):
        """Outer(mid, x), mid(v) = OnePole(Inner(fn, v)) + v/2, fn(w) = Lopass(clamp(3w, -1, 1)); Outer / Inner are
        Upsample2xFunction<1> (True) or Downsample2xFunction<1> (False) (mlref_rate_nested_run)."""
        fnc = self.lib.mlref_rate_nested_run
        fnc.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, c_f32p, c_f32p, c_f32p, c_f32p]
        x = np.ascontiguousarray(x, np.float32)
        lp = np.ascontiguousarray(lopass_coeffs, np.float32)
        op = np.ascontiguousarray(one_pole_coeffs, np.float32)
        out = np.empty_like(x)
        assert fnc(1 if outer_up else 0, 1 if inner_up else 0, x.shape[0], x.shape[1] // 64, _ptr(lp, c_f32p), _ptr(op, c_f32p), _ptr(x, c_f32p), _ptr(out, c_f32p)) == 0
        return out

    def dspbuffer(self, size):
        return _RefDSPBuffer(self.lib, size)

    def rows_case(self, name, inputs, max_rows=16):
        """The reference's own row-plumbing / routing template instantiation called `name` (oracle/ref_wrapper.cpp:
        mlref_rows_case). inputs: list of [rows][64] arrays. Returns the output rows [r][64]."""
        fnc = self.lib.mlref_rows_case
        fnc.restype = ctypes.c_int
        fnc.argtypes = [ctypes.c_char_p, ctypes.POINTER(c_f32p), c_f32p]
        ins = [np.ascontiguousarray(x, np.float32) for x in inputs]
        arr = (c_f32p * max(1, len(ins)))(*[_ptr(x, c_f32p) for x in ins])
        out = np.zeros((max_rows, 64), np.float32)
        r = fnc(name.encode(), arr, _ptr(out, c_f32p))
        assert r >= 0, name
        return out[:r].copy()

    def bench_saw_bandpass_gain(self, V, T, freq, g0, g1, g2, gain, n_threads, out=None):
        sink = ctypes.c_double(0)
        arrs = [np.ascontiguousarray(x, np.float32) for x in (freq, g0, g1, g2)]
        s = self.lib.mlref_bench_saw_bandpass_gain(V, T, *[_ptr(a, c_f32p) for a in arrs], float(gain), n_threads,
                                                    _ptr(out, c_f32p), ctypes.byref(sink))
        return s, sink.value

    def bench_lopass_cascade8(self, V, T, coeffs8x3, n_threads):
        sink = ctypes.c_double(0)
        c = np.ascontiguousarray(coeffs8x3, np.float32)
        s = self.lib.mlref_bench_lopass_cascade8(V, T, _ptr(c, c_f32p), n_threads, ctypes.byref(sink))
        return s, sink.value

    def bench_op(self, op, x, n_threads, reps):
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty_like(x)
        return self.lib.mlref_bench_op(int(op), _ptr(x, c_f32p), _ptr(out, c_f32p), x.size, n_threads, reps)


class _RefDSPBuffer:
    """The reference's own DSPBuffer (MLDSPBuffer.h) behind oracle/ref_wrapper.cpp, same method names as
    madronalib_amd.DSPBuffer."""

    def __init__(self, lib, size):
        self.L = lib
        sz, vp = ctypes.c_size_t, ctypes.c_void_p
        for name, res, args in [("create", vp, [ctypes.c_int]), ("destroy", None, [vp]), ("read_available", sz, [vp]),
                                ("write_available", sz, [vp]), ("write", None, [vp, c_f32p, sz]), ("read", sz, [vp, c_f32p, sz]),
                                ("discard", None, [vp, sz]), ("clear", None, [vp]), ("write_overlap_add", None, [vp, c_f32p, sz, sz]),
                                ("read_overlap", None, [vp, c_f32p, sz, sz]), ("peek_most_recent", None, [vp, c_f32p, sz])]:
            f = getattr(lib, "mlref_dspbuffer_" + name)
            f.restype, f.argtypes = res, args
        self.h = ctypes.c_void_p(lib.mlref_dspbuffer_create(int(size)))

    def __del__(self):
        try:
            self.L.mlref_dspbuffer_destroy(self.h)
        except Exception:
            pass

    def read_available(self):
        return self.L.mlref_dspbuffer_read_available(self.h)

    def write_available(self):
        return self.L.mlref_dspbuffer_write_available(self.h)

    def write(self, x):
        x = np.ascontiguousarray(x, np.float32)
        self.L.mlref_dspbuffer_write(self.h, _ptr(x, c_f32p), x.size)

    def read(self, n):
        out = np.full(n, np.float32(-99.0))
        got = self.L.mlref_dspbuffer_read(self.h, _ptr(out, c_f32p), n)
        return out[:got].copy()

    def discard(self, n):
        self.L.mlref_dspbuffer_discard(self.h, n)

    def clear(self):
        self.L.mlref_dspbuffer_clear(self.h)

    def write_with_overlap_add(self, x, overlap):
        x = np.ascontiguousarray(x, np.float32)
        self.L.mlref_dspbuffer_write_overlap_add(self.h, _ptr(x, c_f32p), x.size, overlap)

    def read_with_overlap(self, n, overlap):
        out = np.full(n, np.float32(-99.0))
        self.L.mlref_dspbuffer_read_overlap(self.h, _ptr(out, c_f32p), n, overlap)
        return out

    def peek_most_recent(self, n):
        out = np.full(n, np.float32(-99.0))
        self.L.mlref_dspbuffer_peek_most_recent(self.h, _ptr(out, c_f32p), n)
        return out


def host_threads(cap=64):
    """Threads this process may really use (CPU affinity, cut to the cgroup quota): what the heavy full-size checks run the CPU side on."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def fast_checker():
    """The CPU checker for full-size comparisons: the compiled reference when it was built (several times the plain-C port's speed; the
    two are pinned to each other by tests/test_oracle_vs_ref.py), else None - the caller then keeps to its subset of voices."""
    return Ref() if ref_available() else None


def ref_available():
    try:
        return build_ref() is not None
    except Exception:
        return False
