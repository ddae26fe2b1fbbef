"""madronalib_amd — MI355X-native evaluation of madronalib's mldsp.h hot path.

Thin Python host over the C-ABI in include/mlgpu.h (madronalib_amd/csrc/libmlgpu.so). Python
is used by the tests and bench only; the product is the C-ABI library and the C++ host
mirror in include/mlgpu/*.hpp. Names follow the reference (source/DSP/MLDSPFilters.h,
MLDSPGens.h, MLDSPFunctional.h): `Lopass.makeCoeffs`, `Bank`, `clear`, ...

There is no CPU fallback: every compute call goes to a gfx950 kernel or raises MlgpuError.
"""
import ctypes
import weakref

import numpy as np

from . import _lib
from .constants import FLOATS_PER_DSPVECTOR, Layout, Op, Proc, Region, Route, RowOp, RowsRule, Status, Vop

__all__ = ["Engine", "Bank", "Graph", "DSPBuffer", "ProcessBuffer", "Resampler", "Events", "Event", "EventType", "jit_selftest", "DeviceBuffer", "MlgpuError", "Layout", "Op", "Proc", "RowOp", "Status", "Vop", "Route", "RowsRule", "Allpass1", "FractionalDelay", "LinearGlide", "SampleAccurateLinearGlide",
           "Lopass", "Hipass", "Bandpass", "LoShelf", "HiShelf", "Bell", "OnePole", "DCBlocker", "ADSR",
           "dBToGain", "device_count", "FLOATS_PER_DSPVECTOR"]


BUSY = 7   # MLGPU_ERR_BUSY: a job started on the object has not finished (Graph.compile_async)


def behaviour_revision():
    """mlgpu_behaviour_revision(): goes up when a documented behaviour of an existing entry point changes under an unchanged ABI."""
    return _lib.load().mlgpu_behaviour_revision()


class MlgpuError(RuntimeError):
    def __init__(self, status, detail=""):
        L = _lib.load()
        self.status = status
        super().__init__(f"mlgpu status {status} ({L.mlgpu_status_string(status).decode()}) {detail}")


def mixdown_shard_level(voices_per_shard):
    """The level of the mixdown tree a shard of that many voices hands over (0: not a multiple of 64, no exact hand-over)."""
    return int(_lib.load().mlgpu_mixdown_shard_level(int(voices_per_shard)))


def mixdown_shard_rows(voices_per_shard):
    return int(_lib.load().mlgpu_mixdown_shard_rows(int(voices_per_shard)))


def mixdown_finish(rows, flush_denormals=False):
    """The host's finish of the mixdown tree over all shards' rows, [n_rows][64 T] numpy in voice order -> [64 T] (mlgpu_mixdown_finish)."""
    rows = np.ascontiguousarray(rows, np.float32)
    n, S = rows.shape
    out = np.empty(S, np.float32)
    nscr = ((n + 63) // 64 + (n + 4095) // 4096) * S
    scratch = np.empty(max(1, nscr), np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    st = _lib.load().mlgpu_mixdown_finish(rows.ctypes.data_as(fp), n, S // 64, int(bool(flush_denormals)), out.ctypes.data_as(fp), scratch.ctypes.data_as(fp))
    if st:
        raise MlgpuError(st, "(mixdown_finish)")
    return out


def jit_cache_export():
    """Every generated kernel this process holds, as one bundle (bytes): for an installation without hiprtc (mlgpu_jit_cache_export)."""
    L = _lib.load()
    need = ctypes.c_size_t()
    st = L.mlgpu_jit_cache_export(None, 0, ctypes.byref(need))
    if st:
        raise MlgpuError(st, "(jit_cache_export)")
    buf = ctypes.create_string_buffer(need.value)
    st = L.mlgpu_jit_cache_export(buf, need.value, ctypes.byref(need))
    if st:
        raise MlgpuError(st, "(jit_cache_export)")
    return buf.raw[:need.value]


def jit_cache_import(bundle):
    """Take a bundle made by jit_cache_export (the same build of the device code); -> kernels taken."""
    n = ctypes.c_size_t()
    st = _lib.load().mlgpu_jit_cache_import(bundle, len(bundle), ctypes.byref(n))
    if st:
        raise MlgpuError(st, "(jit_cache_import)")
    return n.value


def jit_compiler_available():
    return bool(_lib.load().mlgpu_jit_compiler_available())


def device_source_hash():
    """SHA-256 of the device sources and compiler flags the loaded library was built from (mlgpu_device_source_hash)."""
    return _lib.load().mlgpu_device_source_hash().decode()


def jit_stats():
    """hiprtc work since the library was loaded: kernels compiled (and seconds), disk-cache hits (and seconds), memory hits."""
    L = _lib.load()
    a, b, c_ = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
    t, u = ctypes.c_double(), ctypes.c_double()
    L.mlgpu_jit_stats(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c_), ctypes.byref(t), ctypes.byref(u))
    return dict(compiles=a.value, disk_hits=b.value, memory_hits=c_.value, compile_seconds=t.value, disk_load_seconds=u.value)


class _RegistryEntry(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("node_type", ctypes.c_int), ("kind", ctypes.c_int), ("n_inputs", ctypes.c_int),
                ("n_required_inputs", ctypes.c_int), ("n_params", ctypes.c_int), ("output_name", ctypes.c_char_p)]


def registry():
    """Every processor / op a graph can hold, by name: {name: dict(node_type, kind, inputs [names], required_inputs, params
    [coefficient names], output)} (mlgpu_registry_*; the naming convention of source/procs/MLProcMultiply.cpp)."""
    L = _lib.load()
    out = {}
    buf = ctypes.create_string_buffer(64)
    for i in range(L.mlgpu_registry_count()):
        e = _RegistryEntry()
        assert L.mlgpu_registry_get(i, ctypes.byref(e)) == 0
        ins, pars = [], []
        for k in range(e.n_inputs):
            assert L.mlgpu_registry_input_name(e.name, k, buf, 64) == 0
            ins.append(buf.value.decode())
        for k in range(e.n_params):
            assert L.mlgpu_registry_param_name(e.name, k, buf, 64) == 0
            pars.append(buf.value.decode())
        out[e.name.decode()] = dict(node_type=e.node_type, kind=e.kind, inputs=ins, required_inputs=e.n_required_inputs, params=pars,
                                    output=e.output_name.decode())
    return out


def device_count():
    return _lib.load().mlgpu_device_count()


def _np_ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class DeviceBuffer:
    """A caller-owned HBM allocation (mlgpu_alloc)."""

    def __init__(self, engine, nbytes):
        self.engine = engine
        self.nbytes = int(nbytes)
        p = ctypes.c_void_p()
        engine._check(engine.L.mlgpu_alloc(engine.h, self.nbytes, ctypes.byref(p)))
        self.ptr = p.value or 0
        engine._children.add(self)

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.engine._check(self.engine.L.mlgpu_upload(self.engine.h, self.ptr, _np_ptr(arr), arr.nbytes))
        return self

    def download(self, dtype=np.float32, count=None):
        dtype = np.dtype(dtype)
        n = self.nbytes // dtype.itemsize if count is None else int(count)
        out = np.empty(n, dtype)
        self.engine._check(self.engine.L.mlgpu_download(self.engine.h, _np_ptr(out), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr and self.engine.h:
            self.engine.L.mlgpu_free(self.engine.h, self.ptr)
        self.ptr = 0

    close = free

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Engine:
    """One MI355X + one HIP stream (mlgpu_engine)."""

    def __init__(self, device=0, stream=None, urgency=0):
        self.L = _lib.load()
        h = ctypes.c_void_p()
        if stream is None and urgency:
            st = self.L.mlgpu_engine_create_urgency(int(device), int(urgency), ctypes.byref(h))   # +1 urgent, -1 background
        elif stream is None:
            st = self.L.mlgpu_engine_create(int(device), ctypes.byref(h))
        else:
            st = self.L.mlgpu_engine_create_on_stream(int(device), ctypes.c_void_p(int(stream)), ctypes.byref(h))
        if st != 0:
            raise MlgpuError(st, "(engine_create)")
        self.h = h
        self.device = int(device)
        self._children = weakref.WeakSet()  # banks and buffers: released before the engine goes away

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != 0:
            raise MlgpuError(st, self.L.mlgpu_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            for child in list(self._children):
                child.close()
            self.L.mlgpu_engine_destroy(self.h)
            self.h = None

    def sync(self):
        self._check(self.L.mlgpu_engine_sync(self.h))

    def set_flush_denormals(self, on):
        """ml::UsingFlushDenormalsToZero for every kernel launched from now on (MLDSPUtils.h:51-96)."""
        self._check(self.L.mlgpu_engine_set_flush_denormals(self.h, 1 if on else 0))

    def get_flush_denormals(self):
        return bool(self.L.mlgpu_engine_get_flush_denormals(self.h))

    def fence(self):
        """A point another engine of the same device can wait for (mlgpu_fence): `a.signal(f)` ... `b.wait(f)`."""
        return Fence(self)

    def signal(self, fence):
        self._check(self.L.mlgpu_engine_signal(self.h, fence.h))

    def wait(self, fence):
        self._check(self.L.mlgpu_engine_wait(self.h, fence.h))

    def set_strict_svf(self, on):
        """Banks and graphs created from now on update SVF memories with two instructions (`ic + 2 t`) instead of one fused."""
        self._check(self.L.mlgpu_engine_set_strict_svf(self.h, 1 if on else 0))

    def get_strict_svf(self):
        return bool(self.L.mlgpu_engine_get_strict_svf(self.h))

    def set_cascade_lanes(self, lanes):
        """Wavefront lanes per channel of an SVF-cascade bank: 0 = by bank size, 1 / 2 / 4, -1 = the round-2 kernel."""
        self._check(self.L.mlgpu_engine_set_cascade_lanes(self.h, int(lanes)))

    def get_cascade_lanes(self):
        return int(self.L.mlgpu_engine_get_cascade_lanes(self.h))

    def set_jit(self, enabled):
        """hiprtc fusion of chains that have no ahead-of-time kernel (default on)."""
        self._check(self.L.mlgpu_engine_set_jit(self.h, 1 if enabled else 0))

    @property
    def stream(self):
        return self.L.mlgpu_engine_stream(self.h)

    def device_info(self):
        name = ctypes.create_string_buffer(256)
        cu = ctypes.c_int()
        mem = ctypes.c_uint64()
        self._check(self.L.mlgpu_device_info(self.device, name, 256, ctypes.byref(cu), ctypes.byref(mem)))
        pci = ctypes.create_string_buffer(64)
        self._check(self.L.mlgpu_device_pci_bus_id(self.device, pci, 64))
        return dict(name=name.value.decode(), cu_count=cu.value, mem_bytes=mem.value, pci_bus_id=pci.value.decode())

    def device_sync(self):
        """hipDeviceSynchronize on this engine's device (all streams)."""
        self._check(self.L.mlgpu_device_synchronize(self.device))

    # ---- memory ----
    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        return DeviceBuffer(self, max(arr.nbytes, 16)).upload(arr)

    def validate(self, d_signal, n_elems):
        """ml::validate over a device signal: (number of NaN / |x| > 1e8 samples, flat index of the first or None)."""
        cnt, first = ctypes.c_uint64(), ctypes.c_uint64()
        self._check(self.L.mlgpu_validate(self.h, ctypes.c_void_p(d_signal.ptr if hasattr(d_signal, "ptr") else int(d_signal)), int(n_elems),
                                          ctypes.byref(cnt), ctypes.byref(first)))
        return cnt.value, (None if cnt.value == 0 else first.value)

    def timer_start(self):
        self._check(self.L.mlgpu_timer_start(self.h))

    def timer_stop_ms(self):
        ms = ctypes.c_float()
        self._check(self.L.mlgpu_timer_stop_ms(self.h, ctypes.byref(ms)))
        return ms.value

    def lap_times_ms(self, work, laps):
        """Run work() `laps` times with an event after each on the engine's stream; the durations between consecutive events (ms)."""
        self._check(self.L.mlgpu_timer_laps_begin(self.h, int(laps)))
        for _ in range(laps):
            work()
            self._check(self.L.mlgpu_timer_lap(self.h))
        out = (ctypes.c_float * laps)()
        n = ctypes.c_size_t()
        self._check(self.L.mlgpu_timer_laps_end(self.h, out, laps, ctypes.byref(n)))
        return np.frombuffer(out, dtype=np.float32, count=n.value).copy()

    # ---- stateless ops on device buffers ----
    def op_apply(self, op, a, b, c, out, n_elems):
        g = lambda x: None if x is None else ctypes.c_void_p(x.ptr)  # noqa: E731
        self._check(self.L.mlgpu_op_apply(self.h, int(op), g(a), g(b), g(c), g(out), int(n_elems)))

    def op(self, op, a, b=None, c=None):
        """Host convenience: upload operands, run the op kernel, download the result (uint32 bits)."""
        a = np.ascontiguousarray(a)
        n = a.size
        bufs = [None if x is None else self.to_device(np.ascontiguousarray(x)) for x in (a, b, c)]
        out = self.alloc(max(4 * n, 16))
        self.op_apply(op, bufs[0], bufs[1], bufs[2], out, n)
        return out.download(np.uint32, n).reshape(a.shape)

    def op_f32(self, op, a, b=None, c=None):
        return self.op(op, a, b, c).view(np.float32)

    def op_rows1(self, op, a, b64):
        a = np.ascontiguousarray(a, np.float32)
        da, db = self.to_device(a), self.to_device(np.ascontiguousarray(b64, np.float32))
        out = self.alloc(a.nbytes)
        self._check(self.L.mlgpu_op_apply_rows1(self.h, int(op), da.ptr, db.ptr, out.ptr, a.size // 64))
        return out.download(np.float32, a.size).reshape(a.shape)

    def row_reduce(self, rowop, rows):
        rows = np.ascontiguousarray(rows, np.float32)
        d = self.to_device(rows)
        out = self.alloc(max(16, 4 * (rows.size // 64)))
        self._check(self.L.mlgpu_row_reduce(self.h, int(rowop), d.ptr, out.ptr, rows.size // 64))
        return out.download(np.float32, rows.size // 64)

    # ---- row plumbing and routing (host convenience: numpy rows [n][64] in, numpy out) ----
    def rows_map(self, rule, p0, p1, sample_rotate, src, src_rows, dst_rows, dst_offset, dst_step, count, groups, dst=None):
        """mlgpu_rows_map on host arrays. `dst` (numpy [groups*dst_rows][64]) is updated and returned, so the
        reference's multi-call functions (concatRows, shuffleRows) can accumulate into one destination."""
        src = np.ascontiguousarray(src, np.float32).reshape(-1, 64)
        if dst is None:
            dst = np.zeros((groups * dst_rows, 64), np.float32)
        d_src, d_dst = self.to_device(src), self.to_device(dst)
        self._check(self.L.mlgpu_rows_map(self.h, int(rule), int(p0), int(p1), int(sample_rotate), d_src.ptr, int(src_rows),
                                          d_dst.ptr, int(dst_rows), int(dst_offset), int(dst_step), int(count), int(groups)))
        return d_dst.download(np.float32, dst.size).reshape(dst.shape)

    def rows_add(self, rows, rows_per_group, groups):
        rows = np.ascontiguousarray(rows, np.float32)
        d, out = self.to_device(rows), self.alloc(groups * 256)
        self._check(self.L.mlgpu_rows_add(self.h, d.ptr, int(rows_per_group), out.ptr, int(groups)))
        return out.download(np.float32, groups * 64).reshape(groups, 64)

    def rows_normalize(self, rows):
        rows = np.ascontiguousarray(rows, np.float32).reshape(-1, 64)
        d, out = self.to_device(rows), self.alloc(rows.nbytes)
        self._check(self.L.mlgpu_rows_normalize(self.h, d.ptr, out.ptr, rows.shape[0]))
        return out.download(np.float32, rows.size).reshape(rows.shape)

    def rows_index(self, rows_per_group, groups):
        out = self.alloc(groups * rows_per_group * 256)
        self._check(self.L.mlgpu_rows_index(self.h, out.ptr, int(rows_per_group), int(groups)))
        return out.download(np.float32, groups * rows_per_group * 64).reshape(-1, 64)

    def multiplex(self, selector, inputs, linear=False):
        sel = np.ascontiguousarray(selector, np.float32)
        ins = [np.ascontiguousarray(x, np.float32) for x in inputs]
        n = ins[0].size
        d_sel, d_ins, out = self.to_device(sel), [self.to_device(x) for x in ins], self.alloc(4 * n)
        arr = (ctypes.c_void_p * len(ins))(*[b.ptr for b in d_ins])
        self._check(self.L.mlgpu_multiplex(self.h, d_sel.ptr, sel.size, arr, len(ins), out.ptr, n, 1 if linear else 0))
        return out.download(np.float32, n).reshape(ins[0].shape)

    def demultiplex(self, selector, x, n_outputs, linear=False):
        sel = np.ascontiguousarray(selector, np.float32)
        x = np.ascontiguousarray(x, np.float32)
        d_sel, d_x = self.to_device(sel), self.to_device(x)
        outs = [self.alloc(4 * x.size) for _ in range(n_outputs)]
        arr = (ctypes.c_void_p * n_outputs)(*[b.ptr for b in outs])
        self._check(self.L.mlgpu_demultiplex(self.h, d_sel.ptr, sel.size, d_x.ptr, arr, n_outputs, x.size, 1 if linear else 0))
        return [o.download(np.float32, x.size).reshape(x.shape) for o in outs]

    def mixdown_reserve(self, max_voices, max_vectors):
        """Scratch for mixdown's partial sums: once, at setup (mlgpu_mixdown never allocates)."""
        self._check(self.L.mlgpu_mixdown_reserve(self.h, int(max_voices), int(max_vectors)))

    def mixdown(self, d_signal, layout, n_voices, n_vectors, d_out, d_gains=None):
        g = lambda x: None if x is None else ctypes.c_void_p(x.ptr if hasattr(x, "ptr") else int(x))  # noqa: E731
        self._check(self.L.mlgpu_mixdown(self.h, g(d_signal), int(layout), int(n_voices), int(n_vectors), g(d_gains), g(d_out)))

    def mixdown_shard(self, d_signal, layout, n_voices, n_vectors, d_rows, d_gains=None):
        """This engine's SHARD of a larger bank: its voices' mixdown up to the hand-over level, shard_rows(n_voices) rows of 64 * n_vectors
        floats for mixdown_finish (mlgpu_mixdown_shard)."""
        g = lambda x: None if x is None else ctypes.c_void_p(x.ptr if hasattr(x, "ptr") else int(x))  # noqa: E731
        self._check(self.L.mlgpu_mixdown_shard(self.h, g(d_signal), int(layout), int(n_voices), int(n_vectors), g(d_gains), g(d_rows)))

    def mixdown_groups(self, d_signal, layout, n_groups, group_size, n_vectors, d_out, out_layout=Layout.QUAD):
        """Sum every `group_size` consecutive voices (a Synth's voices, MLSynth.h:43-57) into one signal per group."""
        self._check(self.L.mlgpu_mixdown_groups(self.h, ctypes.c_void_p(d_signal.ptr), int(layout), int(n_groups), int(group_size), int(n_vectors),
                                                ctypes.c_void_p(d_out.ptr), int(out_layout)))

    def layout_convert(self, src, src_layout, dst, dst_layout, n_voices, n_vectors):
        self._check(self.L.mlgpu_layout_convert(self.h, src.ptr, int(src_layout), dst.ptr, int(dst_layout),
                                                int(n_voices), int(n_vectors)))

    def bank(self, procs, n_voices):
        return Bank(self, procs, n_voices)

    def record(self):
        """Context manager: the launches made inside are captured into a Sequence (one hipGraph launch per replay).
            with eng.record() as seq: bank.process(...); graph.process(...)
            seq.launch()"""
        return _Recording(self)


class Fence:
    """Ordering between two engines (= two HIP streams) of one device; see Engine.signal / Engine.wait."""

    def __init__(self, engine):
        self.engine, self.L = engine, engine.L
        h = ctypes.c_void_p()
        engine._check(self.L.mlgpu_fence_create(engine.h, ctypes.byref(h)))
        self.h = h
        engine._children.add(self)

    def close(self):
        if getattr(self, "h", None):
            self.L.mlgpu_fence_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Sequence:
    """A recorded launch sequence (mlgpu_sequence)."""

    def __init__(self, engine):
        self.engine, self.h = engine, None

    def launch(self):
        self.engine._check(self.engine.L.mlgpu_sequence_launch(self.h))

    @property
    def num_nodes(self):
        return int(self.engine.L.mlgpu_sequence_num_nodes(self.h))

    def close(self):
        if self.h and self.engine.h:
            self.engine.L.mlgpu_sequence_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Recording:
    def __init__(self, engine):
        self.engine, self.seq = engine, Sequence(engine)

    def __enter__(self):
        self.engine._check(self.engine.L.mlgpu_engine_begin_recording(self.engine.h))
        return self.seq

    def __exit__(self, exc_type, exc, tb):
        h = ctypes.c_void_p()
        st = self.engine.L.mlgpu_engine_end_recording(self.engine.h, ctypes.byref(h))
        if exc_type is None:
            self.engine._check(st)
            self.seq.h = h
            self.engine._children.add(self.seq)
        return False


class DSPBuffer:
    """The reference's DSPBuffer (MLDSPBuffer.h): a host SPSC float ring (mlgpu_dspbuffer)."""

    def __init__(self, size):
        self.L = _lib.load()
        self.h = ctypes.c_void_p(self.L.mlgpu_dspbuffer_create())
        self.size = self.L.mlgpu_dspbuffer_resize(self.h, int(size))

    def __del__(self):
        try:
            if self.h:
                self.L.mlgpu_dspbuffer_destroy(self.h)
            self.h = None
        except Exception:
            pass

    def read_available(self):
        return self.L.mlgpu_dspbuffer_read_available(self.h)

    def write_available(self):
        return self.L.mlgpu_dspbuffer_write_available(self.h)

    def write(self, x):
        x = np.ascontiguousarray(x, np.float32)
        self.L.mlgpu_dspbuffer_write(self.h, _np_ptr(x), x.size)

    def read(self, n):
        out = np.full(n, np.float32(-99.0))
        got = self.L.mlgpu_dspbuffer_read(self.h, _np_ptr(out), n)
        return out[:got].copy()

    def read_vector(self):
        out = np.full(64, np.float32(-99.0))
        ok = self.L.mlgpu_dspbuffer_read_vector(self.h, _np_ptr(out))
        return bool(ok), out

    def discard(self, n):
        self.L.mlgpu_dspbuffer_discard(self.h, n)

    def clear(self):
        self.L.mlgpu_dspbuffer_clear(self.h)

    def write_with_overlap_add(self, x, overlap):
        x = np.ascontiguousarray(x, np.float32)
        self.L.mlgpu_dspbuffer_write_with_overlap_add(self.h, _np_ptr(x), x.size, overlap)

    def read_with_overlap(self, n, overlap):
        out = np.full(n, np.float32(-99.0))
        self.L.mlgpu_dspbuffer_read_with_overlap(self.h, _np_ptr(out), n, overlap)
        return out

    def peek_most_recent(self, n):
        out = np.full(n, np.float32(-99.0))
        self.L.mlgpu_dspbuffer_peek_most_recent(self.h, _np_ptr(out), n)
        return out


WINDOW_SHAPES = {"rectangle": 0, "triangle": 1, "raisedCosine": 2, "hamming": 3, "blackman": 4, "flatTop": 5}


def make_window(size, shape="triangle"):
    """makeWindow(dest, size, dspwindows::<shape>) (source/DSP/MLDSPUtils.h:22-47): a host table of `size` floats."""
    out = np.empty(int(size), np.float32)
    st = _lib.load().mlgpu_make_window(_np_ptr(out), int(size), WINDOW_SHAPES[shape] if isinstance(shape, str) else int(shape))
    if st != 0:
        raise ValueError(f"make_window: unknown shape {shape!r}")
    return out


class ProcessBuffer:
    """The reference's SignalProcessBuffer (MLSignalProcessBuffer.h) over the engine: host blocks of any size in and
    out, `fn(n_vectors, d_inputs, d_outputs)` called once per block with single-voice device signals (raw pointers)."""

    _CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p))

    def __init__(self, engine, n_inputs, n_outputs, max_frames):
        self.engine, self.L = engine, engine.L
        self.n_in, self.n_out = n_inputs, n_outputs
        h = ctypes.c_void_p()
        engine._check(self.L.mlgpu_process_buffer_create(engine.h, n_inputs, n_outputs, max_frames, ctypes.byref(h)))
        self.h = h
        engine._children.add(self)

    def close(self):
        if getattr(self, "h", None) and self.engine.h:
            self.L.mlgpu_process_buffer_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_pipelined(self, on=True):
        """Double-buffered mode: a call returns at once with what earlier calls computed; fixed delay latency_frames()."""
        self.engine._check(self.L.mlgpu_process_buffer_set_pipelined(self.h, 1 if on else 0))

    def latency_frames(self):
        return int(self.L.mlgpu_process_buffer_latency_frames(self.h))

    def process(self, inputs, n_frames, fn):
        """inputs: list of n_frames-float arrays (or None). Returns the list of output blocks."""
        ins = [None if x is None else np.ascontiguousarray(x, np.float32) for x in inputs]
        outs = [np.zeros(n_frames, np.float32) for _ in range(self.n_out)]
        pin = (ctypes.c_void_p * max(1, self.n_in))(*[None if x is None else x.ctypes.data for x in ins])
        pout = (ctypes.c_void_p * max(1, self.n_out))(*[o.ctypes.data for o in outs])
        err = []

        def cb(_user, n_vectors, d_in, d_out):
            try:
                fn(int(n_vectors), [d_in[i] for i in range(self.n_in)], [d_out[i] for i in range(self.n_out)])
                return 0
            except MlgpuError as ex:  # pragma: no cover
                err.append(ex)
                return ex.status
        cbf = self._CB(cb)
        st = self.L.mlgpu_process_buffer_process(self.h, pin, pout, int(n_frames), cbf, None)
        if err:
            raise err[0]
        self.engine._check(st)
        return outs


class EventType:  # ml::EventType, source/app/MLEvent.h:13-26
    NULL, NOTE_ON, NOTE_RETRIG, NOTE_SUSTAIN, NOTE_OFF, SUSTAIN_PEDAL, CONTROLLER, PITCH_BEND, NOTE_PRESSURE, CHANNEL_PRESSURE, PROGRAM_CHANGE = range(11)


class Event(ctypes.Structure):  # mlgpu_event == ml::Event
    _fields_ = [("type", ctypes.c_uint8), ("channel", ctypes.c_uint8), ("source_idx", ctypes.c_uint16), ("time", ctypes.c_int32),
                ("value1", ctypes.c_float), ("value2", ctypes.c_float)]


class Events:
    """EventsToSignals (source/app/MLEventsToSignals.h) for n_instruments instruments of `polyphony` voices (mlgpu_events)."""
    ROWS = ("pitch", "gate", "vox", "z", "x", "y", "mod", "time")

    def __init__(self, engine, n_instruments, polyphony, sr=48000.0):
        self.engine, self.L = engine, engine.L
        h = ctypes.c_void_p()
        engine._check(self.L.mlgpu_events_create(engine.h, int(n_instruments), int(polyphony), ctypes.byref(h)))
        self.h = h
        self.n_instruments, self.polyphony = int(n_instruments), int(polyphony)
        self.V = self.n_instruments * self.polyphony
        engine._children.add(self)
        self.set_sample_rate(sr)

    def close(self):
        if getattr(self, "h", None) and self.engine.h:
            self.L.mlgpu_events_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_sample_rate(self, sr):
        self.engine._check(self.L.mlgpu_events_set_sample_rate(self.h, float(sr)))

    def configure(self, mpe=None, unison=None, mod_cc=None, pitch_bend=None, mpe_pitch_bend=None, glide_seconds=None, drift=None):
        c, L, h = self.engine._check, self.L, self.h
        if mpe is not None:
            c(L.mlgpu_events_set_protocol(h, 1 if mpe else 0))
        if unison is not None:
            c(L.mlgpu_events_set_unison(h, 1 if unison else 0))
        if mod_cc is not None:
            c(L.mlgpu_events_set_mod_cc(h, int(mod_cc)))
        if pitch_bend is not None:
            c(L.mlgpu_events_set_pitch_bend_semitones(h, float(pitch_bend)))
        if mpe_pitch_bend is not None:
            c(L.mlgpu_events_set_mpe_pitch_bend_semitones(h, float(mpe_pitch_bend)))
        if glide_seconds is not None:
            c(L.mlgpu_events_set_pitch_glide_seconds(h, float(glide_seconds)))
        if drift is not None:
            c(L.mlgpu_events_set_drift_amount(h, float(drift)))

    def set_wanted_rows(self, rows):
        """rows: iterable of row indices (Events.ROWS order) that will be asked for; the others are not computed."""
        self.wanted = sorted(set(int(r) for r in rows))
        self.engine._check(self.L.mlgpu_events_set_wanted_rows(self.h, sum(1 << r for r in self.wanted)))

    def add_event(self, instrument, ev):
        self.engine._check(self.L.mlgpu_events_add_event(self.h, int(instrument), ctypes.byref(ev)))

    def add_events(self, instruments, events):
        """A block's events in one call: events[i] (Event) goes to instrument instruments[i]."""
        n = len(events)
        if n == 0:
            return
        inst = np.ascontiguousarray(instruments, np.uint32)
        arr = (Event * n)(*events)
        self.engine._check(self.L.mlgpu_events_add_events(self.h, _np_ptr(inst), ctypes.cast(arr, ctypes.c_void_p), n))

    @staticmethod
    def pack_events(instruments, events):
        """A block's events ready to hand over: (uint32 instrument indices, ctypes array of Event, count) for add_events_packed."""
        n = len(events)
        return np.ascontiguousarray(instruments, np.uint32), (Event * max(1, n))(*events), n

    def add_events_packed(self, packed):
        inst, arr, n = packed
        if n:
            self.engine._check(self.L.mlgpu_events_add_events(self.h, _np_ptr(inst), ctypes.cast(arr, ctypes.c_void_p), n))

    def clear_events(self):
        self.engine._check(self.L.mlgpu_events_clear_events(self.h))

    def newest_voice(self, instrument):
        return self.L.mlgpu_events_newest_voice(self.h, int(instrument))

    def process(self, n_vectors, start_offset, d_outputs, layout=Layout.QUAD):
        """d_outputs: 8 DeviceBuffers (or None) in the order of Events.ROWS."""
        arr = (ctypes.c_void_p * 8)(*[None if b is None else b.ptr for b in d_outputs])
        self.engine._check(self.L.mlgpu_events_process(self.h, int(n_vectors), int(start_offset), arr, int(layout)))

    def reserve_for_graph(self, max_vectors):
        """Setup-time reserve of the control records and side signals a graph bound to this object needs for blocks of up to max_vectors
        DSPVectors (mlgpu_events_reserve_for_graph): from then on Graph.process_events never allocates and refuses longer blocks."""
        self.engine._check(self.L.mlgpu_events_reserve_for_graph(self.h, int(max_vectors)))

    def graph_reserve_bytes(self, max_vectors):
        return int(self.L.mlgpu_events_graph_reserve_bytes(self.h, int(max_vectors)))

    def watch_controllers(self, numbers, max_vectors):
        """The smoothed controller signals (AudioContext::getInputController) to make from now on, one per instrument per number,
        for launches of up to max_vectors DSPVectors; every process call advances them."""
        numbers = [int(n) for n in numbers]
        arr = (ctypes.c_int * max(1, len(numbers)))(*numbers)
        self.engine._check(self.L.mlgpu_events_watch_controllers(self.h, arr, len(numbers), int(max_vectors)))
        self.watched = numbers

    def controller_signal(self, slot):
        """Device pointer of watched controller `slot`'s signal for the last process call: QUAD over n_instruments rows."""
        p = self.L.mlgpu_events_controller_signal(self.h, int(slot))
        if not p:
            raise MlgpuError(Status.ERR_RANGE, "no such watched controller")
        return p

    def controllers_host(self, n_vectors):
        """Test convenience: the watched controllers' signals of the last process call as numpy [slot][instrument][64 T]."""
        N, T = self.n_instruments, int(n_vectors)
        out = []
        for slot in range(len(self.watched)):
            q = np.empty(16 * T * N * 4, np.float32)
            self.engine._check(self.L.mlgpu_download(self.engine.h, _np_ptr(q), self.controller_signal(slot), q.nbytes))
            out.append(q.reshape(16 * T, N, 4).transpose(1, 0, 2).reshape(N, 64 * T))
        return np.stack(out) if out else np.zeros((0, N, 64 * T), np.float32)

    def process_host(self, n_vectors, start_offset=0):
        """Test convenience: returns the 8 rows as numpy [8][V][64 T]."""
        eng, V, T = self.engine, self.V, int(n_vectors)
        wanted = getattr(self, "wanted", list(range(8)))
        bufs = [eng.alloc(4 * V * T * 64) if r in wanted else None for r in range(8)]
        self.process(T, start_offset, bufs, Layout.VOICE_MAJOR)
        return np.stack([np.zeros((V, T * 64), np.float32) if b is None else b.download(np.float32, V * T * 64).reshape(V, T * 64) for b in bufs])


class Transport:
    """AudioContext::ProcessTime (source/app/MLAudioContext.cpp:16-104) for n contexts (mlgpu_transport): the quarter-note phasor
    behind ctx->getBeatPhase(). index None = every context."""
    ALL = (1 << 64) - 1

    def __init__(self, engine, n, max_vectors):
        self.engine, self.L = engine, engine.L
        h = ctypes.c_void_p()
        engine._check(self.L.mlgpu_transport_create(engine.h, int(n), int(max_vectors), ctypes.byref(h)))
        self.h, self.n = h, int(n)
        engine._children.add(self)

    def close(self):
        if getattr(self, "h", None) and self.engine.h:
            self.L.mlgpu_transport_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def update_time(self, ppq_pos, bpm, is_playing, sample_rate, index=None):
        """AudioContext::updateTime: what the host reports for the start of the next process call."""
        self.engine._check(self.L.mlgpu_transport_set_time_and_rate(self.h, self.ALL if index is None else int(index), float(ppq_pos), float(bpm),
                                                                   int(bool(is_playing)), float(sample_rate)))

    def clear(self, index=None):
        self.engine._check(self.L.mlgpu_transport_clear(self.h, self.ALL if index is None else int(index)))

    def process(self, n_vectors):
        self.engine._check(self.L.mlgpu_transport_process(self.h, int(n_vectors)))

    def reserve(self, max_vectors):
        """Another launch length from now on; the phasors go on (the signal pointer changes)."""
        self.engine._check(self.L.mlgpu_transport_reserve(self.h, int(max_vectors)))

    @property
    def beat_phase(self):
        """Device pointer of the last process call's signal: QUAD over the n contexts."""
        return self.L.mlgpu_transport_beat_phase(self.h)

    def samples_since_start(self, index=0):
        return int(self.L.mlgpu_transport_samples_since_start(self.h, int(index)))

    def process_host(self, n_vectors):
        """Test convenience: process, then the signal as numpy [n][64 T]."""
        T = int(n_vectors)
        self.process(T)
        q = np.empty(16 * T * self.n * 4, np.float32)
        self.engine._check(self.L.mlgpu_download(self.engine.h, _np_ptr(q), self.beat_phase, q.nbytes))
        return q.reshape(16 * T, self.n, 4).transpose(1, 0, 2).reshape(self.n, 64 * T)


class PublishedSignal:
    """SignalProcessor::PublishedSignal (source/app/MLSignalProcessor.h:26-105): a decimated frame-major copy of a few
    channels of a few voices, for displays. write() takes device signals (DeviceBuffer) of `n_voices_total` voices."""

    def __init__(self, engine, max_frames, max_voices, channels, octaves_down):
        self.engine, self.L = engine, engine.L
        self.channels = int(channels)
        h = ctypes.c_void_p()
        engine._check(self.L.mlgpu_published_signal_create(engine.h, int(max_frames), int(max_voices), self.channels, int(octaves_down), ctypes.byref(h)))
        self.h = h
        engine._children.add(self)

    def close(self):
        if getattr(self, "h", None) and self.engine.h:
            self.L.mlgpu_published_signal_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def write(self, n_vectors, d_channels, n_voices_total, first_voice=0, n_voices=None, layout=Layout.QUAD):
        arr = (ctypes.c_void_p * self.channels)(*[ctypes.c_void_p(d.ptr) for d in d_channels])
        nv = n_voices_total - first_voice if n_voices is None else n_voices
        self.engine._check(self.L.mlgpu_published_signal_write(self.h, int(n_vectors), arr, int(layout), int(n_voices_total), int(first_voice), int(nv)))

    def read_available(self):
        return int(self.L.mlgpu_published_signal_read_available(self.h))

    def available_frames(self):
        return int(self.L.mlgpu_published_signal_available_frames(self.h))

    def _get(self, fn, frames):
        out = np.zeros(int(frames) * self.channels, np.float32)
        n = fn(self.h, _np_ptr(out), int(frames))
        return out, (int(n) if n is not None else None)

    def read(self, frames):
        return self._get(self.L.mlgpu_published_signal_read, frames)

    def read_latest(self, frames):
        return self._get(self.L.mlgpu_published_signal_read_latest, frames)

    def peek_latest(self, frames):
        return self._get(self.L.mlgpu_published_signal_peek_latest, frames)[0]


class Resampler:
    """Downsampler / Upsampler (MLDSPFilters.h:1316-1473) for V voices: a HalfBandFilter cascade, one stage per octave."""

    def __init__(self, engine, n_voices, octaves, up):
        self.engine, self.L = engine, engine.L
        self.V, self.octaves, self.up = int(n_voices), int(octaves), bool(up)
        h = ctypes.c_void_p()
        engine._check(self.L.mlgpu_resampler_create(engine.h, self.V, self.octaves, 1 if up else 0, ctypes.byref(h)))
        self.h = h
        engine._children.add(self)

    def close(self):
        if getattr(self, "h", None) and self.engine.h:
            self.L.mlgpu_resampler_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clear(self):
        self.engine._check(self.L.mlgpu_resampler_clear(self.h))

    def get_state(self):
        out = np.zeros((self.octaves * 9, self.V), np.float32)
        if out.size:
            self.engine._check(self.L.mlgpu_resampler_get_state(self.h, _np_ptr(out)))
        return out

    def set_state(self, st):
        st = np.ascontiguousarray(st, np.float32)
        if st.size:
            self.engine._check(self.L.mlgpu_resampler_set_state(self.h, _np_ptr(st)))

    def process(self, n_vectors_in, d_in, d_out, in_layout=Layout.QUAD, out_layout=Layout.QUAD):
        self.engine._check(self.L.mlgpu_resampler_process(self.h, int(n_vectors_in), ctypes.c_void_p(d_in.ptr), int(in_layout),
                                                         ctypes.c_void_p(d_out.ptr), int(out_layout)))

    def process_host(self, x, layout=Layout.QUAD):
        """x [V][64*Tin] numpy -> [V][64*Tout] numpy (conversion to / from `layout` on the device)."""
        eng, V = self.engine, self.V
        x = np.ascontiguousarray(x, np.float32)
        Tin = x.shape[1] // 64
        Tout = Tin << self.octaves if self.up else Tin >> self.octaves
        d_vm = eng.to_device(x)
        d_in, d_out = d_vm, eng.alloc(4 * V * 64 * Tout)
        if layout != Layout.VOICE_MAJOR:
            d_in = eng.alloc(x.nbytes)
            eng.layout_convert(d_vm, Layout.VOICE_MAJOR, d_in, layout, V, Tin)
        self.process(Tin, d_in, d_out, layout, layout)
        res = d_out
        if layout != Layout.VOICE_MAJOR:
            res = eng.alloc(4 * V * 64 * Tout)
            eng.layout_convert(d_out, layout, res, Layout.VOICE_MAJOR, V, Tout)
        return res.download(np.float32, V * 64 * Tout).reshape(V, 64 * Tout)


class Bank:
    """Runtime-sized Bank<T,ROWS> (reference MLDSPFunctional.h:321-360): V voices of one chain."""

    def __init__(self, engine, procs, n_voices):
        self.engine = engine
        self.L = engine.L
        self.procs = [int(p) for p in procs]
        self.V = int(n_voices)
        arr = (ctypes.c_int32 * len(self.procs))(*self.procs)
        h = ctypes.c_void_p()
        engine._check(self.L.mlgpu_bank_create(engine.h, arr, len(self.procs), self.V, ctypes.byref(h)))
        self.h = h
        engine._children.add(self)

    def close(self):
        if getattr(self, "h", None) and self.engine.h:
            self.L.mlgpu_bank_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def fused(self):
        return bool(self.L.mlgpu_bank_is_fused(self.h))

    @property
    def kernel_name(self):
        return self.L.mlgpu_bank_kernel_name(self.h).decode()

    def num_coeffs(self, p):
        return self.L.mlgpu_bank_num_coeffs(self.h, p)

    def num_state(self, p):
        return self.L.mlgpu_bank_num_state(self.h, p)

    def clear(self):
        self.engine._check(self.L.mlgpu_bank_clear(self.h))

    def set_coeff(self, proc_idx, coeff_idx, value):
        """value: scalar (broadcast) or per-voice array."""
        if np.isscalar(value):
            self.engine._check(self.L.mlgpu_bank_set_coeff_uniform(self.h, proc_idx, coeff_idx, float(value)))
        else:
            v = np.ascontiguousarray(value, np.float32)
            assert v.shape == (self.V,)
            self.engine._check(self.L.mlgpu_bank_set_coeff(self.h, proc_idx, coeff_idx, _np_ptr(v)))

    def get_coeff(self, proc_idx, coeff_idx):
        out = np.empty(self.V, np.float32)
        self.engine._check(self.L.mlgpu_bank_get_coeff(self.h, proc_idx, coeff_idx, _np_ptr(out)))
        return out

    def set_coeffs(self, proc_idx, coeffs):
        """coeffs: sequence of NC scalars (broadcast) or array [NC][V] — the reference's `coeffs` member."""
        for i, c in enumerate(coeffs):
            self.set_coeff(proc_idx, i, c if np.ndim(c) else float(c))

    def set_all_coeffs(self, coeffs_soa):
        """coeffs_soa: [totalNC][V] in chain order (same layout the CPU checkers take)."""
        row = 0
        for p in range(len(self.procs)):
            for i in range(self.num_coeffs(p)):
                self.set_coeff(p, i, np.ascontiguousarray(coeffs_soa[row]))
                row += 1

    def get_state(self, proc_idx, state_idx):
        out = np.empty(self.V, np.uint32)
        self.engine._check(self.L.mlgpu_bank_get_state(self.h, proc_idx, state_idx, _np_ptr(out)))
        return out

    def set_state(self, proc_idx, state_idx, value):
        if np.isscalar(value):
            self.engine._check(self.L.mlgpu_bank_set_state_uniform(self.h, proc_idx, state_idx, int(value)))
        else:
            v = np.ascontiguousarray(value, np.uint32)
            assert v.shape == (self.V,)
            self.engine._check(self.L.mlgpu_bank_set_state(self.h, proc_idx, state_idx, _np_ptr(v)))

    def get_all_state(self):
        rows = [self.get_state(p, i) for p in range(len(self.procs)) for i in range(self.num_state(p))]
        return np.stack(rows) if rows else np.zeros((0, self.V), np.uint32)

    def set_all_state(self, state_soa):
        row = 0
        for p in range(len(self.procs)):
            for i in range(self.num_state(p)):
                self.set_state(p, i, np.ascontiguousarray(state_soa[row]))
                row += 1

    def set_input_const(self, per_voice):
        v = np.ascontiguousarray(per_voice, np.float32)
        assert v.shape == (self.V,)
        self.engine._check(self.L.mlgpu_bank_set_input_const(self.h, _np_ptr(v)))

    def process(self, n_vectors, d_out, out_layout=Layout.QUAD, d_in=None, in_layout=Layout.QUAD):
        """Enqueue n_vectors DSPVectors for every voice. d_in/d_out: DeviceBuffer or raw int pointer."""
        pin = None if d_in is None else ctypes.c_void_p(d_in.ptr if hasattr(d_in, "ptr") else int(d_in))
        pout = ctypes.c_void_p(d_out.ptr if hasattr(d_out, "ptr") else int(d_out))
        self.engine._check(self.L.mlgpu_bank_process(self.h, int(n_vectors), pin, int(in_layout), pout, int(out_layout)))

    def prepare_mixdown(self):
        """Setup: generate the summing form of this bank's kernel if it has none ahead of time (mlgpu_bank_prepare_mixdown)."""
        self.engine._check(self.L.mlgpu_bank_prepare_mixdown(self.h))

    def process_mixdown(self, n_vectors, d_out, d_in=None, in_layout=Layout.QUAD, d_gains=None):
        """process + Engine.mixdown of its output (no gains) in one call, the voices' signals never written: d_out gets the 64 *
        n_vectors samples of their sum, the same bits as the two calls give (mlgpu_bank_process_mixdown; Status.ERR_UNSUPPORTED for
        banks whose chain has no ahead-of-time summing form until prepare_mixdown() was called)."""
        pin = None if d_in is None else ctypes.c_void_p(d_in.ptr if hasattr(d_in, "ptr") else int(d_in))
        pout = ctypes.c_void_p(d_out.ptr if hasattr(d_out, "ptr") else int(d_out))
        pg = None if d_gains is None else ctypes.c_void_p(d_gains.ptr if hasattr(d_gains, "ptr") else int(d_gains))
        self.engine._check(self.L.mlgpu_bank_process_mixdown(self.h, int(n_vectors), pin, int(in_layout), pg, pout))

    def process_mixdown_shard(self, n_vectors, d_rows, d_in=None, in_layout=Layout.QUAD, d_gains=None):
        """process_mixdown for a bank that is one engine's shard of a larger one: d_rows gets shard_rows(V) rows of 64 * n_vectors
        floats, finished on the host over all shards' rows by mixdown_finish (mlgpu_bank_process_mixdown_shard)."""
        pin = None if d_in is None else ctypes.c_void_p(d_in.ptr if hasattr(d_in, "ptr") else int(d_in))
        pout = ctypes.c_void_p(d_rows.ptr if hasattr(d_rows, "ptr") else int(d_rows))
        pg = None if d_gains is None else ctypes.c_void_p(d_gains.ptr if hasattr(d_gains, "ptr") else int(d_gains))
        self.engine._check(self.L.mlgpu_bank_process_mixdown_shard(self.h, int(n_vectors), pin, int(in_layout), pg, pout))

    def process_host(self, n_vectors, in_signal=None, layout=Layout.QUAD):
        """Test convenience: in_signal/out are VOICE_MAJOR [V][64T] numpy; the kernel runs in `layout`
        (conversion done by the device layout kernel)."""
        eng, V, T = self.engine, self.V, int(n_vectors)
        nbytes = V * T * 64 * 4
        d_in = None
        if in_signal is not None:
            x = np.ascontiguousarray(in_signal, np.float32)
            assert x.shape == (V, 64 * T)
            d_vm = eng.to_device(x)
            if layout == Layout.VOICE_MAJOR:
                d_in = d_vm
            else:
                d_in = eng.alloc(nbytes)
                eng.layout_convert(d_vm, Layout.VOICE_MAJOR, d_in, layout, V, T)
        d_out = eng.alloc(nbytes)
        self.process(T, d_out, layout, d_in, layout)
        if layout == Layout.VOICE_MAJOR:
            res = d_out
        else:
            res = eng.alloc(nbytes)
            eng.layout_convert(d_out, layout, res, Layout.VOICE_MAJOR, V, T)
        return res.download(np.float32, V * T * 64).reshape(V, 64 * T)


class OfflineEngine:
    """Stands in for an Engine where no device is needed: Graph(OfflineEngine(), V, ...).emit() generates a graph's kernel
    source and gfx950 code object with hiprtc (mlgpu_graph_emit). Such a graph cannot be compiled or run."""

    def __init__(self):
        self.L = _lib.load()
        self.h = None
        self._children = set()

    def _check(self, st):
        if st != 0:
            raise MlgpuError(st, "offline graph (no engine): status %d" % st)


class Graph:
    """A run-time defined per-voice DAG of processors and ops, fused into one kernel (mlgpu_graph).

    Nodes are named (after the reference's proc convention, source/procs/MLProcMultiply.cpp:12-18) and
    are added in topological order. `description` form (see patches.py): a list of dicts
      {"name", "type": "input"|"param"|"const"|"proc"|"op", "kind": Proc.X / Op.X, "inputs": [names], "value"}
    """

    def __init__(self, engine, n_voices, description=None, outputs=None, voices_per_lane=0, delay_windows=False, autotune=False,
                 live_constants=False, output_groups=None, input_groups=None, compile_now=True):
        self.engine = engine
        self.L = engine.L
        self.V = int(n_voices)
        h = ctypes.c_void_p()
        engine._check(self.L.mlgpu_graph_create(engine.h, self.V, ctypes.byref(h)))
        self.h = h
        self.ids = {}
        self.inputs, self.outputs, self.controls = [], [], []
        engine._children.add(self)
        if voices_per_lane:
            engine._check(self.L.mlgpu_graph_set_voices_per_lane(self.h, int(voices_per_lane)))
        if delay_windows:   # True / 1: 32-byte sectors behind LDS windows; 2: transposed 64-byte pieces on a wave-uniform clock;
            # 3 / "best": 2 where it applies (at most four rings, not three), else 1
            engine._check(self.L.mlgpu_graph_set_delay_layout(self.h, 3 if delay_windows in (3, "best") else (int(delay_windows) if delay_windows in (2, 4) else 1)))
        if autotune:
            engine._check(self.L.mlgpu_graph_set_autotune(self.h, 1))
        if live_constants:
            self._check(self.L.mlgpu_graph_set_live_constants(self.h, 1))
        if description is not None:
            for n in description:
                self.add(**n)
            for n in description:   # feedback sources may be nodes defined after the feedback node
                if n["type"] == "feedback":
                    self.set_feedback(n["name"], n["source"])
            for o in (outputs or [description[-1]["name"]]):
                self.add_output(o)
            for idx, group in (output_groups or {}).items():   # {output index: voices per group}: set_output_group_sum
                self.set_output_group_sum(idx, group)
            for idx, group in (input_groups or {}).items():    # {input index: voices per row}: set_input_group
                self.set_input_group(idx, group)
            if engine.h is not None and compile_now:
                self.compile()

    def close(self):
        if getattr(self, "h", None) and (self.engine.h or isinstance(self.engine, OfflineEngine)):
            self.L.mlgpu_graph_destroy(self.h)
        self.h = None

    @property
    def device_bytes(self):
        """Device memory the compiled graph owns (mlgpu_graph_device_bytes)."""
        return int(self.L.mlgpu_graph_device_bytes(self.h))

    @property
    def delay_layout(self):
        """The delay-ring layout in effect (mlgpu_graph_delay_layout): 0 rows, 1 sectors, 2 transposed pieces."""
        return self._ret(self.L.mlgpu_graph_delay_layout(self.h), None)

    def workgroups_per_cu(self):
        """Workgroups of this graph's kernel a CU holds at once (mlgpu_graph_workgroups_per_cu)."""
        return self._ret(self.L.mlgpu_graph_workgroups_per_cu(self.h), None)

    def tuning(self):
        """(settled, voices per lane, quads per trip) of the kernel form in use (mlgpu_graph_tuning)."""
        vl, u = ctypes.c_int(), ctypes.c_int()
        r = self.L.mlgpu_graph_tuning(self.h, ctypes.byref(vl), ctypes.byref(u))
        return bool(r > 0), vl.value, u.value

    def emit(self):
        """(kernel source, gfx950 code object bytes) without a device (mlgpu_graph_emit)."""
        code, size = ctypes.c_void_p(), ctypes.c_size_t()
        self._check(self.L.mlgpu_graph_emit(self.h, ctypes.byref(code), ctypes.byref(size)))
        return self.source, ctypes.string_at(code, size.value)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _id(self, ref):
        return self.ids[ref] if isinstance(ref, str) else int(ref)

    # ---- everything by name (the registry: mlgpu_registry_*, source/procs/MLProcMultiply.cpp:12-18,44-47) ----
    def add_named(self, proc_name, node_name, input_names=()):
        """Add a node of registered kind `proc_name` ("multiply", "saw_gen", "lopass", ...) whose inputs are existing nodes, by name."""
        arr = (ctypes.c_char_p * max(1, len(input_names)))(*[n.encode() for n in input_names])
        return self._ret(self.L.mlgpu_graph_add_named(self.h, proc_name.encode(), node_name.encode(), arr, len(input_names)), node_name)

    def set_named_coeff(self, node_name, coeff_name, value):
        """`node.coeffs.<coeff_name> = value` (scalar: every voice; array: per voice)."""
        if np.ndim(value):
            v = np.ascontiguousarray(value, np.float32)
            assert v.shape == (self.V,)
            self._check(self.L.mlgpu_graph_set_named_coeff(self.h, node_name.encode(), coeff_name.encode(), _np_ptr(v), 0.0))
        else:
            self._check(self.L.mlgpu_graph_set_named_coeff(self.h, node_name.encode(), coeff_name.encode(), None, float(value)))

    def set_param_by_name(self, name, value):
        if np.ndim(value):
            v = np.ascontiguousarray(value, np.float32)
            assert v.shape == (self.V,)
            self._check(self.L.mlgpu_graph_set_param_by_name(self.h, name.encode(), _np_ptr(v), 0.0))
        else:
            self._check(self.L.mlgpu_graph_set_param_by_name(self.h, name.encode(), None, float(value)))

    def _ret(self, r, name):
        if r < 0:
            raise MlgpuError(-r, self.L.mlgpu_last_error(self.engine.h).decode() if self.engine.h else "offline graph: status %d" % -r)
        if name:
            self.ids[name] = r
        return r

    def _check(self, st):
        if st != 0:
            raise MlgpuError(st, self.L.mlgpu_last_error(self.engine.h).decode() if self.engine.h else self.L.mlgpu_graph_last_error(self.h).decode())

    def set_const(self, node, value):
        """Change a const node (live_constants=True graphs: between launches, no recompilation)."""
        self._check(self.L.mlgpu_graph_set_const(self.h, self._id(node), float(value)))

    def update_constants_from(self, other):
        """Take the constants of `other`, a graph with the same nodes and wiring (compiled or not)."""
        self._check(self.L.mlgpu_graph_update_constants_from(self.h, other.h))

    def set_feedback(self, feedback_node, value_node):
        self.engine._check(self.L.mlgpu_graph_set_feedback(self.h, self._id(feedback_node), self._id(value_node)))

    def set_max_delay(self, node, max_delay_in_samples):
        self.engine._check(self.L.mlgpu_graph_set_max_delay(self.h, self._id(node), float(max_delay_in_samples)))

    def add(self, name, type, kind=None, inputs=(), value=None, index=0, n_outputs=0, source=None, max_delay=None):
        bname = name.encode() if name else None
        ins = [self._id(i) for i in inputs]
        arr = (ctypes.c_int * max(1, len(ins)))(*ins)
        if type == "input":
            self.inputs.append(name)
            return self._ret(self.L.mlgpu_graph_add_input(self.h, bname), name)
        if type == "control":
            self.controls.append(name)
            return self._ret(self.L.mlgpu_graph_add_control(self.h, bname), name)
        if type == "event_row":   # kind: 0 pitch, 1 gate of the Events object bound with bind_events()
            return self._ret(self.L.mlgpu_graph_add_event_row(self.h, int(kind), bname), name)
        if type == "vop":
            return self._ret(self.L.mlgpu_graph_add_vop(self.h, int(kind), arr, len(ins), bname), name)
        if type == "route":
            return self._ret(self.L.mlgpu_graph_add_route(self.h, int(kind), arr, len(ins), int(index), int(n_outputs), bname), name)
        if type == "param":
            return self._ret(self.L.mlgpu_graph_add_param(self.h, bname), name)
        if type == "const":
            r = self._ret(self.L.mlgpu_graph_add_const(self.h, float(value)), name)
            if name:
                self.L.mlgpu_graph_set_node_name(self.h, r, name.encode())   # so that add_named can wire it by name
            return r
        if type == "const_vector":   # value: 64 floats, the same DSPVector for every voice and vector
            tbl = np.ascontiguousarray(value, np.float32)
            if tbl.shape != (64,):
                raise MlgpuError(Status.ERR_INVALID, "const_vector: 64 floats")
            return self._ret(self.L.mlgpu_graph_add_const_vector(self.h, tbl.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), bname), name)
        if type == "feedback":
            return self._ret(self.L.mlgpu_graph_add_feedback(self.h, bname), name)
        if type == "proc":
            r = self._ret(self.L.mlgpu_graph_add_proc(self.h, int(kind), arr, len(ins), bname), name)
            if max_delay is not None:
                self.set_max_delay(r, max_delay)
            return r
        if type == "op":
            return self._ret(self.L.mlgpu_graph_add_op(self.h, int(kind), arr, len(ins), bname), name)
        raise ValueError(type)

    def begin_region(self, kind, inputs, names=None):
        """Open a rate region (Region.UPSAMPLE_2X / DOWNSAMPLE_2X): returns the nodes that carry `inputs` inside it."""
        ins = [self._id(i) for i in inputs]
        arr = (ctypes.c_int * max(1, len(ins)))(*ins)
        out = (ctypes.c_int * max(1, len(ins)))()
        self.engine._check(self.L.mlgpu_graph_begin_region(self.h, int(kind), arr, len(ins), out))
        ids = [int(out[j]) for j in range(len(ins))]
        for nm, r in zip(names or [], ids):
            self.ids[nm] = r
        return ids

    def end_region(self, result, name=None):
        """Close the open region; returns the outer node carrying fn's resampled result."""
        return self._ret(self.L.mlgpu_graph_end_region(self.h, self._id(result), name.encode() if name else None), name)

    def add_output(self, node):
        self.outputs.append(node)
        self.engine._check(self.L.mlgpu_graph_add_output(self.h, self._id(node)))

    def set_output_group_sum(self, output_index, group):
        """Output `output_index` becomes the in-order sum of groups of `group` adjacent voices (V / group channels)."""
        self._check(self.L.mlgpu_graph_set_output_group_sum(self.h, int(output_index), int(group)))

    def set_output_mixdown(self, output_index, on=True):
        """Output `output_index` becomes one channel: the mixdown of all voices, mlgpu_mixdown's bits, made inside the voice kernel
        (mlgpu_graph_set_output_mixdown). process() then wants 64 * n_vectors floats for it. on="shard": this graph is one engine's part of
        a larger bank - the output takes mixdown_shard_rows(V) rows of 64 * n_vectors floats for mixdown_finish."""
        self._check(self.L.mlgpu_graph_set_output_mixdown(self.h, int(output_index), 2 if on == "shard" else (1 if on else 0)))

    def reserve_mixdown(self, max_vectors):
        """Setup: the engine's mixdown scratch for this graph's mixed-down outputs (mlgpu_graph_reserve_mixdown)."""
        self._check(self.L.mlgpu_graph_reserve_mixdown(self.h, int(max_vectors)))

    def set_input_group(self, input_index, group):
        """Input `input_index` is a signal of V / group rows: voice v reads row v // group (one controller or transport signal per
        instrument of `group` voices)."""
        self._check(self.L.mlgpu_graph_set_input_group(self.h, int(input_index), int(group)))

    def compile(self):
        self.engine._check(self.L.mlgpu_graph_compile(self.h))

    def compile_async(self):
        """Start the compile on a thread of the library's (mlgpu_graph_compile_async); poll with compile_poll(). On a graph of an
        OfflineEngine: an ahead-of-time compile that fills the memory and disk caches for this description."""
        self.engine._check(self.L.mlgpu_graph_compile_async(self.h))

    def compile_poll(self):
        """False while the compile is in flight; True once the graph is ready; raises MlgpuError when the compile failed."""
        st = self.L.mlgpu_graph_compile_poll(self.h)
        if st == BUSY:
            return False
        self.engine._check(st)
        return True

    @property
    def source(self):
        return self.L.mlgpu_graph_source(self.h).decode()

    def clear(self):
        self.engine._check(self.L.mlgpu_graph_clear(self.h))

    def set_param(self, node, value):
        if np.isscalar(value):
            self.engine._check(self.L.mlgpu_graph_set_param_uniform(self.h, self._id(node), float(value)))
        else:
            v = np.ascontiguousarray(value, np.float32)
            assert v.shape == (self.V,)
            self.engine._check(self.L.mlgpu_graph_set_param(self.h, self._id(node), _np_ptr(v)))

    def set_coeff(self, node, idx, value):
        if np.isscalar(value):
            self.engine._check(self.L.mlgpu_graph_set_coeff_uniform(self.h, self._id(node), idx, float(value)))
        else:
            v = np.ascontiguousarray(value, np.float32)
            assert v.shape == (self.V,)
            self.engine._check(self.L.mlgpu_graph_set_coeff(self.h, self._id(node), idx, _np_ptr(v)))

    def set_coeffs(self, node, coeffs):
        for i, c in enumerate(coeffs):
            self.set_coeff(node, i, c if np.ndim(c) else float(c))

    def num_state(self, node):
        return self.L.mlgpu_graph_num_state(self.h, self._id(node))

    def get_state(self, node, idx):
        out = np.empty(self.V, np.uint32)
        self.engine._check(self.L.mlgpu_graph_get_state(self.h, self._id(node), idx, _np_ptr(out)))
        return out

    def set_state(self, node, idx, value):
        v = np.ascontiguousarray(np.broadcast_to(np.asarray(value, np.uint32), (self.V,)))
        self.engine._check(self.L.mlgpu_graph_set_state(self.h, self._id(node), idx, _np_ptr(v)))

    def set_input_layout(self, input_index, layout):
        self.engine._check(self.L.mlgpu_graph_set_input_layout(self.h, int(input_index), int(layout)))

    def process(self, n_vectors, d_inputs, d_outputs, in_layout=Layout.QUAD, out_layout=Layout.QUAD, d_controls=()):
        """d_inputs / d_outputs / d_controls: lists of DeviceBuffer in the order inputs / outputs / controls were
        added; a control buffer holds [n_vectors][V] floats."""
        raw = lambda b: b.ptr if hasattr(b, "ptr") else int(b)  # noqa: E731
        pi = (ctypes.c_void_p * max(1, len(d_inputs)))(*[raw(b) for b in d_inputs])
        po = (ctypes.c_void_p * max(1, len(d_outputs)))(*[raw(b) for b in d_outputs])
        pc = (ctypes.c_void_p * max(1, len(d_controls)))(*[raw(b) for b in d_controls])
        self.engine._check(self.L.mlgpu_graph_process_ctl(self.h, int(n_vectors), pi, int(in_layout), pc, po, int(out_layout)))

    def bind_events(self, events):
        """The Events object whose pitch / gate rows the graph's event_row nodes compute (MIDI protocol, same number of voices)."""
        self.engine._check(self.L.mlgpu_graph_bind_events(self.h, events.h))
        self._events = events

    def process_events(self, n_vectors, start_offset, d_inputs, d_outputs, in_layout=Layout.QUAD, out_layout=Layout.QUAD, d_controls=()):
        """process() for a graph with event rows: the block's events (Events.add_events) starting at frame start_offset."""
        raw = lambda b: b.ptr if hasattr(b, "ptr") else int(b)  # noqa: E731
        pi = (ctypes.c_void_p * max(1, len(d_inputs)))(*[raw(b) for b in d_inputs])
        po = (ctypes.c_void_p * max(1, len(d_outputs)))(*[raw(b) for b in d_outputs])
        pc = (ctypes.c_void_p * max(1, len(d_controls)))(*[raw(b) for b in d_controls])
        self.engine._check(self.L.mlgpu_graph_process_events(self.h, int(n_vectors), int(start_offset), pi, int(in_layout), pc, po, int(out_layout)))

    def process_host(self, n_vectors, in_signals, layout=Layout.QUAD):
        """Test convenience: VOICE_MAJOR numpy in ({name: [V][64T]}) -> list of VOICE_MAJOR numpy outs."""
        eng, V, T = self.engine, self.V, int(n_vectors)
        nbytes = V * T * 64 * 4
        d_in = []
        for name in self.inputs:
            d_vm = eng.to_device(np.ascontiguousarray(in_signals[name], np.float32))
            if layout == Layout.VOICE_MAJOR:
                d_in.append(d_vm)
            else:
                d = eng.alloc(nbytes)
                eng.layout_convert(d_vm, Layout.VOICE_MAJOR, d, layout, V, T)
                d_in.append(d)
        # controls: {name: [V][T]} on the host -> [T][V] on the device
        d_ctl = [eng.to_device(np.ascontiguousarray(np.asarray(in_signals[name], np.float32).reshape(V, T).T))
                 for name in self.controls]
        d_out = [eng.alloc(nbytes) for _ in self.outputs]
        self.process(T, d_in, d_out, layout, layout, d_ctl)
        res = []
        for d in d_out:
            if layout != Layout.VOICE_MAJOR:
                r = eng.alloc(nbytes)
                eng.layout_convert(d, layout, r, Layout.VOICE_MAJOR, V, T)
                d = r
            res.append(d.download(np.float32, V * T * 64).reshape(V, 64 * T))
        return res


def jit_selftest():
    """Device-free: do the run-time generated chain / graph kernels compile for gfx950?"""
    L = _lib.load()
    buf = ctypes.create_string_buffer(1 << 16)
    st = L.mlgpu_jit_selftest(buf, len(buf))
    return st, buf.value.decode()


# ---- coefficient makers: host libm through the C-ABI (reference: static T::makeCoeffs) ----

def _mk(name, nout, *params):
    L = _lib.load()
    out = (ctypes.c_float * nout)()
    getattr(L, name)(*[float(p) for p in params], out)
    return np.array(out[:], np.float32)


class Lopass:
    @staticmethod
    def makeCoeffs(omega, k):
        return _mk("mlgpu_lopass_make_coeffs", 3, omega, k)


class Hipass:
    @staticmethod
    def makeCoeffs(omega, k):
        return _mk("mlgpu_hipass_make_coeffs", 4, omega, k)


class Bandpass:
    @staticmethod
    def makeCoeffs(omega, k):
        return _mk("mlgpu_bandpass_make_coeffs", 3, omega, k)


class LoShelf:
    @staticmethod
    def makeCoeffs(omega, k, A):
        return _mk("mlgpu_loshelf_make_coeffs", 5, omega, k, A)


class HiShelf:
    @staticmethod
    def makeCoeffs(omega, k, A):
        return _mk("mlgpu_hishelf_make_coeffs", 6, omega, k, A)


class Bell:
    @staticmethod
    def makeCoeffs(omega, k, A):
        return _mk("mlgpu_bell_make_coeffs", 4, omega, k, A)


class OnePole:
    @staticmethod
    def makeCoeffs(omega):
        return _mk("mlgpu_onepole_make_coeffs", 2, omega)


class DCBlocker:
    @staticmethod
    def makeCoeffs(omega):
        return np.float32(_lib.load().mlgpu_dcblocker_make_coeffs(float(omega)))


class ADSR:
    @staticmethod
    def calcCoeffs(a, d, s, r, sr):
        return _mk("mlgpu_adsr_calc_coeffs", 4, a, d, s, r, sr)


class Allpass1:
    @staticmethod
    def makeCoeffs(d):
        return np.float32(_lib.load().mlgpu_allpass1_make_coeffs(float(d)))


class FractionalDelay:
    @staticmethod
    def makeState(delay_in_samples):
        """setDelayInSamples -> (delayInt as int32 bits, allpass coefficient): state words 3 and 4"""
        return _mk("mlgpu_fractional_delay_make_state", 2, delay_in_samples)


class LinearGlide:
    @staticmethod
    def makeCoeffs(glide_time_in_samples):
        """setGlideTimeInSamples -> C{vectorsPerGlide as int32 bits, dyPerVector}"""
        return _mk("mlgpu_linear_glide_make_coeffs", 2, glide_time_in_samples)


class SampleAccurateLinearGlide:
    @staticmethod
    def makeCoeffs(glide_time_in_samples):
        return _mk("mlgpu_sample_accurate_linear_glide_make_coeffs", 2, glide_time_in_samples)


def dBToGain(dB):
    return np.float32(_lib.load().mlgpu_db_to_gain(float(dB)))
