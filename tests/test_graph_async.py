"""mlgpu_graph_compile_async / mlgpu_graph_compile_poll without a device: a graph created without an engine compiles AHEAD OF TIME
(code generation + hiprtc for gfx950 on a thread of the library's, result into the memory and disk caches); while the job is in
flight the graph answers MLGPU_ERR_BUSY to everything else. The live patch swap on a device is tests/test_gpu_graph.py."""
import os
import time

import numpy as np
import pytest

import madronalib_amd as ml
from madronalib_amd import _lib, patches
from madronalib_amd.constants import Op


def _fresh_description():
    """The synth16 voice with one more node whose constant no earlier run has seen: a cold compile (seconds), whatever the caches hold."""
    desc, outs = patches.synth16()
    salt = float(np.float32(1.0 + (time.time_ns() % 1000003) * 1e-7))
    desc = desc + [dict(name="salt", type="const", value=salt), dict(name="salted", type="op", kind=Op.MULTIPLY, inputs=[outs[0], "salt"])]
    return desc, ["salted"]


def test_status_and_revision_are_exported():
    L = _lib.load()
    assert L.mlgpu_status_string(ml.BUSY).decode().startswith("busy")
    assert ml.behaviour_revision() >= 3


def test_ahead_of_time_compile_is_busy_while_in_flight_and_fills_the_cache():
    L = _lib.load()
    desc, outs = _fresh_description()
    g = ml.Graph(ml.OfflineEngine(), 4096, desc, outs)
    before = ml.jit_stats()
    t0 = time.perf_counter()
    g.compile_async()
    started = time.perf_counter() - t0
    assert started < 0.25, f"compile_async returned after {started:.3f} s: it must not do the compile itself"
    # the graph is the job's now
    assert L.mlgpu_graph_compile(g.h) == ml.BUSY
    assert L.mlgpu_graph_compile_async(g.h) == ml.BUSY
    assert L.mlgpu_graph_emit(g.h, None, None) == ml.BUSY
    assert L.mlgpu_graph_add_output(g.h, 0) == ml.BUSY
    # ... every mutator, also the ones that are legal after a compile (round 6: the worker reads names and constants while it
    # generates code), and the strings it writes are not handed out meanwhile
    import ctypes
    assert L.mlgpu_graph_set_node_name(g.h, 0, b"renamed") == ml.BUSY
    const_node = L.mlgpu_graph_node(g.h, b"salt")
    assert const_node == -ml.BUSY
    assert L.mlgpu_graph_set_const(g.h, 0, ctypes.c_float(2.0)) == ml.BUSY
    assert L.mlgpu_graph_set_live_constants(g.h, 1) == ml.BUSY
    assert L.mlgpu_graph_set_input_layout(g.h, 0, 0) == ml.BUSY
    assert L.mlgpu_graph_clear(g.h) == ml.BUSY
    assert L.mlgpu_graph_delay_layout(g.h) == -ml.BUSY
    assert L.mlgpu_graph_source(g.h) == b"" and L.mlgpu_graph_last_error(g.h) == b""
    busy_polls = 0
    while not g.compile_poll():
        busy_polls += 1
        assert time.perf_counter() - t0 < 120
        time.sleep(0.01)
    cold = time.perf_counter() - t0
    assert busy_polls > 0, "the compile finished before the first poll: not a cold compile?"
    after = ml.jit_stats()
    assert after["compiles"] == before["compiles"] + 1
    # the code exists now: emit answers from memory, and a second graph of the same description does not run hiprtc again
    src, code = g.emit()
    assert len(code) > 1000 and "mlgpu_graph_kernel" in src
    # an ahead-of-time graph stays "done": every later poll (and a second compile_async) answers OK (round 5 said INVALID from the second poll on)
    assert L.mlgpu_graph_compile_poll(g.h) == 0 and L.mlgpu_graph_compile_poll(g.h) == 0
    assert L.mlgpu_graph_compile_async(g.h) == 0 and L.mlgpu_graph_compile_poll(g.h) == 0
    g2 = ml.Graph(ml.OfflineEngine(), 4096, desc, outs)
    t1 = time.perf_counter()
    g2.compile_async()
    while not g2.compile_poll():
        time.sleep(0.001)
    warm = time.perf_counter() - t1
    assert ml.jit_stats()["compiles"] == after["compiles"], (cold, warm)   # the second graph found the first one's code: hiprtc did not run again
    assert after["diskWrites"] >= before["diskWrites"] if "diskWrites" in after else True   # (the disk cache itself: tests/test_registry.py / test_abi.py)
    g.close()
    g2.close()


def test_a_failed_compile_is_reported_by_poll_and_destroy_waits_for_a_job():
    L = _lib.load()
    g = ml.Graph(ml.OfflineEngine(), 64)
    g.add("x", "input")
    g.compile_async()               # no outputs: the job fails at once
    while True:
        st = L.mlgpu_graph_compile_poll(g.h)
        if st != ml.BUSY:
            break
        time.sleep(0.001)
    assert st == 1 and "no outputs" in L.mlgpu_graph_last_error(g.h).decode()
    assert L.mlgpu_graph_compile_poll(g.h) == 1       # nothing in flight any more: invalid
    g.close()
    desc, outs = _fresh_description()
    g = ml.Graph(ml.OfflineEngine(), 4096, desc, outs)
    g.compile_async()
    g.close()                       # joins the job; must neither crash nor leak the thread
