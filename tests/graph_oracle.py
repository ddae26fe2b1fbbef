"""CPU evaluation of a graph description with the oracle (TEST INFRASTRUCTURE): node by node over the
whole signal. Valid because every node is causal and there are no feedback edges, so per-node
evaluation over the full stream equals per-sample evaluation in topological order."""
import ctypes

import numpy as np

from madronalib_amd.constants import Proc


def _pulse2(chk, V, T, omega32, freq, width):
    fn = getattr(chk.lib, chk.prefix + "pulse2_process")
    fn.restype = ctypes.c_int
    f32p, u32p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_uint32)
    fn.argtypes = [ctypes.c_size_t, ctypes.c_size_t, u32p, f32p, f32p, f32p]
    out = np.empty((V, 64 * T), np.float32)
    freq = np.ascontiguousarray(freq, np.float32)
    width = np.ascontiguousarray(width, np.float32)
    assert fn(V, T, omega32.ctypes.data_as(u32p), freq.ctypes.data_as(f32p), width.ctypes.data_as(f32p),
              out.ctypes.data_as(f32p)) == 0
    return out


def evaluate(chk, description, outputs, V, T, in_signals, params, coeffs, states):
    """chk: Oracle or Ref. params {name: [V] or scalar}, coeffs {name: [NC][V]}, states {name: [NS][V] uint32,
    updated in place}. Returns list of output signals [V][64T]."""
    S = 64 * T
    val = {}
    for n in description:
        name, ty = n["name"], n["type"]
        ins = [val[i] for i in n.get("inputs", [])]
        if ty == "input":
            val[name] = np.ascontiguousarray(in_signals[name], np.float32)
        elif ty == "param":
            p = np.broadcast_to(np.asarray(params[name], np.float32), (V,))
            val[name] = np.ascontiguousarray(np.repeat(p[:, None], S, 1))
        elif ty == "control":
            c = np.asarray(in_signals[name], np.float32).reshape(V, T)
            val[name] = np.ascontiguousarray(np.repeat(c, 64, 1))
        elif ty == "vop":
            a = [np.ascontiguousarray(x) for x in ins] + [None, None]
            val[name] = chk.vop(n["kind"], V, T, a[0], a[1])
        elif ty == "const":
            val[name] = np.full((V, S), np.float32(n["value"]), np.float32)
        elif ty == "const_vector":
            val[name] = np.ascontiguousarray(np.tile(np.asarray(n["value"], np.float32), (V, T)))
        elif ty == "op":
            a = [np.ascontiguousarray(x) for x in ins] + [None, None]
            val[name] = chk.op(n["kind"], a[0], a[1], a[2]).view(np.float32).reshape(V, S)
        elif ty == "proc":
            kind = n["kind"]
            st = states[name]
            co = coeffs.get(name, np.zeros((chk.num_coeffs(kind), V), np.float32))
            if len(ins) > 1 or kind in Proc.VECTOR_RATE:
                val[name] = chk.proc_multi(kind, T, co, st, ins)
            else:
                sig = ins[0] if ins else None
                val[name] = chk.chain_process([kind], T, np.ascontiguousarray(co, np.float32), st, sig, None)
        else:
            raise ValueError(ty)
    return [val[o] for o in outputs]


def ring_len(max_delay):
    """IntegerDelay::setMaxDelayInSamples (MLDSPFilters.h:823-831): 2^bitsToContain(floor(d) + 64)"""
    n, bits = int(np.floor(max_delay)) + 64, 0
    while (1 << bits) < n:
        bits += 1
    return 1 << bits


def new_stream_state(chk, description, V):
    """Per-node state for evaluate_stream: processor state words, delay rings, feedback vectors."""
    st = {}
    for n in description:
        if n["type"] == "proc":
            st[n["name"]] = chk.chain_default_state([n["kind"]], V)
            if n["kind"] in Proc.DELAYS:
                rings = 2 if n["kind"] == Proc.PITCHBENDABLE_DELAY else 1
                st[n["name"] + "/mem"] = np.zeros((V, rings, ring_len(n["max_delay"])), np.float32)
        elif n["type"] == "feedback":
            st[n["name"]] = np.zeros((V, 64), np.float32)
    return st


def evaluate_stream(chk, description, outputs, V, T, in_signals, params, coeffs, st):
    """Vector-by-vector evaluation (graphs with delay lines and one-vector feedback): every DSPVector is evaluated
    node by node; feedback nodes hand the previous vector's value of their source to this one."""
    outs = [np.empty((V, 64 * T), np.float32) for _ in outputs]
    for t in range(T):
        sl = slice(64 * t, 64 * (t + 1))
        val = {}
        for n in description:
            name, ty = n["name"], n["type"]
            ins = [val[i] for i in n.get("inputs", [])]
            if ty == "input":
                val[name] = np.ascontiguousarray(in_signals[name][:, sl], np.float32)
            elif ty == "control":
                c = np.asarray(in_signals[name], np.float32).reshape(V, T)[:, t]
                val[name] = np.ascontiguousarray(np.repeat(c[:, None], 64, 1))
            elif ty == "param":
                p = np.broadcast_to(np.asarray(params[name], np.float32), (V,))
                val[name] = np.ascontiguousarray(np.repeat(p[:, None], 64, 1))
            elif ty == "const":
                val[name] = np.full((V, 64), np.float32(n["value"]), np.float32)
            elif ty == "const_vector":
                val[name] = np.ascontiguousarray(np.tile(np.asarray(n["value"], np.float32), (V, 1)))
            elif ty == "feedback":
                val[name] = st[name].copy()
            elif ty == "vop":
                a = [np.ascontiguousarray(x) for x in ins] + [None, None]
                val[name] = chk.vop(n["kind"], V, 1, a[0], a[1])
            elif ty == "op":
                a = [np.ascontiguousarray(x) for x in ins] + [None, None]
                val[name] = chk.op(n["kind"], a[0], a[1], a[2]).view(np.float32).reshape(V, 64)
            elif ty == "proc":
                kind = n["kind"]
                co = coeffs.get(name, np.zeros((chk.num_coeffs(kind), V), np.float32))
                if kind in Proc.DELAYS:
                    val[name] = chk.delay_process(kind, 1, st[name], st[name + "/mem"], ins)
                elif len(ins) > 1 or kind in Proc.VECTOR_RATE:
                    val[name] = chk.proc_multi(kind, 1, co, st[name], ins)
                else:
                    val[name] = chk.chain_process([kind], 1, np.ascontiguousarray(co, np.float32), st[name], ins[0] if ins else None, None)
            else:
                raise ValueError(ty)
        for n in description:
            if n["type"] == "feedback":
                st[n["name"]] = np.ascontiguousarray(val[n["source"]], np.float32).copy()
        for o, name in zip(outs, outputs):
            o[:, sl] = val[name]
    return outs
