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
