"""Loader for the committed golden vectors (tests/golden/*.npz, made by make_golden.py
from the compiled reference) and the SURVEY Appendix-B anchors."""
import os

import numpy as np

from madronalib_amd.constants import Op, Proc

HERE = os.path.dirname(os.path.abspath(__file__))


def load_ops():
    return np.load(os.path.join(HERE, "golden", "ops.npz"))


def load_chains():
    return np.load(os.path.join(HERE, "golden", "chains.npz"))


def load_multi():
    return np.load(os.path.join(HERE, "golden", "multi.npz"))


def multi_golden_case(d, name):
    """A multi-input / vector-rate case of multi.npz: inputs [(rate, array)] as inputs.multi_case returns them."""
    rates = d[name + "_rates"]
    ins = [("control" if rates[i] else "audio", d[f"{name}_in{i}"]) for i in range(len(rates))]
    return dict(kind=int(d[name + "_kind"]), coeffs=d[name + "_coeffs"], state0=d[name + "_state0"], inputs=ins,
                out=[d[name + "_out1"], d[name + "_out2"]], state=[d[name + "_state1"], d[name + "_state2"]])


def load_rows():
    return np.load(os.path.join(HERE, "golden", "rows.npz"))


def rows_golden_case(d, name):
    ins = []
    while f"{name}|in{len(ins)}" in d.files:
        ins.append(d[f"{name}|in{len(ins)}"])
    return ins, d[f"{name}|out"]


def chain_case_names(d=None):
    d = d or load_chains()
    return sorted(k[:-6] for k in d.files if k.endswith("_procs"))


def chain_case(d, name):
    g = lambda k: d[name + k] if (name + k) in d.files else None  # noqa: E731
    return dict(procs=[int(x) for x in d[name + "_procs"]], coeffs=d[name + "_coeffs"], state0=d[name + "_state0"],
                in_signal=g("_in_signal"), in_const=g("_in_const"), out1=d[name + "_out1"],
                state1=d[name + "_state1"], out2=d[name + "_out2"], state2=d[name + "_state2"])


def hexf(s):
    return np.float32(float.fromhex(s))


# produced by the reference built with the pinned flags.
ANCHORS = dict(
    cfg1=dict(procs=[Proc.SINE_GEN, Proc.LOPASS], freq=220.0 / 48000.0, lopass=(0.1, 1.0),
              y0_3=[hexf(x) for x in ("-0x1.09fe14p-9", "-0x1.5d1a5cp-7", "-0x1.d1fd4ap-6", "-0x1.bb038p-5")],
              y63=hexf("-0x1.f1279ep-1")),
    cfg3=dict(procs=[Proc.SAW_GEN, Proc.BANDPASS, Proc.GAIN], freq=440.0 / 48000.0, bandpass=(0.05, 0.5), gain=0.25,
              z0_3=[hexf(x) for x in ("-0x1.205afcp-5", "-0x1.8c0f74p-4", "-0x1.1d388ep-3", "-0x1.4b4bd8p-3")],
              z63=hexf("0x1.b99136p-7")),
    saw_vec3=dict(w0=0xbf228f5e, w63=0x3f051eb6),
    noise_vec0=dict(n0=0xbf07221c, n63=0xbdb5b380),
    onepole=dict(omega=0.15, a0=0x3f1c3f2c, b1=0x3ec781a9, o1=0x3e73887c, o63=0x14458c96),
    cfg4_vec4=dict(y0=0xbbb12288, y31=0xbe529af7, y63=0x3d836700),
    ramp_elem5={Op.SIN_APPROX: 0xbef4de79, Op.EXP_APPROX: 0x3d91b894, Op.SIN: 0xbef4ddb4, Op.EXP: 0x3d91b880},
    ramp_hash={Op.SIN_APPROX: 0x0d876596, Op.COS_APPROX: 0x8062bd7c, Op.EXP_APPROX: 0x8476f9e5,
               Op.SIN: 0xf5029f27, Op.COS: 0xa141ae3d, Op.EXP: 0x63ce3866},
    ramp_hash_log={Op.LOG_APPROX: 0x9efffdac, Op.LOG: 0x232e4d02},  # input a*a + 0.1
)


def hash31(bits):
    h = 0
    for b in np.asarray(bits, np.uint32).ravel():
        h = (h * 31 + int(b)) & 0xFFFFFFFF
    return h
