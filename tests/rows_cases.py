"""Row-plumbing / routing cases: each name is a template instantiation the compiled reference exposes
(oracle/ref_wrapper.cpp: mlref_rows_case); `run(api, name, inputs)` evaluates the same function through the
rule-based calls (madronalib_amd/rows.py) on any backend (CPU oracle or the GPU engine)."""
import numpy as np

from madronalib_amd import rows as R

# name -> row counts of the inputs
ROWS_CASES = {
    "repeatRows<3>(2)": [2], "stretchRows<7>(3)": [3], "stretchRows<4>(6)": [6], "zeroPadRows<5>(3)": [3], "zeroPadRows<2>(3)": [3],
    "shiftRows<5>(+2)": [5], "shiftRows<5>(-1)": [5], "rotateRows<5>(+2)": [5], "rotateRows<5>(-7)": [5],
    "concatRows(2,3)": [2, 3], "concatRows(1,2,3)": [1, 2, 3], "concatRows(1,2,3,1)": [1, 2, 3, 1],
    "rotateLeft<3>": [3], "rotateRight<3>": [3], "shuffleRows(2,4)": [2, 4], "shuffleRows(4,1)": [4, 1],
    "evenRows<5>": [5], "oddRows<5>": [5], "separateRows<1,4>(6)": [6], "addRows<5>": [5], "rowIndex<4>": [],
    "columnIndex<3>": [], "normalize<3>": [3],
    # routing: selector (1 row) + 2-row signals; mix: gains (3 rows) + three 2-row signals
    "multiplex(3)x2": [1, 2, 2, 2], "multiplexLinear(3)x2": [1, 2, 2, 2], "demultiplex(3)x2": [1, 2], "demultiplexLinear(3)x2": [1, 2],
    "mix(3)x2": [3, 2, 2, 2],
}


def case_inputs(name, seed=0):
    rng = np.random.default_rng(seed + sum(map(ord, name)))
    ins = [(rng.standard_normal((r, 64)) * 3).astype(np.float32) for r in ROWS_CASES[name]]
    if "multiplex" in name:   # selector in [0, 1.999]: the integer part is dropped by s - truncf(s)
        ins[0] = rng.uniform(0.0, 1.999, (1, 64)).astype(np.float32)
        ins[0][0, :4] = [0.0, 0.999999, 1.0, 0.5]
    if name.startswith("addRows"):
        ins[0][0, :3] = -0.0      # 0 + (-0) = +0
    return ins


def run(api, name, ins):
    x = ins
    f = {
        "repeatRows<3>(2)": lambda: R.repeatRows(api, x[0], 3), "stretchRows<7>(3)": lambda: R.stretchRows(api, x[0], 7),
        "stretchRows<4>(6)": lambda: R.stretchRows(api, x[0], 4), "zeroPadRows<5>(3)": lambda: R.zeroPadRows(api, x[0], 5),
        "zeroPadRows<2>(3)": lambda: R.zeroPadRows(api, x[0], 2), "shiftRows<5>(+2)": lambda: R.shiftRows(api, x[0], 2),
        "shiftRows<5>(-1)": lambda: R.shiftRows(api, x[0], -1), "rotateRows<5>(+2)": lambda: R.rotateRows(api, x[0], 2),
        "rotateRows<5>(-7)": lambda: R.rotateRows(api, x[0], -7), "concatRows(2,3)": lambda: R.concatRows(api, *x),
        "concatRows(1,2,3)": lambda: R.concatRows(api, *x), "concatRows(1,2,3,1)": lambda: R.concatRows(api, *x),
        "rotateLeft<3>": lambda: R.rotateLeft(api, x[0]), "rotateRight<3>": lambda: R.rotateRight(api, x[0]),
        "shuffleRows(2,4)": lambda: R.shuffleRows(api, *x), "shuffleRows(4,1)": lambda: R.shuffleRows(api, *x),
        "evenRows<5>": lambda: R.evenRows(api, x[0]), "oddRows<5>": lambda: R.oddRows(api, x[0]),
        "separateRows<1,4>(6)": lambda: R.separateRows(api, x[0], 1, 4), "addRows<5>": lambda: R.addRows(api, x[0]),
        "rowIndex<4>": lambda: R.rowIndex(api, 4),
        "columnIndex<3>": lambda: R.repeatRows(api, np.arange(64, dtype=np.float32)[None, :], 3),
        "normalize<3>": lambda: R.normalize(api, x[0]),
        "multiplex(3)x2": lambda: api.multiplex(x[0], x[1:], False), "multiplexLinear(3)x2": lambda: api.multiplex(x[0], x[1:], True),
        "demultiplex(3)x2": lambda: np.concatenate(api.demultiplex(x[0], x[1], 3, False), 0),
        "demultiplexLinear(3)x2": lambda: np.concatenate(api.demultiplex(x[0], x[1], 3, True), 0),
        "mix(3)x2": lambda: R.mix(api, x[0], *x[1:]),
    }[name]
    return np.asarray(f(), np.float32).reshape(-1, 64)
