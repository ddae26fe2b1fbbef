"""The reference's row-plumbing and routing functions (MLDSPOps.h:1041-1383, MLDSPRouting.h:59-234) expressed as
the rule-based calls of the C-ABI (mlgpu_rows_map & co, include/mlgpu.h). The reference's templates carry row
counts at compile time; here arrays are numpy [rows][64] and the counts are read from the shapes.

`api` is anything with the Engine's host-convenience methods rows_map / rows_add / rows_normalize / rows_index /
multiplex / demultiplex (madronalib_amd.Engine on the GPU; the tests pass a CPU checker with the same methods)."""
import numpy as np

from .constants import RowsRule


def _rows(x):
    return np.ascontiguousarray(x, np.float32).reshape(-1, 64)


def repeatRows(api, x, R):       # MLDSPOps.h:1057-1068
    x = _rows(x)
    n = x.shape[0]
    return api.rows_map(RowsRule.REPEAT, 0, 0, 0, x, n, R * n, 0, 1, R * n, 1)


def stretchRows(api, x, R):      # :1073-1083
    x = _rows(x)
    return api.rows_map(RowsRule.STRETCH, 0, 0, 0, x, x.shape[0], R, 0, 1, R, 1)


def zeroPadRows(api, x, R):      # :1088-1099
    x = _rows(x)
    return api.rows_map(RowsRule.SHIFT, 0, 0, 0, x, x.shape[0], R, 0, 1, R, 1)


def shiftRows(api, x, k):        # :1103-1121
    x = _rows(x)
    n = x.shape[0]
    return api.rows_map(RowsRule.SHIFT, k, 0, 0, x, n, n, 0, 1, n, 1)


def rotateRows(api, x, k):       # :1126-1139
    x = _rows(x)
    n = x.shape[0]
    return api.rows_map(RowsRule.ROTATE, k, 0, 0, x, n, n, 0, 1, n, 1)


def concatRows(api, *xs):        # :1145-1215
    xs = [_rows(x) for x in xs]
    total = sum(x.shape[0] for x in xs)
    dst, off = np.zeros((total, 64), np.float32), 0
    for x in xs:
        dst = api.rows_map(RowsRule.STRIDED, 0, 1, 0, x, x.shape[0], total, off, 1, x.shape[0], 1, dst)
        off += x.shape[0]
    return dst


def rotateLeft(api, x):          # :1219-1245
    x = _rows(x)
    n = x.shape[0]
    return api.rows_map(RowsRule.STRIDED, 0, 1, +1, x, n, n, 0, 1, n, 1)


def rotateRight(api, x):         # :1249-1276
    x = _rows(x)
    n = x.shape[0]
    return api.rows_map(RowsRule.STRIDED, 0, 1, -1, x, n, n, 0, 1, n, 1)


def shuffleRows(api, a, b):      # :1281-1305: interleave, then append the excess of the longer one
    a, b = _rows(a), _rows(b)
    na, nb = a.shape[0], b.shape[0]
    m, total = min(na, nb), na + nb
    dst = np.zeros((total, 64), np.float32)
    dst = api.rows_map(RowsRule.STRIDED, 0, 1, 0, a, na, total, 0, 2, m, 1, dst)
    dst = api.rows_map(RowsRule.STRIDED, 0, 1, 0, b, nb, total, 1, 2, m, 1, dst)
    if na > m:
        dst = api.rows_map(RowsRule.STRIDED, m, 1, 0, a, na, total, 2 * m, 1, na - m, 1, dst)
    if nb > m:
        dst = api.rows_map(RowsRule.STRIDED, m, 1, 0, b, nb, total, 2 * m, 1, nb - m, 1, dst)
    return dst


def evenRows(api, x):            # :1310-1319
    x = _rows(x)
    n = x.shape[0]
    return api.rows_map(RowsRule.STRIDED, 0, 2, 0, x, n, (n + 1) // 2, 0, 1, (n + 1) // 2, 1)


def oddRows(api, x):             # :1321-1330
    x = _rows(x)
    n = x.shape[0]
    return api.rows_map(RowsRule.STRIDED, 1, 2, 0, x, n, n // 2, 0, 1, n // 2, 1)


def separateRows(api, x, A, B):  # :1333-1344
    x = _rows(x)
    return api.rows_map(RowsRule.STRIDED, A, 1, 0, x, x.shape[0], B - A, 0, 1, B - A, 1)


def addRows(api, x):             # :1349-1359
    x = _rows(x)
    return api.rows_add(x, x.shape[0], 1)


def rowIndex(api, R):            # :1365-1374
    return api.rows_index(R, 1)


def normalize(api, x):           # :1041-1050
    return api.rows_normalize(_rows(x))


def mix(api, gains, *xs):        # MLDSPRouting.h:59-76: x0*g0 + (x1*g1 + (x2*g2 + ...)), gain row k repeated over x_k's rows
    from .constants import Op
    gains = _rows(gains)
    terms = [api.op_f32(Op.MULTIPLY, _rows(x), repeatRows(api, gains[k:k + 1], _rows(x).shape[0])) for k, x in enumerate(xs)]
    acc = terms[-1]
    for t in reversed(terms[:-1]):
        acc = api.op_f32(Op.ADD, t, acc)
    return acc
