"""Helpers to read the committed fixtures (tests/golden/*.npz, made by oracle/make_golden.py)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    return d, json.loads(str(d["manifest"]))


def bits_equal(x, y):
    """Bit-for-bit equality, NaNs included."""
    x, y = np.ascontiguousarray(x), np.ascontiguousarray(y)
    return x.dtype == y.dtype and x.shape == y.shape and x.tobytes() == y.tobytes()


def scalar_tol(dtype, n):
    """Relative tolerance for a length-n running sum kept in `dtype` against an f64 / re-ordered sum
    (SURVEY.md section 7 hard part 4; the reference's one-thread f32 running sum of 8.4e6 terms is
    itself off by ~1e-2): f32 5e-9*n clamped to [1e-5, 3e-2]; f64 1e-12."""
    if np.dtype(dtype) == np.float64:
        return 1e-12
    return float(min(3e-2, max(1e-5, 5e-9 * n)))
