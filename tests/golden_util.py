"""Helpers to read the committed fixtures (tests/golden/*.npz, made by oracle/make_golden.py)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    return d, json.loads(str(d["manifest"]))


def bits_equal(x, y):
    """Bit-for-bit equality of every non-NaN element, and NaNs in exactly the same places.

    The sign/payload of a NaN *result* is implementation-defined in IEEE 754 (x86 SUBSS hands back
    its NaN operand untouched, the gfx950 VALU may apply a negate source modifier to it), so a NaN is
    only required to BE a NaN where the reference has one."""
    x, y = np.ascontiguousarray(x), np.ascontiguousarray(y)
    if x.dtype != y.dtype or x.shape != y.shape:
        return False
    if x.tobytes() == y.tobytes():
        return True
    nx, ny = np.isnan(x), np.isnan(y)
    if not np.array_equal(nx, ny):
        return False
    ui = np.uint32 if x.dtype == np.float32 else np.uint64
    return bool(np.array_equal(x.view(ui)[~nx], y.view(ui)[~ny]))


def scalar_tol(dtype, n):
    """Relative tolerance for a length-n running sum kept in `dtype` against an f64 / re-ordered sum
    (SURVEY.md section 7 hard part 4; the reference's one-thread f32 running sum of 8.4e6 terms is
    itself off by ~1e-2): f32 5e-9*n clamped to [1e-4, 3e-2] (a 2e3-term f32 running sum is already 1.5e-5 off);
    f64 1e-12."""
    if np.dtype(dtype) == np.float64:
        return 1e-12
    return float(min(3e-2, max(1e-4, 5e-9 * n)))
