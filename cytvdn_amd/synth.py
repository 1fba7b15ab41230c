"""Synthetic hyperspectral inputs: counter-based, libm-free, partition-independent.

Two generators, matching SURVEY.md section 8(d):

* ``stem4d(shape, seed)``   4D-STEM-like cube (scan x, scan y, qx, qy): a central disc plus six
  first-order Bragg discs whose bright set switches between two crystal orientations across a
  slanted grain boundary in scan space; Poisson counts (means 20 / 4 / 0.5, background 0.05).
* ``eels3d(shape, seed)``   EELS-like spectrum image (x, y, E): a decaying background in eight
  rate levels plus an edge above E = NE/2 whose extra weight follows a two-phase spatial map.

A voxel's value depends only on (seed, its GLOBAL linear index, the global shape): the hash is
splitmix64 and the Poisson draw counts how many committed 24-bit thresholds of the voxel's
rate level the hash reaches (tables: ``_synth_tables.py``, made by
``tools/make_synth_tables.py``).  Host (NumPy, this file) and device
(``tvdn_synth_fill_*`` in ``csrc/tvdn_capi.hip``) produce identical bits, as does any slab
partition of axis 0 (``row0`` / ``rows``).
"""
from __future__ import annotations

import numpy as np

from ._synth_tables import MEANS_3D, MEANS_4D, NT, THRESHOLDS

SEED_3D = 20260301
SEED_4D = 20260302

_TAB = np.asarray(THRESHOLDS, dtype=np.uint32)  # [12][NT]
_MEANS = np.asarray([float(m) for m in MEANS_4D + MEANS_3D], dtype=np.float64)
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _hash24(seed: int, lin: np.ndarray) -> np.ndarray:
    """Top 24 bits of splitmix64(seed + (lin+1)*golden)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + (lin.astype(np.uint64) + np.uint64(1)) * _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(40)).astype(np.uint32)


def level_4d(x, y, qx, qy, shape):
    """Rate level 0..3 of a 4D-STEM voxel; integer comparisons only."""
    NX, NY, NQX, NQY = (int(s) for s in shape)
    cx, cy = NQX // 2, NQY // 2
    r0 = max(NQX // 16, 1)
    R = max(NQX // 4, 1)
    rb = max(NQX // 20, 1)
    h, s = R // 2, (7 * R) // 8
    dx = qx.astype(np.int64) - cx
    dy = qy.astype(np.int64) - cy
    central = dx * dx + dy * dy <= r0 * r0
    set_a = ((R, 0), (-R, 0), (h, s), (h, -s), (-h, s), (-h, -s))
    set_b = ((0, R), (0, -R), (s, h), (s, -h), (-s, h), (-s, -h))

    def near(spots):
        m = np.zeros(np.broadcast(dx, dy).shape, dtype=bool)
        for sx, sy in spots:
            m |= (dx - sx) * (dx - sx) + (dy - sy) * (dy - sy) <= rb * rb
        return m

    in_a, in_b = near(set_a), near(set_b)
    grain_a = 4 * x.astype(np.int64) * NY < 2 * NX * NY + (2 * y.astype(np.int64) - NY) * NX
    lvl = np.zeros(np.broadcast(x, y, qx, qy).shape, dtype=np.int32)
    lvl = np.where(in_b, np.where(grain_a, 1, 2), lvl)
    lvl = np.where(in_a, np.where(grain_a, 2, 1), lvl)
    lvl = np.where(central, 3, lvl)
    return lvl.astype(np.int32)


def level_3d(x, y, e, shape):
    """Rate level 4..11 (index into the shared table) of an EELS voxel."""
    NX, NY, NE = (int(s) for s in shape)
    base = 7 - np.minimum(7, (8 * e.astype(np.int64)) // NE)
    phase_b = 4 * (x.astype(np.int64) ** 2 + y.astype(np.int64) ** 2) < NX * NX + NY * NY
    edge = (2 * e.astype(np.int64) >= NE) & phase_b
    lvl = np.minimum(7, base + np.where(edge, 2, 0))
    return (lvl + 4).astype(np.int32)


def _draw(lvl: np.ndarray, u: np.ndarray) -> np.ndarray:
    """Poisson count = number of the level's thresholds that u reaches (u >= t)."""
    out = np.zeros(u.shape, dtype=np.int32)
    for L in np.unique(lvl):
        m = lvl == L
        out[m] = np.searchsorted(_TAB[L], u[m], side="right")
    return out


def _gen(shape, seed, dtype, row0, rows, kind, ndim):
    shape = tuple(int(s) for s in shape)
    assert len(shape) == ndim
    rows = shape[0] - row0 if rows is None else rows
    assert 0 <= row0 and row0 + rows <= shape[0]
    out = np.empty((rows,) + shape[1:], dtype=dtype)
    plane = int(np.prod(shape[1:]))
    chunk = max(1, (1 << 22) // max(plane, 1))
    for r in range(0, rows, chunk):
        n = min(chunk, rows - r)
        idx = np.indices((n,) + shape[1:], dtype=np.int64)
        idx[0] += row0 + r
        if ndim == 4:
            lvl = level_4d(idx[0], idx[1], idx[2], idx[3], shape)
        else:
            lvl = level_3d(idx[0], idx[1], idx[2], shape)
        if kind == "mean":
            out[r:r + n] = _MEANS[lvl].astype(dtype)
        else:
            lin = np.ravel_multi_index(tuple(idx), shape).astype(np.uint64)
            out[r:r + n] = _draw(lvl, _hash24(seed, lin)).astype(dtype)
    return out


def stem4d(shape, seed=SEED_4D, dtype=np.float32, row0=0, rows=None, kind="counts"):
    """4D-STEM-like cube; ``kind='mean'`` returns the noise-free rates (for reference_data)."""
    return _gen(shape, seed, dtype, row0, rows, kind, 4)


def eels3d(shape, seed=SEED_3D, dtype=np.float32, row0=0, rows=None, kind="counts"):
    """EELS-like spectrum image; ``kind='mean'`` returns the noise-free rates."""
    return _gen(shape, seed, dtype, row0, rows, kind, 3)


def cube(shape, seed=None, dtype=np.float32, **kw):
    if len(shape) == 4:
        return stem4d(shape, SEED_4D if seed is None else seed, dtype, **kw)
    return eels3d(shape, SEED_3D if seed is None else seed, dtype, **kw)


__all__ = ["stem4d", "eels3d", "cube", "level_4d", "level_3d", "SEED_3D", "SEED_4D", "NT"]
