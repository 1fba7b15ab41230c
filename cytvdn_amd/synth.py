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
    z = lin.astype(np.uint64) + np.uint64(1)
    return _hash24_inplace(seed, z, np.empty_like(z)).astype(np.uint32)


def _hash24_inplace(seed: int, z: np.ndarray, tmp: np.ndarray) -> np.ndarray:
    """z holds lin+1 (uint64) on entry and the 24-bit hash on exit; no temporaries are allocated
    (fresh multi-megabyte temporaries cost more than the arithmetic)."""
    z *= _GOLD
    z += np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
    np.right_shift(z, np.uint64(30), out=tmp); z ^= tmp; z *= _M1
    np.right_shift(z, np.uint64(27), out=tmp); z ^= tmp; z *= _M2
    np.right_shift(z, np.uint64(31), out=tmp); z ^= tmp
    np.right_shift(z, np.uint64(40), out=z)
    return z


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


_KEYS = (np.arange(_TAB.shape[0], dtype=np.int64)[:, None] * (1 << 24) + _TAB.astype(np.int64)).ravel()


def _draw(lvl: np.ndarray, u: np.ndarray) -> np.ndarray:
    """Poisson count = number of the level's thresholds that u reaches (u >= t).

    One search over the concatenated tables: key = level * 2^24 + u; every threshold of a lower
    level is below the key, so the rank minus level * NT is the count within the level."""
    key = lvl.astype(np.int64) * (1 << 24) + u.astype(np.int64)
    return (np.searchsorted(_KEYS, key.ravel(), side="right").reshape(u.shape) - lvl.astype(np.int64) * NT).astype(np.int32)


def _patterns(shape):
    """The level functions factor into a class of the two leading (spatial) indices and a pattern over
    the remaining (spectral / diffraction) indices: return (pattern if class, pattern if not class),
    both flattened, evaluated with level_4d / level_3d themselves."""
    if len(shape) == 4:
        NX, NY, NQX, NQY = shape
        qx, qy = np.meshgrid(np.arange(NQX), np.arange(NQY), indexing="ij")
        zero = np.zeros_like(qx)
        pat_t = level_4d(zero, zero + NY, qx, qy, shape)   # x = 0, y = NY  -> grain A
        pat_f = level_4d(zero + NX, zero, qx, qy, shape)   # x = NX, y = 0  -> grain B
    else:
        NX, NY, NE = shape
        e = np.arange(NE)
        pat_t = level_3d(np.zeros_like(e), np.zeros_like(e), e, shape)            # x = y = 0: phase B
        pat_f = level_3d(np.zeros_like(e) + NX, np.zeros_like(e) + NY, e, shape)  # far corner: phase A
    return pat_t.ravel().astype(np.int32), pat_f.ravel().astype(np.int32)


def _classes(shape, o0, o1):
    """Class (grain A / phase B) of the spatial positions with flat index x*NY + y in [o0, o1)."""
    NX, NY = int(shape[0]), int(shape[1])
    o = np.arange(o0, o1, dtype=np.int64)
    x, y = o // NY, o % NY
    if len(shape) == 4:
        return 4 * x * NY < 2 * NX * NY + (2 * y - NY) * NX
    return 4 * (x * x + y * y) < NX * NX + NY * NY


def _gen(shape, seed, dtype, row0, rows, kind, ndim):
    shape = tuple(int(s) for s in shape)
    assert len(shape) == ndim
    rows = shape[0] - row0 if rows is None else rows
    assert 0 <= row0 and row0 + rows <= shape[0]
    out = np.empty((rows,) + shape[1:], dtype=dtype)
    NY = shape[1]
    inner = int(np.prod(shape[2:]))
    flat = out.reshape(rows * NY, inner)
    pat_t, pat_f = _patterns(shape)
    step = max(1, (1 << 21) // inner)          # spatial positions per chunk (~2M voxels)
    z = np.empty(step * inner, np.uint64)
    tmp = np.empty_like(z)
    for o in range(0, rows * NY, step):
        n = min(step, rows * NY - o)
        cls = _classes(shape, row0 * NY + o, row0 * NY + o + n)
        lvl = np.where(cls[:, None], pat_t[None, :], pat_f[None, :])
        if kind == "mean":
            flat[o:o + n] = _MEANS[lvl].astype(dtype)
        else:
            zz, tt = z[:n * inner], tmp[:n * inner]
            zz[:] = np.arange(1, n * inner + 1, dtype=np.uint64)
            zz += np.uint64((row0 * NY + o) * inner)
            u = _hash24_inplace(seed, zz, tt).reshape(n, inner)
            flat[o:o + n] = _draw(lvl, u).astype(dtype)
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
