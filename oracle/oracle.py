"""CPU oracle for the cyTVDN anisotropic TV hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import this module.  Nothing under ``cytvdn_amd/`` imports it, and the product path
raises when its HIP library is missing instead of falling back to anything in here.

Contents
--------
* ctypes bindings to ``oracle/libtvdn_oracle.so`` (our own C restatement,
  ``oracle/tvdn_oracle.c``) under the reference's kernel names and signatures
  (reference ``cyTVDN/anisotropic.pyx``, ``cyTVDN/utils.pyx``).
* ``denoise3D`` / ``denoise4D``: restatement of the reference's iteration loop
  (reference ``cyTVDN/cyTVDN.py:147-242`` and ``:368-430``) with the printing, tqdm and
  psutil parts left out.
* ``load_reference_kernels()``: imports the reference's OWN compiled kernels from
  ``oracle/_ref`` (built by ``oracle/Makefile`` from the C the reference ships).  These
  binaries stay in the build container (git-ignored and gpurun-ignored), as the reference's sources do.
* ``load_reference_driver()``: imports the reference's own ``cyTVDN/cyTVDN.py`` from
  ``/root/reference`` (build container only) -- used by ``oracle/make_golden.py``.

Parity pin: PINNED by ``tests/golden/*.npz`` (see ``oracle/make_golden.py``).
"""
from __future__ import annotations

import ctypes as C
import importlib
import importlib.util
import os
import subprocess
import sys
import types

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtvdn_oracle.so")
_lib = None
_threads = 1


def build(force: bool = False) -> None:
    """Compile the C restatement.  (The reference's own kernels, oracle/_ref, are built on demand by `make -C oracle ref`
    -- oracle/make_golden.py and tools/port_vs_reference.py do -- in the build container only.)"""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH)
        < max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("tvdn_oracle.c", "tvdn_oracle_impl.h"))
    ) or not os.path.exists(os.path.join(_HERE, "libtvdn_oracle_timed.so")):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle"], stdout=sys.stderr)   # bench.py's stdout carries one JSON line only


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_abi_version.restype = C.c_int
        assert _lib.orc_abi_version() == 1
    return _lib


def set_threads(n: int) -> None:
    """OpenMP threads used by the restatement (1 = reference order with one thread)."""
    global _threads
    _threads = int(n)


def get_threads() -> int:
    return _threads


_SUF = {np.dtype(np.float32): "f32", np.dtype(np.float64): "f64"}
_CT = {np.dtype(np.float32): C.c_float, np.dtype(np.float64): C.c_double}


def _chk(nd, *arrs):
    dt = arrs[0].dtype
    if dt not in _SUF:
        raise TypeError("No matching signature found")
    for x in arrs:
        if x.ndim != nd:
            raise TypeError("No matching signature found")
        if x.dtype != dt:
            raise ValueError("Buffer dtype mismatch")
        if not x.flags["C_CONTIGUOUS"]:
            raise ValueError("oracle restatement takes C-contiguous arrays only")
        if x.shape != arrs[0].shape:
            raise ValueError("shape mismatch")
    return dt


def _shape4(shape):
    s = (1,) * (4 - len(shape)) + tuple(int(v) for v in shape)
    return (C.c_int64 * 4)(*s)


def _ptr(x):
    return x.ctypes.data_as(C.c_void_p)


def acc_update(a, b, d, tk, ax, clip, BC_mode=2, L=None):
    """Generic accumulator update; returns (norm in T, norm in f64)."""
    nd = a.ndim
    arrs = (a, b) if d is None else (a, b, d)
    dt = _chk(nd, *arrs)
    if not (0 <= ax < nd):
        raise ValueError("ax out of range")
    if BC_mode not in (0, 1, 2):
        raise ValueError("BC_mode must be 0, 1 or 2")
    if BC_mode == 1 and a.shape[ax] < 2:
        raise ValueError("mirror BC needs at least 2 entries along ax")
    fn = getattr(L or lib(), "orc_accumulator_update_" + _SUF[dt])
    ct = _CT[dt]
    nT, n64 = C.c_double(), C.c_double()
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, ct, C.c_int, ct, C.c_int,
                   C.c_int64 * 4, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    fn(_ptr(a), _ptr(b), None if d is None else _ptr(d), ct(tk), ax + (4 - nd), ct(clip),
       int(BC_mode), _shape4(a.shape), _threads, C.byref(nT), C.byref(n64))
    return nT.value, n64.value


def recon_update(orig, recon, bs, lambda_mu, BC_mode=2, L=None):
    """Generic reconstruction update; returns (delta/rnorm in T, sum|delta| f64, sum|old| f64)."""
    nd = orig.ndim
    dt = _chk(nd, orig, recon, *bs)
    if len(bs) != nd:
        raise ValueError("need one accumulator per axis")
    if BC_mode not in (0, 2):
        raise NotImplementedError("BC_mode=1 recon update is undefined behaviour upstream (utils.pyx:117-120)")
    lm = np.ascontiguousarray(lambda_mu)
    if lm.dtype != dt:
        raise ValueError("Buffer dtype mismatch")
    if lm.shape != (nd,):
        raise ValueError("lambda_mu must have one entry per axis")
    fn = getattr(L or lib(), "orc_datacube_update_" + _SUF[dt])
    barr = (C.c_void_p * nd)(*[x.ctypes.data for x in bs])
    out = (C.c_double * 3)()
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p * nd, C.c_void_p, C.c_int, C.c_int64 * 4,
                   C.c_int, C.c_double * 3]
    fn(_ptr(orig), _ptr(recon), barr, _ptr(lm), nd, _shape4(orig.shape), _threads, out)
    return out[0], out[1], out[2]


def sse(a, b):
    """Sum of squared differences; returns (in T, in f64)."""
    dt = _chk(a.ndim, a, b)
    fn = getattr(lib(), "orc_sum_square_error_" + _SUF[dt])
    out = (C.c_double * 2)()
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64 * 4, C.c_int, C.c_double * 2]
    fn(_ptr(a), _ptr(b), _shape4(a.shape), _threads, out)
    return out[0], out[1]


# ---- reference-named kernel-level functions (return the T-precision scalar, as upstream) ----

def _need(nd, a):
    if a.ndim != nd:
        raise TypeError("No matching signature found")


def accumulator_update_4D(a, b, ax, clip, BC_mode=2):
    _need(4, a)
    return acc_update(a, b, None, 0.0, ax, clip, BC_mode)[0]


def accumulator_update_4D_FISTA(a, b, d, tk, ax, clip, BC_mode=2):
    _need(4, a)
    return acc_update(a, b, d, tk, ax, clip, BC_mode)[0]


def accumulator_update_3D(a, b, ax, clip, BC_mode=2):
    _need(3, a)
    return acc_update(a, b, None, 0.0, ax, clip, BC_mode)[0]


def accumulator_update_3D_FISTA(a, b, d, tk, ax, clip, BC_mode=2):
    _need(3, a)
    return acc_update(a, b, d, tk, ax, clip, BC_mode)[0]


def datacube_update_4D(orig, recon, b1, b2, b3, b4, lambda_mu, BC_mode=2):
    _need(4, orig)
    return recon_update(orig, recon, (b1, b2, b3, b4), lambda_mu, BC_mode)[0]


def datacube_update_3D(orig, recon, b1, b2, b3, lambda_mu, BC_mode=2):
    _need(3, orig)
    return recon_update(orig, recon, (b1, b2, b3), lambda_mu, BC_mode)[0]


def sum_square_error_4D(a, b):
    _need(4, a)
    return sse(a, b)[0]


def sum_square_error_3D(a, b):
    _need(3, a)
    return sse(a, b)[0]


# ---- timed variant: the same restatement compiled without the f64 yardstick sums ----

_TIMED_PATH = os.path.join(_HERE, "libtvdn_oracle_timed.so")
_timed = None


def timed_kernels():
    """Namespace with reference-named kernel functions backed by libtvdn_oracle_timed.so (-DORC_NO_YARDSTICK):
    the reference's passes, visiting order, dtype-width sums and serial boundary hyperslab, nothing else.
    This is what bench.py's cpu_baseline times (kind "port"); arrays are bit-identical to the full oracle."""
    global _timed
    if _timed is None:
        if not os.path.exists(_TIMED_PATH):
            build(force=True)
        _timed = C.CDLL(_TIMED_PATH)
    L = _timed
    ns = types.SimpleNamespace()
    ns.accumulator_update_4D = lambda a, b, ax, clip, BC_mode=2: acc_update(a, b, None, 0.0, ax, clip, BC_mode, L)[0]
    ns.accumulator_update_4D_FISTA = lambda a, b, d, tk, ax, clip, BC_mode=2: acc_update(a, b, d, tk, ax, clip, BC_mode, L)[0]
    ns.accumulator_update_3D = ns.accumulator_update_4D
    ns.accumulator_update_3D_FISTA = ns.accumulator_update_4D_FISTA
    ns.datacube_update_4D = lambda o, r, b1, b2, b3, b4, lm, BC_mode=2: recon_update(o, r, (b1, b2, b3, b4), lm, BC_mode, L)[0]
    ns.datacube_update_3D = lambda o, r, b1, b2, b3, lm, BC_mode=2: recon_update(o, r, (b1, b2, b3), lm, BC_mode, L)[0]
    return ns


# ---- the iteration loop (reference cyTVDN/cyTVDN.py:147-242, :368-430) ----

def fista_schedule(n):
    """tk_ratio for FISTA iterations 0..n-1, float64 as upstream (cyTVDN.py:153-156)."""
    tk = 1.0
    out = np.empty(n, np.float64)
    for i in range(n):
        tk_new = (1 + np.sqrt(1 + 4 * tk ** 2)) / 2
        out[i] = (tk - 1.0) / tk_new
        tk = tk_new
    return out


def denoise(datacube, mu, iterations, FISTA, stopping_relative_change=None, reference_data=None,
            BC_mode=2, lam=None, return_state=False):
    """Loop restatement for 3-D and 4-D.  Returns dict with recon, b_norm, delta_recon
    (T, as upstream), their f64 yardsticks, MSE when reference_data is given, and the
    final acc/d arrays when return_state is set."""
    nd = datacube.ndim
    dt = datacube.dtype
    if lam is None:
        lam = mu * 1.0 / 32.0 if nd == 4 else mu / 16.0
    lambdaInv = 1.0 / lam
    lam_mu = (lam / mu).astype(dt)
    unacc = not FISTA
    if type(iterations) in (list, tuple):
        FISTA, unacc = True, True
        nF, nU = int(iterations[0]), int(iterations[1])
    else:
        nF, nU = int(iterations * FISTA), int(iterations * (not FISTA))
    n = nF + nU
    b_norm = np.zeros(n, dt)
    delta_recon = np.zeros(n, dt)
    b_norm64 = np.zeros(n, np.float64)
    delta64 = np.zeros(n, np.float64)
    rnorm64 = np.zeros(n, np.float64)
    mse = mse64 = None
    if reference_data is not None:
        mse = np.zeros(n + 1, dt)
        mse64 = np.zeros(n + 1, np.float64)
        mse[0], mse64[0] = sse(datacube, reference_data)
    acc = [np.zeros_like(datacube) for _ in range(nd)]
    dd = [np.zeros_like(datacube) for _ in range(nd)] if FISTA else None
    recon = datacube.copy()
    iters_done = 0

    def one(i, tk_ratio):
        for ax in range(nd):
            nT, n64 = acc_update(recon, acc[ax], dd[ax] if tk_ratio is not None else None,
                                 0.0 if tk_ratio is None else tk_ratio, ax, lambdaInv[ax], BC_mode)
            b_norm[i] += nT
            b_norm64[i] += n64
        r, dl, rn = recon_update(datacube, recon, acc, lam_mu, BC_mode)
        delta_recon[i] = r
        delta64[i], rnorm64[i] = dl, rn
        if mse is not None:
            mse[i + 1], mse64[i + 1] = sse(reference_data, recon)
        return stopping_relative_change is not None and delta_recon[i] < stopping_relative_change

    if FISTA:
        tks = fista_schedule(nF)
        for i in range(nF):
            iters_done += 1
            if one(i, tks[i]):
                break
    if unacc:
        for j in range(nU):
            iters_done += 1
            if one(j + nF, None):
                break
    out = dict(recon=recon, b_norm=b_norm, delta_recon=delta_recon, b_norm64=b_norm64,
               delta64=delta64, rnorm64=rnorm64, iters_done=iters_done)
    if mse is not None:
        out["MSE"], out["MSE64"] = mse, mse64
    if return_state:
        out["acc"], out["d"] = acc, dd
    return out


def denoise4D(datacube, mu, iterations=10, FISTA=True, stopping_relative_change=None,
              isotropic_R=False, isotropic_Q=False, reference_data=None, BC_mode=2, lam=None, quiet=False):
    assert datacube.ndim == 4 and not isotropic_R and not isotropic_Q
    r = denoise(datacube, mu, iterations, FISTA, stopping_relative_change, reference_data, BC_mode, lam)
    t = (r["recon"], r["b_norm"], r["delta_recon"])
    return t + (r["MSE"],) if reference_data is not None else t


def denoise3D(datacube, mu, iterations=7500, stopping_relative_change=None, BC_mode=2, FISTA=False,
              reference_data=None, lam=None, quiet=False):
    assert datacube.ndim == 3
    r = denoise(datacube, mu, iterations, FISTA, stopping_relative_change, reference_data, BC_mode, lam)
    t = (r["recon"], r["b_norm"], r["delta_recon"])
    return t + (r["MSE"],) if reference_data is not None else t


# ---- the real reference (build container only: oracle/_ref is git-ignored and gpurun-ignored) ----

_REF_DIR = os.path.join(_HERE, "_ref")


def have_reference_kernels() -> bool:
    d = os.path.join(_REF_DIR, "cyTVDN")
    return os.path.isdir(d) and any(f.startswith("anisotropic") for f in os.listdir(d)) \
        and any(f.startswith("utils") for f in os.listdir(d))


def load_reference_kernels():
    """Namespace with the reference's own compiled kernel functions (oracle/_ref)."""
    if not have_reference_kernels() and os.path.isdir("/root/reference/cyTVDN"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"], stdout=sys.stderr)   # build container: on demand
    if not have_reference_kernels():
        raise ImportError("oracle/_ref is empty: run `make -C oracle ref` where /root/reference exists")
    if _REF_DIR not in sys.path:
        sys.path.insert(0, _REF_DIR)
    an = importlib.import_module("cyTVDN.anisotropic")
    ut = importlib.import_module("cyTVDN.utils")
    ns = types.SimpleNamespace()
    for m in (an, ut):
        for k in dir(m):
            if k.startswith(("accumulator_update", "datacube_update", "sum_square_error")):
                setattr(ns, k, getattr(m, k))
    return ns


def load_reference_driver(reference="/root/reference"):
    """Import the reference's own cyTVDN/cyTVDN.py (build container only).

    The driver imports ``hurry.filesize`` (absent from this image) solely to pretty-print
    byte counts (cyTVDN.py:13,95,113,118).  A two-attribute in-process placeholder
    module is registered for that name; it is never written to disk and takes no part
    in any arithmetic."""
    path = os.path.join(reference, "cyTVDN", "cyTVDN.py")
    if not os.path.exists(path):
        raise ImportError(f"{path} not present (reference sources do not travel)")
    load_reference_kernels()
    if "hurry.filesize" not in sys.modules:
        h, hf = types.ModuleType("hurry"), types.ModuleType("hurry.filesize")
        hf.alternative = None
        hf.size = lambda n, system=None: f"{n} B"
        h.filesize = hf
        sys.modules["hurry"], sys.modules["hurry.filesize"] = h, hf
    name = "cyTVDN.cyTVDN"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod
