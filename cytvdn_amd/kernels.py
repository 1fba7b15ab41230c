"""Kernel-level functions under the reference's names and signatures.

Reference: cyTVDN/anisotropic.pyx (accumulator_update_*), cyTVDN/utils.pyx (datacube_update_*,
sum_square_error_*); re-exported by cyTVDN/__init__.py:1 and called directly by cyTVDN/mpi.py:317-398.

Each call is ONE HIP pass through the C ABI (include/tvdn.h).  Arguments may be
  * NumPy arrays (the reference's calling convention): staged to HBM, updated there, copied back
    into the caller's arrays, which are therefore mutated in place exactly as upstream; or
  * torch CUDA tensors, or any other device array that speaks DLPack (`__dlpack__`: CuPy, JAX, ...):
    updated in place in HBM with no host round trip and no copy.
The return value is the Python float the reference returns (kept in f64 by a fixed reduction
tree instead of the reference's thread-count-dependent dtype-width sum).

Error behaviour follows the Cython fused-type dispatch: wrong rank or a non-float dtype raises
TypeError("No matching signature found"), mixed dtypes raise ValueError("Buffer dtype mismatch ...");
where upstream has undefined behaviour (ax out of range, shapes that disagree, unknown BC_mode,
mirror-BC reconstruction update) this raises ValueError / NotImplementedError instead.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

_TORCH_DT = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}
_CNAME = {np.dtype(np.float32): "float", np.dtype(np.float64): "double"}


def _as_array(x):
    """NumPy arrays and torch tensors pass; any other DLPack producer is viewed (zero-copy) as a torch tensor."""
    if isinstance(x, (np.ndarray, torch.Tensor)) or not hasattr(x, "__dlpack__"):
        return x
    return torch.from_dlpack(x)


def _np_dtype(x):
    if isinstance(x, torch.Tensor):
        return {torch.float32: np.dtype(np.float32), torch.float64: np.dtype(np.float64)}.get(x.dtype)
    return x.dtype if isinstance(x, np.ndarray) else None


class _Staged:
    """Arrays of one call, resident in HBM; copies NumPy outputs back on `finish`."""

    def __init__(self, nd, arrays, writable):
        arrays = [_as_array(x) for x in arrays]
        first = arrays[0]
        if not isinstance(first, (np.ndarray, torch.Tensor)) or first.ndim != nd:
            raise TypeError("No matching signature found")
        dt = _np_dtype(first)
        if dt not in _TORCH_DT:
            raise TypeError("No matching signature found")
        self.dt = dt
        self.shape = tuple(int(s) for s in first.shape)
        self.dev_tensors, self.back = [], []
        dev = None
        # validate everything before touching the GPU, so argument errors surface as upstream
        for x in arrays:
            if not isinstance(x, (np.ndarray, torch.Tensor)) or x.ndim != nd:
                raise TypeError("No matching signature found")
            xdt = _np_dtype(x)
            if xdt != dt:
                got = _CNAME.get(xdt, str(xdt))
                raise ValueError(f"Buffer dtype mismatch, expected '{_CNAME[dt]}' but got '{got}'")
            if tuple(x.shape) != self.shape:
                raise ValueError(f"array shapes disagree: {tuple(x.shape)} vs {self.shape}")
            if isinstance(x, np.ndarray):
                if not x.flags.writeable:
                    raise ValueError("buffer source array is read-only")
            else:
                if not x.is_cuda:
                    raise ValueError("torch tensors must live on the GPU (NumPy arrays are staged automatically)")
                if not x.is_contiguous():
                    raise ValueError("device tensors must be contiguous")
                dev = x.device.index
        self._arrays, self._writable, self._dev = arrays, writable, dev

    def stage(self):
        dev = self._dev
        if dev is None:
            dev = torch.cuda.current_device() if torch.cuda.is_available() else 0
        self.device = dev
        _lib.ctx(self.device)  # raises TvdnError without a GPU: no CPU fallback
        cuda = torch.device("cuda", self.device)
        for x, w in zip(self._arrays, self._writable):
            if isinstance(x, np.ndarray):
                # host arrays cross PCIe through the library's pinned multi-lane staging (csrc/tvdn_hostio.hip)
                t = torch.empty(self.shape, dtype=_TORCH_DT[self.dt], device=cuda)
                torch.cuda.current_stream(self.device).synchronize()
                _lib.copy_to_device(np.ascontiguousarray(x), t)
                if w:
                    self.back.append((x, t))
            else:
                t = x
            self.dev_tensors.append(t)
        self.out = torch.zeros(4, dtype=torch.float64, device=cuda)

    def finish(self):
        vals = self.out.cpu().numpy()  # synchronises the stream
        for x, t in self.back:
            if x.flags["C_CONTIGUOUS"]:
                _lib.check(_lib.lib().tvdn_copy_to_host(C.c_void_p(x.ctypes.data), C.c_void_p(t.data_ptr()), x.nbytes,
                                                        self.device))       # straight into the caller's array
            else:
                x[...] = _lib.copy_to_host(t, self.dt)                       # strided view: via a dense temporary
        return vals


def _acc(nd, a, b, d, tk, ax, clip, BC_mode):
    arrays = [a, b] + ([d] if d is not None else [])
    st = _Staged(nd, arrays, [False, True] + ([True] if d is not None else []))
    ax = int(ax)
    if not (0 <= ax < nd):
        raise ValueError(f"ax = {ax} out of range for a {nd}-D array")
    if int(BC_mode) not in (0, 1, 2):
        raise ValueError(f"BC_mode must be 0, 1 or 2, got {BC_mode}")
    if int(BC_mode) == 1 and st.shape[ax] < 2:
        raise ValueError("mirror BC needs at least 2 entries along ax")
    st.stage()
    t = st.dev_tensors
    _lib.check(_lib.lib().tvdn_accumulator_update(
        _lib.ctx(st.device), _lib.dtype_code(st.dt), nd, _lib.shape_arr(st.shape), t[0].data_ptr(), t[1].data_ptr(),
        t[2].data_ptr() if d is not None else None, float(tk), ax, float(clip), int(BC_mode), st.out.data_ptr(),
        _lib.current_stream(st.device)))
    return float(st.finish()[0])


def accumulator_update_4D(a, b, ax, clip, BC_mode=2):
    """b <- clip(a - roll(a,1,ax) + b) in place; returns sum|b| (reference anisotropic.pyx:17-84)."""
    return _acc(4, a, b, None, 0.0, ax, clip, BC_mode)


def accumulator_update_4D_FISTA(a, b, d, tk, ax, clip, BC_mode=2):
    """FISTA form, updates b and d in place (reference anisotropic.pyx:89-164)."""
    return _acc(4, a, b, d, tk, ax, clip, BC_mode)


def accumulator_update_3D(a, b, ax, clip, BC_mode=2):
    """Reference anisotropic.pyx:169-237."""
    return _acc(3, a, b, None, 0.0, ax, clip, BC_mode)


def accumulator_update_3D_FISTA(a, b, d, tk, ax, clip, BC_mode=2):
    """Reference anisotropic.pyx:243-317."""
    return _acc(3, a, b, d, tk, ax, clip, BC_mode)


def _recon(nd, orig, recon, bs, lambda_mu, BC_mode):
    st = _Staged(nd, [orig, recon] + list(bs), [False, True] + [False] * nd)
    lm = lambda_mu.detach().cpu().numpy() if isinstance(lambda_mu, torch.Tensor) else np.asarray(lambda_mu)
    if lm.ndim != 1:
        raise TypeError("No matching signature found")
    if lm.dtype != st.dt:
        raise ValueError(f"Buffer dtype mismatch, expected '{_CNAME[st.dt]}' but got '{_CNAME.get(lm.dtype, lm.dtype)}'")
    if lm.shape[0] < nd:
        raise ValueError(f"lambda_mu needs {nd} entries")
    if int(BC_mode) == 1:
        raise NotImplementedError("BC_mode=1 (mirror) reconstruction update reads out of bounds upstream "
                                  "(utils.pyx:117-120, :192-197): unsupported")
    if int(BC_mode) not in (0, 2):
        raise ValueError(f"BC_mode must be 0 or 2, got {BC_mode}")
    st.stage()
    t = st.dev_tensors
    bptr = (C.c_void_p * nd)(*[x.data_ptr() for x in t[2:]])
    lmd = (C.c_double * nd)(*[float(v) for v in lm[:nd]])
    _lib.check(_lib.lib().tvdn_datacube_update(
        _lib.ctx(st.device), _lib.dtype_code(st.dt), nd, _lib.shape_arr(st.shape), t[0].data_ptr(), t[1].data_ptr(),
        bptr, lmd, int(BC_mode), st.out.data_ptr(), _lib.current_stream(st.device)))
    v = st.finish()
    # the reference divides its two dtype-width sums in the array dtype (utils.pyx:125)
    with np.errstate(divide="ignore", invalid="ignore"):
        return float(st.dt.type(v[0]) / st.dt.type(v[1]))


def datacube_update_4D(orig, recon, b1, b2, b3, b4, lambda_mu, BC_mode=2):
    """recon <- orig - sum lambda_mu*(b - roll(b,-1)) in place; returns sum|delta|/sum|old| (utils.pyx:54-125)."""
    return _recon(4, orig, recon, (b1, b2, b3, b4), lambda_mu, BC_mode)


def datacube_update_3D(orig, recon, b1, b2, b3, lambda_mu, BC_mode=2):
    """Reference utils.pyx:131-199."""
    return _recon(3, orig, recon, (b1, b2, b3), lambda_mu, BC_mode)


def _sse(nd, a, b):
    st = _Staged(nd, [a, b], [False, False])
    st.stage()
    t = st.dev_tensors
    _lib.check(_lib.lib().tvdn_sum_square_error(
        _lib.ctx(st.device), _lib.dtype_code(st.dt), nd, _lib.shape_arr(st.shape), t[0].data_ptr(), t[1].data_ptr(),
        st.out.data_ptr(), _lib.current_stream(st.device)))
    return float(st.finish()[0])


def sum_square_error_4D(a, b):
    """sum((a-b)**2) -- a sum, not a mean, as upstream (utils.pyx:14-30)."""
    return _sse(4, a, b)


def sum_square_error_3D(a, b):
    """Reference utils.pyx:35-49."""
    return _sse(3, a, b)


def iso_accumulator_update_4D(*args, **kwargs):
    """Semi-isotropic scheme: declared erroneous upstream (README.md:9) and racy
    (halfisotropic.pyx:70-82); deliberately not implemented."""
    raise NotImplementedError("the semi-isotropic scheme 'appears to have an error, and should not be used' "
                              "(reference README.md:9); it is not part of this engine")


def iso_accumulator_update_4D_FISTA(*args, **kwargs):
    """See iso_accumulator_update_4D."""
    return iso_accumulator_update_4D()


__all__ = [
    "accumulator_update_4D", "accumulator_update_4D_FISTA", "accumulator_update_3D", "accumulator_update_3D_FISTA",
    "datacube_update_4D", "datacube_update_3D", "sum_square_error_4D", "sum_square_error_3D",
    "iso_accumulator_update_4D", "iso_accumulator_update_4D_FISTA",
]
