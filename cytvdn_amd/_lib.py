"""ctypes binding of libtvdn_hip.so (C ABI: include/tvdn.h).

There is no CPU fallback anywhere in this package: if the shared library is missing, or no
MI355X is visible when a compute call is made, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

# torch is imported BEFORE the library on purpose: it ships its own HIP runtime with soname
# libamdhip64.so.7, and loading it first makes libtvdn_hip.so bind to that same runtime, so
# pointers and streams handed over from torch tensors are valid in our launches.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# TVDN_LIB: measurement aid -- load another build of the same library (e.g. a -DTVDN_NT_LOADS=0 variant)
LIB_PATH = os.environ.get("TVDN_LIB") or os.path.join(_HERE, "libtvdn_hip.so")

TVDN_F32, TVDN_F64 = 0, 1
EDGE_BC, EDGE_HALO, EDGE_ZERO, EDGE_WRAP = 0, 1, 2, 3
ITER_PLAIN, ITER_FISTA, ITER_FISTA_D, ITER_FISTA_D_TO_PLAIN = 0, 1, 2, 3
SWEEP_CHAIN_LO, SWEEP_STORE_AHEAD = 1, 2

EXPORTS = (
    "tvdn_abi_version", "tvdn_last_error", "tvdn_device_count", "tvdn_ctx_create", "tvdn_ctx_destroy",
    "tvdn_ctx_timing_enable", "tvdn_ctx_timing_read", "tvdn_ctx_timing_read_each",
    "tvdn_accumulator_update", "tvdn_datacube_update", "tvdn_sum_square_error", "tvdn_iterate_fused",
    "tvdn_synth_fill", "tvdn_run", "tvdn_pipeline_plan", "tvdn_run_workspace_bytes", "tvdn_release_cache", "tvdn_copy_to_device", "tvdn_copy_to_host", "tvdn_iterate_many", "tvdn_plan", "tvdn_copy_many", "tvdn_stream_mix", "tvdn_stream_mix_march",
    "tvdn_stream_host_need", "tvdn_stream_plan", "tvdn_wait_background", "tvdn_slab_host_need", "tvdn_slab_row_map", "tvdn_fista_ratios", "tvdn_iter_mode", "tvdn_roles_bind", "tvdn_roles_advance",
    "tvdn_mem_alloc", "tvdn_mem_free", "tvdn_state_kept_bytes", "tvdn_recon_from_state",
    "tvdn_mem_alloc_shared", "tvdn_mem_status", "tvdn_mem_selftest", "tvdn_mem_resize", "tvdn_warm_up",
)
CANARY_NOT_RUN, CANARY_PASSED, CANARY_STALE, CANARY_FAILED = 0, 1, -1, -2
MEM_PLAIN, MEM_GRANULES, MEM_CALLER = 0, 1, 2


class TvdnError(RuntimeError):
    pass


class IterArgs(C.Structure):
    """struct tvdn_iter_args (include/tvdn.h)."""
    _fields_ = [
        ("dtype", C.c_int32), ("ndim", C.c_int32), ("shape", C.c_int64 * 4),
        ("row_lo", C.c_int64), ("row_hi", C.c_int64),
        ("lo_mode", C.c_int32), ("hi_mode", C.c_int32), ("bc_mode", C.c_int32), ("mode", C.c_int32),
        ("tk", C.c_double), ("tk_prev", C.c_double), ("clip", C.c_double * 4), ("lambda_mu", C.c_double * 4),
        ("orig", C.c_void_p), ("recon_in", C.c_void_p), ("recon_out", C.c_void_p),
        ("b_in", C.c_void_p * 4), ("b_out", C.c_void_p * 4), ("d_in", C.c_void_p * 4), ("d_out", C.c_void_p * 4),
        ("dprev_in", C.c_void_p * 4),
        ("sweep_lo", C.c_int64), ("sweep_hi", C.c_int64), ("accumulate", C.c_int32), ("chain", C.c_int32),
        ("wrap_recon", C.c_void_p),
        ("ring_rows", C.c_int64), ("orig_ring_rows", C.c_int64),
        ("recon_in_ring_rows", C.c_int64), ("cur_ring_rows", C.c_int64), ("prev_ring_rows", C.c_int64),
        ("recon_out_ring_rows", C.c_int64), ("out_ring_rows", C.c_int64),
    ]


class ManyArgs(C.Structure):
    """struct tvdn_many_args (include/tvdn.h)."""
    _fields_ = [
        ("base", IterArgs), ("recon", C.c_void_p * 2), ("S", (C.c_void_p * 3) * 4),
        ("cur", C.c_int32), ("i_d", C.c_int32), ("i_prev", C.c_int32), ("i_out", C.c_int32),
        ("i_b", C.c_int32), ("i_bout", C.c_int32), ("d_form", C.c_int32), ("tk_prev", C.c_double),
    ]


class PlanOut(C.Structure):
    """struct tvdn_plan_out (include/tvdn.h)."""
    _fields_ = [("arrays", C.c_int64), ("bytes_per_slab", C.c_int64), ("free_bytes", C.c_int64),
                ("fits", C.c_int32), ("min_slabs", C.c_int32)]


class RunStats(C.Structure):
    """struct tvdn_run_stats (include/tvdn.h, ABI 9)."""
    _fields_ = [
        ("engine", C.c_int32), ("pipelined", C.c_int32), ("stream_rows", C.c_int32), ("stream_k", C.c_int32),
        ("resident_rows", C.c_int64), ("n_passes", C.c_int64), ("h2d_bytes", C.c_int64), ("d2h_bytes", C.c_int64),
        ("setup_s", C.c_double), ("loop_s", C.c_double), ("total_s", C.c_double),
        ("audition_n", C.c_int32), ("audition_kept", C.c_int32), ("audition_ms", C.c_double * 8),
        ("first_pass_s", C.c_double), ("first_pass_iters", C.c_int32), ("results_under_last_pass", C.c_int32),
        ("state_mem", C.c_int32), ("kept_in_place", C.c_int32), ("peer_check", C.c_int32), ("first_call", C.c_int32),
    ]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k != "audition_ms"}
        d["audition_ms"] = [round(v, 4) for v in self.audition_ms[:max(0, min(8, self.audition_n))]]
        return d


class MemStatus(C.Structure):
    """struct tvdn_mem_status_out (include/tvdn.h, ABI 9): what the allocator of device blocks knows about a device."""
    _fields_ = [("vmm_state", C.c_int32), ("canary", C.c_int32), ("canary_runs", C.c_int32), ("faults", C.c_int32),
                ("flushes", C.c_int64), ("blocks", C.c_int64), ("granules", C.c_int64), ("bytes", C.c_int64),
                ("last_granules", C.c_int32), ("last_pool", C.c_int32), ("first_fault", C.c_char * 200)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["first_fault"] = self.first_fault.decode("utf-8", "replace")
        return d


class StreamPlanOut(C.Structure):
    """struct tvdn_stream_plan_out (include/tvdn.h, ABI 6)."""
    _fields_ = [("rows", C.c_int64), ("k", C.c_int64), ("resident_rows", C.c_int64), ("hbm_bytes", C.c_int64),
                ("host_bytes", C.c_int64)]


SLAB_EXCHANGE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int64)
SLAB_ALLREDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double))
SLAB_RELAY = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int64)


class SlabIO(C.Structure):
    """struct tvdn_slab_io (include/tvdn.h, ABI 6): one slab of a multi-process streamed run and its hooks."""
    _fields_ = [
        ("global_rows", C.c_int64), ("row0", C.c_int64), ("rank", C.c_int32), ("world", C.c_int32),
        ("first_row_nonfinite", C.c_int32), ("reserved", C.c_int32),
        ("exchange", SLAB_EXCHANGE), ("allreduce", SLAB_ALLREDUCE), ("relay_row0", SLAB_RELAY), ("user", C.c_void_p),
    ]


class RunArgs(C.Structure):
    """struct tvdn_run_args (include/tvdn.h)."""
    _fields_ = [
        ("dtype", C.c_int32), ("ndim", C.c_int32), ("shape", C.c_int64 * 4),
        ("bc_mode", C.c_int32), ("device", C.c_int32), ("n_fista", C.c_int32), ("n_plain", C.c_int32),
        ("use_stop", C.c_int32), ("n_devices", C.c_int32), ("stop", C.c_double),
        ("clip", C.c_double * 4), ("lambda_mu", C.c_double * 4),
        ("data", C.c_void_p), ("reference", C.c_void_p), ("recon_out", C.c_void_p),
        ("sums_out", C.c_void_p), ("mse_out", C.c_void_p), ("iters_run", C.c_void_p),
        ("devices", C.c_int32 * 16),
        ("stream_rows", C.c_int32), ("stream_k", C.c_int32),
        ("phase_iters", C.c_void_p),
        ("progress", C.c_void_p),
        ("progress_user", C.c_void_p),
        ("workspace", C.c_void_p),
        ("workspace_bytes", C.c_int64),
        ("stats", C.c_void_p),
        ("stream_resident", C.c_int64),
        ("slab", C.POINTER(SlabIO)),
    ]


_lib = None


def build(verbose: bool = False) -> str:
    """Compile libtvdn_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc")]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return LIB_PATH


def lib():
    """The loaded library; raises TvdnError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TvdnError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C cytvdn_amd/csrc`.  cytvdn_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    L.tvdn_abi_version.restype = C.c_int
    L.tvdn_last_error.restype = C.c_char_p
    L.tvdn_device_count.restype = C.c_int
    L.tvdn_ctx_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    L.tvdn_ctx_destroy.argtypes = [C.c_void_p]
    L.tvdn_ctx_timing_enable.argtypes = [C.c_void_p, C.c_int]
    L.tvdn_ctx_timing_read.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.tvdn_ctx_timing_read_each.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_int64)]
    i64p = C.POINTER(C.c_int64)
    L.tvdn_accumulator_update.argtypes = [C.c_void_p, C.c_int, C.c_int, i64p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_double, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p]
    L.tvdn_datacube_update.argtypes = [C.c_void_p, C.c_int, C.c_int, i64p, C.c_void_p, C.c_void_p,
                                       C.POINTER(C.c_void_p), C.POINTER(C.c_double), C.c_int, C.c_void_p, C.c_void_p]
    L.tvdn_sum_square_error.argtypes = [C.c_void_p, C.c_int, C.c_int, i64p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p]
    L.tvdn_iterate_fused.argtypes = [C.c_void_p, C.POINTER(IterArgs), C.c_void_p, C.c_void_p]
    L.tvdn_run.argtypes = [C.POINTER(RunArgs)]
    L.tvdn_plan.argtypes = [C.c_int, C.c_int, i64p, C.c_int, C.c_int, C.c_int, C.POINTER(PlanOut)]
    L.tvdn_iterate_many.argtypes = [C.c_void_p, C.POINTER(ManyArgs), C.c_int32, C.POINTER(C.c_double), C.c_int32,
                                    C.c_void_p, C.c_void_p]
    L.tvdn_copy_to_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    L.tvdn_copy_to_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    L.tvdn_copy_many.argtypes = [C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int64, C.c_int32, C.c_void_p]
    L.tvdn_stream_mix.argtypes = [C.c_int32, C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p), C.c_int64, C.c_void_p]
    L.tvdn_stream_mix_march.argtypes = [C.c_int32, C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p), C.c_int64, C.c_int64,
                                        C.c_int32, C.c_void_p]
    L.tvdn_synth_fill.argtypes = [C.c_int, C.c_int, i64p, C.c_uint64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    L.tvdn_stream_host_need.argtypes = [C.POINTER(RunArgs), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.tvdn_stream_plan.argtypes = [C.POINTER(RunArgs), C.c_int64, C.POINTER(StreamPlanOut)]
    L.tvdn_fista_ratios.argtypes = [C.c_int32, C.POINTER(C.c_double)]
    L.tvdn_iter_mode.argtypes = [C.c_int32, C.c_int32]
    L.tvdn_roles_bind.argtypes = [C.POINTER(ManyArgs), C.c_int32, C.c_double, C.POINTER(IterArgs)]
    L.tvdn_roles_advance.argtypes = [C.POINTER(ManyArgs), C.c_int32, C.c_double]
    L.tvdn_run_workspace_bytes.argtypes = [C.POINTER(RunArgs), C.POINTER(C.c_int64)]
    L.tvdn_pipeline_plan.argtypes = [C.c_int64, C.c_int32, C.c_int64, C.POINTER(C.c_int32)]
    L.tvdn_recon_from_state.argtypes = [C.c_int, C.c_int, i64p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                        C.POINTER(C.c_double), C.c_double, C.c_int64, C.c_int64, C.c_void_p]
    L.tvdn_mem_alloc.argtypes = [C.POINTER(C.c_void_p), C.c_int64, C.c_int, C.POINTER(C.c_int32)]
    L.tvdn_mem_free.argtypes = [C.c_void_p]
    L.tvdn_mem_alloc_shared.argtypes = [C.POINTER(C.c_void_p), C.c_int64, C.c_int, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32)]
    L.tvdn_mem_status.argtypes = [C.c_int, C.POINTER(MemStatus)]
    L.tvdn_mem_selftest.argtypes = [C.c_int]
    L.tvdn_mem_resize.argtypes = [C.POINTER(C.c_void_p), C.c_int64, C.c_int]
    L.tvdn_warm_up.argtypes = [C.c_int]
    L.tvdn_state_kept_bytes.argtypes = [C.c_int]
    L.tvdn_state_kept_bytes.restype = C.c_int64
    for name in EXPORTS:
        getattr(L, name)  # AttributeError here = header and library out of step
    if L.tvdn_abi_version() != 9:
        raise TvdnError("libtvdn_hip.so ABI version mismatch")
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != 0:
        msg = lib().tvdn_last_error().decode("utf-8", "replace")
        if rc == -2:
            raise NotImplementedError(msg)
        if rc == -1:
            raise ValueError(msg)
        raise TvdnError(f"libtvdn_hip status {rc}: {msg}")


def fista_ratios(n: int) -> np.ndarray:
    """(tk-1)/tk_new for iterations 0..n-1: the float64 recurrence of cyTVDN.py:153-156, computed by the library
    (csrc/tvdn_common.hpp fista_ratios, the one definition every loop in the tree uses).  Host arithmetic: no GPU needed."""
    n = int(n)
    out = np.empty(n, np.float64)
    if n:
        check(lib().tvdn_fista_ratios(n, out.ctypes.data_as(C.POINTER(C.c_double))))
    return out


def pipeline_plan(n0: int, n_iters: int, cube_bytes: int):
    """(rows per chunk, iterations under the upload, iterations over the download) of a resident tvdn_run, or None for the
    plain order (csrc/tvdn_run.hip pipeline_plan; honours TVDN_PIPELINE).  Host arithmetic: no GPU needed."""
    out = (C.c_int32 * 3)()
    check(lib().tvdn_pipeline_plan(int(n0), int(n_iters), int(cube_bytes), out))
    return tuple(out) if out[0] > 0 else None


def iter_mode(use_fista: bool, d_form: bool) -> int:
    """TVDN_ITER_* of an iteration on the compact state (csrc/tvdn_common.hpp iter_mode)."""
    m = lib().tvdn_iter_mode(int(bool(use_fista)), int(bool(d_form)))
    if m < 0:
        check(m)
    return m


def dtype_code(dt) -> int:
    dt = np.dtype(dt)
    if dt == np.float32:
        return TVDN_F32
    if dt == np.float64:
        return TVDN_F64
    raise TypeError("No matching signature found")


def shape_arr(shape):
    return (C.c_int64 * len(shape))(*[int(s) for s in shape])


_ctxs: dict[int, C.c_void_p] = {}


def ctx(device: int) -> C.c_void_p:
    """Per-device reduction scratch (created lazily; raises when no GPU is visible)."""
    h = _ctxs.get(device)
    if h is None:
        if not torch.cuda.is_available():
            raise TvdnError("no MI355X visible (torch.cuda.is_available() is False): cytvdn_amd has no CPU fallback")
        h = C.c_void_p()
        with torch.cuda.device(device):
            check(lib().tvdn_ctx_create(C.byref(h), int(device)))
        _ctxs[device] = h
    return h


def new_ctx(device: int) -> C.c_void_p:
    """A private context (own reduction scratch) for callers that drive a second stream."""
    if not torch.cuda.is_available():
        raise TvdnError("no MI355X visible (torch.cuda.is_available() is False): cytvdn_amd has no CPU fallback")
    h = C.c_void_p()
    with torch.cuda.device(device):
        check(lib().tvdn_ctx_create(C.byref(h), int(device)))
    return h


class DeviceBlock:
    """`nbytes` of device memory from the library's own allocator (tvdn_mem_alloc: composed from physical granules when it
    is big, because a hipMalloc block of tens of GiB decides by its placement how fast the sweep runs on it and such a block
    does not -- csrc/tvdn_devmem.hip).  `tensor(dtype)` views it as a 1-D torch tensor (no copy, no
    ownership: the block lives as long as this object, which the tensor's users must keep)."""

    def __init__(self, nbytes: int, device: int, peers=()):
        """`peers`: other devices that read and write the block (tvdn_mem_alloc_shared: one access descriptor per device)."""
        p, k = C.c_void_p(), C.c_int32()
        pl = (C.c_int32 * max(1, len(peers)))(*[int(d) for d in peers])
        check(lib().tvdn_mem_alloc_shared(C.byref(p), int(nbytes), int(device), pl, len(peers), C.byref(k)))
        self.ptr, self.nbytes, self.device, self.kind = int(p.value), int(nbytes), int(device), int(k.value)

    def tensor(self, dtype: "torch.dtype"):
        item = torch.empty(0, dtype=dtype).element_size()
        n = self.nbytes // item
        typestr = {torch.float32: "<f4", torch.float64: "<f8", torch.uint8: "|u1"}[dtype]

        class _View:   # the CUDA array interface (v2): what torch.as_tensor reads a raw device pointer through
            __cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (self.ptr, False), "version": 2, "strides": None}
        t = torch.as_tensor(_View(), device=torch.device("cuda", self.device))
        if t.data_ptr() != self.ptr:
            raise TvdnError("torch copied the device block instead of viewing it")
        return t

    def resize(self, nbytes: int = None):
        """The block at another size -- or, with the size it has, in another arrangement of its granules (tvdn_mem_resize): new
        address, contents undefined; views made before are dead."""
        nbytes = self.nbytes if nbytes is None else int(nbytes)
        p = C.c_void_p(self.ptr)
        rc = lib().tvdn_mem_resize(C.byref(p), nbytes, self.device)
        if rc == -3:
            self.ptr = 0        # (given back by the library)
        check(rc)
        self.ptr, self.nbytes = int(p.value), nbytes

    def free(self):
        if self.ptr:
            p, self.ptr = self.ptr, 0
            check(lib().tvdn_mem_free(C.c_void_p(p)))

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def mem_status(device: int = 0) -> dict:
    """The allocator's view of `device` (tvdn_mem_status): whether big blocks come from granules (vmm_state 1) or plain
    hipMalloc (-1: TVDN_VMM=0, refused by the runtime, or the remap canary tripped), the canary's verdict, the HIP calls of the
    allocator that failed so far (`faults`, `first_fault`), and the granule blocks alive."""
    st = MemStatus()
    check(lib().tvdn_mem_status(int(device), C.byref(st)))
    return st.as_dict()


def mem_selftest(device: int = 0) -> bool:
    """Run the remap canary on `device` now (tvdn_mem_selftest); False when it tripped or could not run -- the device then
    serves plain blocks for the rest of the process and tvdn_last_error() says what was seen."""
    return lib().tvdn_mem_selftest(int(device)) == 0


def state_kept_bytes(device: int) -> int:
    """Bytes of the device block the last tvdn_run on `device` kept for the next one: used as far as the driver reports,
    free as far as planning goes."""
    return int(lib().tvdn_state_kept_bytes(int(device)))


def copy_to_device(src: np.ndarray, dst) -> None:
    """C-contiguous NumPy array -> contiguous device tensor of the same byte size, through the library's pinned,
    multi-lane staging (csrc/tvdn_hostio.hip).  Synchronous; the caller orders it against its own streams."""
    if not src.flags["C_CONTIGUOUS"] or not dst.is_contiguous() or src.nbytes != dst.numel() * dst.element_size():
        raise ValueError("copy_to_device needs contiguous buffers of equal size")
    check(lib().tvdn_copy_to_device(C.c_void_p(dst.data_ptr()), C.c_void_p(src.ctypes.data), src.nbytes, dst.device.index))


def copy_to_host(src, dtype) -> np.ndarray:
    """Contiguous device tensor -> fresh NumPy array (first touch of the new pages spread over the staging lanes).
    Synchronous; the producing stream must have been synchronised by the caller."""
    if not src.is_contiguous():
        raise ValueError("copy_to_host needs a contiguous tensor")
    out = np.empty(tuple(src.shape), dtype=dtype)
    check(lib().tvdn_copy_to_host(C.c_void_p(out.ctypes.data), C.c_void_p(src.data_ptr()), out.nbytes, src.device.index))
    return out


def copy_many(pairs, device: int, max_blocks: int = 0) -> None:
    """[(dst_tensor_view, src_tensor_view), ...] -> copies on the current stream, batched by size into as few launches
    as possible (tvdn_copy_many); views must be contiguous, equally shaped pairs and must not overlap.  Device tensors
    or pinned host tensors; max_blocks caps the workgroups of a launch (PCIe transfers beside running sweeps)."""
    groups = {}
    for dst, src in pairs:
        nb = dst.numel() * dst.element_size()
        if nb == 0:
            continue
        if nb % 16 or dst.data_ptr() % 16 or src.data_ptr() % 16 or not dst.is_contiguous() or not src.is_contiguous():
            dst.copy_(src, non_blocking=True)    # odd sizes: the runtime's copy
            continue
        groups.setdefault(nb, []).append((dst.data_ptr(), src.data_ptr()))
    for nb, lst in groups.items():
        d = (C.c_void_p * len(lst))(*[p[0] for p in lst])
        s = (C.c_void_p * len(lst))(*[p[1] for p in lst])
        check(lib().tvdn_copy_many(len(lst), d, s, nb, int(max_blocks), current_stream(device)))


def current_stream(device: int) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
