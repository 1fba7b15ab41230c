"""denoise4D / denoise3D / check_memory with the reference's signatures, on the MI355X engine.

Reference: cyTVDN/cyTVDN.py:19-247 (denoise4D), :250-435 (denoise3D), :438-467 (check_memory).
Same arguments, same defaults, same assertions, same return tuple; the iteration loop
(cyTVDN.py:147-242, :368-430) runs device-resident: the datacube is copied to HBM once, every
iteration is one fused HIP sweep (csrc/tvdn_fused.hip), the convergence scalars are reduced on
the device into per-iteration slots and read back once at the end (or per iteration only when
`stopping_relative_change` asks for it), and the result is copied back once.

Differences, all deliberate:
  * BC_mode=1 raises NotImplementedError: upstream's mirror reconstruction update indexes out of
    bounds (utils.pyx:117-120) and the 3-D variant returns NaN (utils.pyx:192-197).
  * isotropic_R / isotropic_Q raise NotImplementedError: the semi-isotropic scheme "appears to have
    an error, and should not be used" (README.md:9).
  * b_norm / delta_recon / MSE are reduced in f64 by a fixed tree and then cast to the data dtype;
    upstream's dtype-width OpenMP sums drift with the thread count (1e-2 at 1e7 f32 voxels).
  * Memory notes are about HBM, not host RAM.
"""
from __future__ import annotations

import os
import threading
from typing import Optional, Tuple

import numpy as np
import torch

from .engine import DEFAULT_STATE, HipBackend, SlabLayout, SlabRunner, hbm_plan
from .planner import plan_run

try:  # tqdm is what upstream shows (cyTVDN.py:148-152); it is optional here
    from tqdm import tqdm as _tqdm
except Exception:  # pragma: no cover
    _tqdm = None


def _fmt_bytes(n: float) -> str:
    for unit in ("B", "KiB", "MiB", "GiB", "TiB"):
        if n < 1024 or unit == "TiB":
            return f"{n:.1f} {unit}" if unit != "B" else f"{int(n)} B"
        n /= 1024.0
    return f"{n:.1f} TiB"


def _hbm_free(device: int):
    if not torch.cuda.is_available():
        return None
    free, total = torch.cuda.mem_get_info(device)
    return free, total


def _split_iterations(iterations, FISTA):
    # reference cyTVDN.py:99-108 / :322-331
    unaccelerated = not FISTA
    if type(iterations) in (list, tuple):
        FISTA = True
        unaccelerated = True
        n_fista, n_plain = int(iterations[0]), int(iterations[1])
    else:
        n_fista, n_plain = int(iterations * FISTA), int(iterations * (not FISTA))
    return bool(FISTA), bool(unaccelerated), n_fista, n_plain


def _audition_candidates(n_total: int) -> int:
    """How many placements of the state `HipBackend.best_of` may try before an in-core run.  A further candidate costs
    about five sweeps' time (allocate, zero, three probe sweeps, free) and the expected gain of the best of three or four
    over a single draw is 3.5-4 % of a sweep (engine.HipBackend.best_of; profiles/r03_placement_audition_*.jsonl), so
    two extra candidates pay for themselves from ~400 iterations on and three from ~800; shorter runs take the first
    allocation.  TVDN_AUDITION=n overrides (1 = never)."""
    e = os.environ.get("TVDN_AUDITION")
    if e is not None:
        return max(1, int(e))
    return 4 if n_total >= 800 else (3 if n_total >= 400 else 1)


class _ProgressBars:
    """Upstream's two tqdm bars (cyTVDN.py:148-151, :196-199) fed from tvdn_run's progress callback, which reports the
    number of the last iteration slot handed to the GPU (the unaccelerated phase counts on from n_fista)."""

    def __init__(self, n_fista, n_plain, quiet, may_stop=False):
        self.active = _tqdm is not None and not quiet
        self.n_fista, self.n_plain = n_fista, n_plain
        self.may_stop = may_stop          # without a stopping rule a phase that is left has run all its iterations
        self.bars = [None, None]
        self.seen = [0, 0]

    def update(self, slots_done):
        phase = 0 if slots_done <= self.n_fista and self.n_fista and not self.bars[1] else 1
        if phase == 1 and self.n_fista and not self.may_stop and self.seen[0] < self.n_fista:
            self._advance(0, self.n_fista)     # one report may cover the end of one phase and the start of the next
        if phase == 1 and self.bars[0] is not None:
            self.bars[0].close()
            self.bars[0] = None
        self._advance(phase, slots_done - (self.n_fista if phase else 0))

    def _advance(self, phase, done):
        if self.bars[phase] is None:
            self.bars[phase] = _tqdm(total=self.n_fista if phase == 0 else self.n_plain,
                                     desc="FISTA Accelerated TV Denoising" if phase == 0 else "Unaccelerated TV Denoising")
        self.bars[phase].update(done - self.seen[phase])
        self.seen[phase] = done

    def close(self):
        for b in self.bars:
            if b is not None:
                b.close()
        self.bars = [None, None]


def _run(nd, datacube, mu, lam, iterations, FISTA, stopping_relative_change, reference_data, BC_mode, quiet,
         device, out=None):
    """`datacube`: NumPy array (a memory-mapped file is one: cubeio.denoise_file).  `out`: None (return the reconstruction as
    a fresh array) or a writable array of the cube's shape and dtype that receives it (a memory-mapped output file)."""
    dtype = datacube.dtype
    lambdaInv = 1.0 / lam                      # cyTVDN.py:77 / :303
    lam_mu = (lam / mu).astype(dtype)          # cyTVDN.py:78 / :304
    FISTA, unaccelerated, n_fista, n_plain = _split_iterations(iterations, FISTA)
    n_total = n_fista + n_plain

    if BC_mode == 1:
        raise NotImplementedError("BC_mode=1 (mirror): the reference's reconstruction update reads out of bounds "
                                  "(utils.pyx:117-120); not reproduced")
    if BC_mode not in (0, 2):
        raise ValueError(f"BC_mode must be 0 or 2, got {BC_mode}")
    if datacube.size == 0:
        # nothing to sweep: upstream's loops fall through, every norm is 0 and delta_recon is 0/0 (utils.pyx:125)
        res = (datacube.copy() if out is None else None, np.zeros(n_total, dtype), np.full(n_total, np.nan, dtype))
        return res + (np.zeros(n_total + 1, dtype),) if reference_data is not None else res

    if not quiet:
        plan = hbm_plan(datacube.shape, dtype, FISTA)
        kind = "FISTA Accelerated" if FISTA else "Unaccelerated"
        print(f"{kind} TV denoising will keep {plan['arrays']} arrays = {_fmt_bytes(plan['bytes'])} in HBM...", flush=True)

    if isinstance(device, (list, tuple)):
        if len(device) > 1:
            # the library decides: slabs resident where they fit their devices, else every slab streamed through its own
            # device from host arrays all of them share (tvdn_run, stream_rows / stream_k = -1 / -1); TVDN_WAVEFRONT=rows,k
            # forces the streamed form
            wf_ = os.environ.get("TVDN_WAVEFRONT")
            return _run_device_list([int(d) for d in device], datacube, lambdaInv, lam_mu, n_fista, n_plain,
                                    stopping_relative_change, reference_data, BC_mode, quiet, out=out,
                                    stream=tuple(int(v) for v in wf_.split(",")) if wf_ else (-1, -1))
        device = int(device[0]) if len(device) else None
    if device is None:
        device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    # engine choice (cytvdn_amd/planner.py): in-core when the state fits in the free HBM (or in TVDN_HBM_LIMIT),
    # otherwise streamed from pinned host memory by the library's own loop (tvdn_run, csrc/tvdn_stream.hip): wavefront
    # schedule, rows resident in HBM where they fit, chained passes; one iteration per pass when a stopping rule needs a
    # decision after every iteration.  TVDN_WAVEFRONT / TVDN_STAGED = "rows,k" force the streamed form on a cube that fits.
    # The library refuses what the host cannot hold before anything is allocated (tvdn_stream_host_need) and switches
    # the exact wrap on by itself when the first row is not finite.
    stop = stopping_relative_change
    forced = os.environ.get("TVDN_WAVEFRONT") or os.environ.get("TVDN_STAGED")
    plan = plan_run(datacube.shape, dtype, FISTA, 1, stop=stop is not None, device=device)
    if plan["mode"] == "does-not-fit" and not forced:
        raise MemoryError(f"cube of shape {datacube.shape} cannot be streamed through {_fmt_bytes(plan['hbm_bytes'])} "
                          f"of HBM: {plan['why']}")
    if forced or plan["mode"] == "wavefront":
        if forced:
            rows, k = (int(v) for v in forced.split(","))
        else:
            # the library's own plan for this much HBM: chunk height, depth, rows resident (choose_stream_shape)
            rows, k = _library_stream_plan(datacube, n_fista if FISTA else 0, n_plain if unaccelerated else 0, stop is not None,
                                           reference_data is not None, BC_mode, device, plan["hbm_bytes"])
        rows, k = max(1, rows), max(1, k)
        if not quiet:
            print(f"State exceeds HBM: streaming the cube from pinned host memory, {1 if stop is not None else k} iterations per pass "
                  f"(wavefront schedule, {rows}-row chunks)", flush=True)
        return _run_device_list([device], datacube, lambdaInv, lam_mu, n_fista if FISTA else 0, n_plain if unaccelerated else 0,
                                stop, reference_data, BC_mode, True, stream=(rows, k), out=out)
    if DEFAULT_STATE == "compact" and os.environ.get("TVDN_LOOP", "run") == "run":
        # A NumPy cube that fits: the whole call behind the library's entry point (tvdn_run, csrc/tvdn_run.hip) -- state
        # allocation and placement audition, the loop, stopping rule and MSE trace, and for long-enough Jia-Zhao runs
        # the first iterations under the upload and the last ones over the download.  The progress bars are fed from
        # the library's callback.  TVDN_LOOP=native|python keeps the loop here (engine.SlabRunner; measurement, tests).
        bars = _ProgressBars(n_fista, n_plain, quiet, may_stop=stop is not None)
        try:
            return _run_device_list([int(device)], datacube, lambdaInv, lam_mu, n_fista, n_plain, stop, reference_data,
                                    BC_mode, quiet, progress=bars.update if bars.active else None, announce=False, out=out)
        finally:
            bars.close()
    layout = SlabLayout(tuple(datacube.shape), 0, 1, int(BC_mode))
    be = HipBackend.best_of(_audition_candidates(n_total), layout, dtype, FISTA, device=device,
                            max_iters=n_total)                                # raises without a GPU
    be.set_params(lambdaInv, lam_mu)
    runner = SlabRunner(be)
    be.set_input(datacube)

    calculate_MSE = reference_data is not None
    mse_dev = ref_dev = None
    if calculate_MSE:
        ref_dev = torch.empty_like(be.orig)          # (host arrays cross through the library's pinned lanes: engine.set_input)
        torch.cuda.current_stream(device).synchronize()
        from . import _lib
        _lib.copy_to_device(np.ascontiguousarray(reference_data, dtype=dtype), ref_dev)
        mse_dev = torch.zeros(n_total + 1, dtype=torch.float64, device=be.orig.device)
        be.sse(ref_dev, mse_dev[0:1])          # MSE[0] = input vs reference (cyTVDN.py:124-125)

    bars = []

    def on_iter(slot):
        if calculate_MSE:
            be.sse(ref_dev, mse_dev[slot + 1:slot + 2])
        if bars:
            bars[-1].update(1)
        if stopping_relative_change is not None:
            s = be.sums[slot].cpu().numpy()   # one small read-back per iteration, only in this mode
            with np.errstate(divide="ignore", invalid="ignore"):
                delta = dtype.type(dtype.type(s[1]) / dtype.type(s[2]))
            if delta < stopping_relative_change:
                return True
        return False

    # nothing to do between iterations (no MSE trace, no stopping rule, no progress bar): the loop runs natively
    watched = calculate_MSE or stopping_relative_change is not None or (_tqdm is not None and not quiet)

    def phase(n, tk_phase, desc):
        if n <= 0:
            return
        if not watched:
            runner.run(n if tk_phase else 0, 0 if tk_phase else n, None)
            return
        if _tqdm is not None:
            bars.append(_tqdm(total=n, desc=desc, leave=not quiet))
        try:
            runner.run(n if tk_phase else 0, 0 if tk_phase else n, on_iter)
        finally:
            if bars:
                bars.pop().close()

    # the plain phase always starts at slot n_fista, also when the FISTA phase broke early (cyTVDN.py:201)
    if FISTA:
        phase(n_fista, True, "FISTA Accelerated TV Denoising")
        runner.iter = n_fista
    if unaccelerated:
        phase(n_plain, False, "Unaccelerated TV Denoising")

    from . import _lib
    torch.cuda.current_stream(device).synchronize()
    sums = _lib.copy_to_host(be.sums, np.float64)[:n_total] if n_total else np.zeros((0, 3))
    ran = np.zeros(n_total, dtype=bool)
    ran[runner.ran] = True
    # slots of iterations that never ran keep the reference's zero tail (cyTVDN.py:127-128)
    b_norm = np.where(ran, sums[:, 0], 0.0).astype(dtype)
    num, den = sums[:, 1].astype(dtype), sums[:, 2].astype(dtype)
    with np.errstate(divide="ignore", invalid="ignore"):
        delta_recon = np.where(ran, num / den, dtype.type(0)).astype(dtype)   # divided in the data dtype (utils.pyx:125)
    recon = be.recon_to_host()
    if out is not None:
        out[...] = recon
        recon = out

    if stopping_relative_change is not None and not quiet and unaccelerated and n_plain and not ran[-1]:
        print(f"Stopping condition reached after {int(np.nonzero(ran)[0][-1])} iterations, stopping.")

    if calculate_MSE:
        return recon, b_norm, delta_recon, _lib.copy_to_host(mse_dev, np.float64).astype(dtype)
    return recon, b_norm, delta_recon


_audition_locks: dict = {}
_audition_locks_guard = threading.Lock()


def _audition_lock(device: int) -> threading.Lock:
    """One placement audition at a time per device: the probes time sweeps with HIP events and hold several states of tens
    of GiB side by side -- two threads doing that at once would share the HBM budget and disturb each other's timings."""
    with _audition_locks_guard:
        return _audition_locks.setdefault(int(device), threading.Lock())


def _library_stream_plan(datacube, n_fista, n_plain, use_stop, mse, BC_mode, device, hbm_bytes):
    """(rows per chunk, iterations per pass) the library would choose for a streamed run of this cube in `hbm_bytes` of HBM
    (tvdn_stream_plan: depth weighed against rows kept resident); the rows it keeps resident follow from what is left."""
    import ctypes as C
    from . import _lib
    a = _lib.RunArgs(dtype=_lib.dtype_code(datacube.dtype), ndim=datacube.ndim, bc_mode=int(BC_mode), device=int(device),
                     n_fista=int(n_fista), n_plain=int(n_plain), use_stop=int(bool(use_stop)), stream_rows=-1, stream_k=-1,
                     stream_resident=-1)
    for i, v in enumerate(datacube.shape):
        a.shape[i] = int(v)
    if mse:                                   # only their being non-NULL is read: an MSE trace keeps every row on the host
        a.reference, a.mse_out = 1, 1
    po = _lib.StreamPlanOut()
    _lib.check(_lib.lib().tvdn_stream_plan(C.byref(a), int(hbm_bytes or 0), C.byref(po)))
    return int(po.rows), int(po.k)


def _wants_torch_workspace(args) -> bool:
    """Small states (under 2 GiB) come from torch's caching allocator -- allocation in microseconds, and no placement to speak
    of at that size; big ones are left to the library, which composes them from physical granules: the sweep's speed on a
    hipMalloc block of tens of GiB (which is what torch would hand over) is a draw, on granules it is not.
    TVDN_WORKSPACE=torch / library forces either."""
    import ctypes as C
    from . import _lib
    e = os.environ.get("TVDN_WORKSPACE")
    if e in ("torch", "library"):
        return e == "torch"
    if os.environ.get("TVDN_VMM", "1") == "0":
        return True
    need = C.c_int64(0)
    _lib.check(_lib.lib().tvdn_run_workspace_bytes(C.byref(args), C.byref(need)))
    from .engine import vmm_min_bytes
    return need.value < vmm_min_bytes()


def _state_workspace(args, shape, dtype, fista, n_total, device, BC_mode):
    """Device memory for the state of a resident tvdn_run, from torch's caching allocator: `hipMalloc` of tens of GiB
    takes 11 ms most times and 3-5 s some times, and a process that denoises cube after cube should pay that once
    (tvdn.h, tvdn_run_args.workspace).  Long runs get the best of a few placements (`HipBackend.best_of`, DESIGN.md
    section 3) -- the library's own audition is off when it is handed a workspace."""
    import ctypes as C
    from . import _lib
    need = C.c_int64(0)
    _lib.check(_lib.lib().tvdn_run_workspace_bytes(C.byref(args), C.byref(need)))
    cands = _audition_candidates(n_total)
    if cands > 1:
        # ctypes releases the GIL inside the library, so denoise3D/4D may run on several threads: the audition is serialised
        # per device and its backends time their sweeps on a reduction context of their OWN (the process-wide context of
        # _lib.ctx keeps one event list and one scratch buffer, which two auditions at once would race on)
        with _audition_lock(device):
            be = HipBackend.best_of(cands, SlabLayout(tuple(shape), 0, 1, int(BC_mode)), dtype, fista, device=device,
                                    max_iters=1, release_losers=False,    # kept in torch's cache for the next call (see best_of)
                                    private_ctx=True)
            torch.cuda.current_stream(int(device)).synchronize()
        slab = getattr(be, "_slab", None)
        if slab is not None and slab.numel() * slab.element_size() >= need.value and slab.data_ptr() % 256 == 0:
            return slab
    _lib.ctx(device)                             # raises without a GPU: no CPU fallback
    return torch.empty(need.value, dtype=torch.uint8, device=torch.device("cuda", int(device)))


def _run_device_list(devices, datacube, lambdaInv, lam_mu, n_fista, n_plain, stop, reference_data, BC_mode, quiet,
                     stream=None, progress=None, announce=True, out=None):
    """`device=[0, 1, ...]`: one slab of axis 0 per listed GPU inside THIS process -- the library's whole-loop entry
    (tvdn_run, csrc/tvdn_run.hip): state in HBM of each device, halo rows by peer copies over xGMI under the interior
    sweeps, global sums, global stopping rule.  No torchrun, no RCCL; every slab must fit its device (tvdn_plan says so
    when not).  The process-per-GPU form over RCCL is cytvdn_amd.distributed.denoise_slabs.
    `stream=(rows, k)` (one device): the library's out-of-core branch instead (csrc/tvdn_stream.hip) -- the cube and the
    result page-locked in place, the wavefront schedule in C++."""
    import ctypes as C
    from . import _lib
    dtype = datacube.dtype
    nd = datacube.ndim
    n = n_fista + n_plain
    if not quiet and (stream is None or len(devices) > 1) and announce:
        print(f"Cutting axis 0 into {len(devices)} slabs on devices {devices} (one process, peer copies)", flush=True)
    a = _lib.RunArgs(dtype=_lib.dtype_code(dtype), ndim=nd, bc_mode=int(BC_mode), device=devices[0], n_fista=n_fista,
                     n_plain=n_plain, use_stop=int(stop is not None), stop=float(stop or 0.0),
                     n_devices=len(devices) if (stream is None or len(devices) > 1) else 0)
    if stream is not None:
        a.stream_rows, a.stream_k = int(stream[0]), int(stream[1])
        a.stream_resident = -1       # the low rows whose state fits beside the rings stay in HBM between the passes
    if len(devices) > len(a.devices):
        raise ValueError(f"at most {len(a.devices)} devices")
    for i, d in enumerate(devices):
        a.devices[i] = d
    for i, s in enumerate(datacube.shape):
        a.shape[i] = int(s)
    for q in range(nd):
        a.clip[q] = float(lambdaInv[q])
        a.lambda_mu[q] = float(lam_mu[q])
    x = np.ascontiguousarray(datacube)
    recon = np.empty(x.shape, x.dtype) if out is None else out      # (a memory-mapped output file receives the rows directly)
    sums = np.zeros((max(n, 1), 3))
    mse = np.zeros(n + 1)
    ran = C.c_int32(0)
    phases = (C.c_int32 * 2)(0, 0)
    a.data, a.recon_out, a.sums_out = x.ctypes.data, recon.ctypes.data, sums.ctypes.data
    a.phase_iters = C.addressof(phases)
    if reference_data is not None:
        ref = np.ascontiguousarray(reference_data)
        a.reference, a.mse_out = ref.ctypes.data, mse.ctypes.data
    a.iters_run = C.addressof(ran)
    workspace = None
    if stream is not None or len(devices) > 1 or reference_data is not None:   # (the reference cube of an MSE trace too)
        if torch.cuda.is_available():
            torch.cuda.empty_cache()       # the library allocates these runs' device memory itself: hand it what torch's cache holds
    if len(devices) == 1 and stream is None and _wants_torch_workspace(a):
        workspace = _state_workspace(a, datacube.shape, dtype, n_fista > 0, n, devices[0], BC_mode)
        a.workspace, a.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    elif len(devices) == 1 and stream is None and torch.cuda.is_available():
        # a big state: the library composes it from physical granules and keeps the block for the next call (csrc/tvdn_devmem.hip,
        # tvdn_release_cache) -- what torch's cache holds from other work goes back to the driver first, so that it can
        if torch.cuda.memory_reserved(int(devices[0])) > (1 << 30):
            torch.cuda.empty_cache()
    if progress is not None:
        hook = C.CFUNCTYPE(None, C.c_int32, C.c_void_p)(lambda slots_done, _user: progress(int(slots_done)))
        a.progress = C.cast(hook, C.c_void_p)
    _lib.check(_lib.lib().tvdn_run(C.byref(a)))
    del workspace
    sums = sums[:n]
    # which slots ran, from the library's own per-phase counts (not guessed from the values: an all-zero cube has zero
    # sums in slots that DID run, and upstream reports 0/0 = NaN there); the rest keep the reference's zero tail
    done = np.zeros(n, dtype=bool)
    done[:phases[0]] = True
    done[n_fista:n_fista + phases[1]] = True
    b_norm = np.where(done, sums[:, 0], 0.0).astype(dtype)
    with np.errstate(divide="ignore", invalid="ignore"):
        delta_recon = np.where(done, sums[:, 1].astype(dtype) / sums[:, 2].astype(dtype), dtype.type(0)).astype(dtype)
    if stop is not None and not quiet and n_plain and not done[-1]:
        print(f"Stopping condition reached after {int(np.nonzero(done)[0][-1])} iterations, stopping.")
    if reference_data is not None:
        return recon, b_norm, delta_recon, mse.astype(dtype)
    return recon, b_norm, delta_recon


def denoise4D(
    datacube: np.ndarray,
    mu: np.ndarray,
    iterations: int = 10,
    FISTA: bool = True,
    stopping_relative_change: Optional[float] = None,
    isotropic_R: bool = False,
    isotropic_Q: bool = False,
    reference_data: Optional[np.ndarray] = None,
    BC_mode: int = 2,
    lam: Optional[np.ndarray] = None,
    quiet: bool = False,
    device=None,
) -> Tuple[np.ndarray, np.ndarray, np.ndarray, Optional[np.ndarray]]:
    """Proximal anisotropic TV denoising of a 4-D datacube on one MI355X (or on several: `device=[0, 1, ...]`).

    Arguments, defaults and return value are those of the reference's denoise4D
    (cyTVDN/cyTVDN.py:19-60); `device` (keyword only in practice) selects the GPU -- or, given a list of
    GPUs, cuts axis 0 into one slab per entry inside this process (peer copies over xGMI; `_run_device_list`).

    datacube  C-contiguous 4-D float32/float64 array (never modified)
    mu        4-element array, TV weight per axis, same dtype
    iterations  int, or [n_FISTA, n_unaccelerated] for the hybrid schedule
    FISTA     use FISTA acceleration (about twice the state in HBM)
    stopping_relative_change  stop a phase when delta_recon drops below this
    reference_data  noise-free cube; adds the per-iteration sum of squared errors (MSE) to the result
    BC_mode   0 periodic, 2 Jia-Zhao (default); 1 (mirror) is not supported (see module docstring)
    lam       4-element array, default mu/32
    Returns (recon, b_norm, delta_recon[, MSE]).
    """
    assert isinstance(datacube, np.ndarray) and datacube.dtype in (
        np.float32,
        np.float64,
    ), "datacube must be floating point datatype."
    if datacube.ndim != 4:
        raise TypeError("No matching signature found")
    mu = np.asarray(mu)
    if lam is None:
        lam = mu * 1.0 / 32.0
    lam = np.asarray(lam)
    assert lam.dtype == datacube.dtype, "Lambda must have same dtype as datacube."
    assert mu.dtype == datacube.dtype, "Mu must have same dtype as datacube."
    assert datacube.flags["C_CONTIGUOUS"], \
        "datacube must be C-contiguous. Try np.ascontiguousarray(datacube) on the array."
    if isotropic_R or isotropic_Q:
        raise NotImplementedError("the semi-isotropic scheme 'appears to have an error, and should not be used' "
                                  "(reference README.md:9); use the anisotropic scheme")
    if mu.shape != (4,) or lam.shape != (4,):
        raise ValueError("mu and lam must have 4 entries")
    if reference_data is not None and (reference_data.shape != datacube.shape or reference_data.dtype != datacube.dtype):
        raise ValueError("reference_data must match the datacube's shape and dtype")

    lam_mu = (lam / mu).astype(datacube.dtype)
    if not quiet:
        try:
            print(f"λ/μ ≈ [1/{mu[0]/lam[0]:.0f}, 1/{mu[1]/lam[1]:.0f}, 1/{mu[2]/lam[2]:.0f}, 1/{mu[3]/lam[3]:.0f}]")
        except Exception:
            print("lambda/mu ~ [" + ", ".join(f"1/{m/l:.0f}" for m, l in zip(mu, lam)) + "]")
    if (np.any(lam_mu > (1.0 / 32.0)) or np.any(lam_mu <= 0)) and not quiet:
        print("WARNING: Parameters must satisfy 0 < λ/μ <= 1/32 or result may diverge!")
    if not quiet:
        fr = _hbm_free(0 if device is None else (device[0] if isinstance(device, (list, tuple)) else device))
        if fr:
            print(f"Available HBM: {_fmt_bytes(fr[0])} of {_fmt_bytes(fr[1])}", flush=True)

    return _run(4, datacube, mu, lam, iterations, FISTA, stopping_relative_change, reference_data, BC_mode, quiet,
                device)


def denoise3D(
    datacube: np.ndarray,
    mu: np.ndarray,
    iterations: int = 7_500,
    stopping_relative_change: Optional[float] = None,
    BC_mode: int = 2,
    FISTA: bool = False,
    reference_data: Optional[np.ndarray] = None,
    lam: Optional[np.ndarray] = None,
    quiet: bool = False,
    device: Optional[int] = None,
) -> Tuple[np.ndarray, np.ndarray, np.ndarray, Optional[np.ndarray]]:
    """Proximal anisotropic TV denoising of a 3-D datacube on one MI355X.

    Arguments, defaults (note: FISTA=False, 7500 iterations, lam = mu/16) and return value are
    those of the reference's denoise3D (cyTVDN/cyTVDN.py:250-290).
    """
    assert isinstance(datacube, np.ndarray) and datacube.dtype in (
        np.float32,
        np.float64,
    ), "datacube must be floating point datatype."
    if datacube.ndim != 3:
        raise TypeError("No matching signature found")
    mu = np.asarray(mu)
    if lam is None:
        lam = mu / 16.0
    lam = np.asarray(lam)
    assert lam.dtype == datacube.dtype, "Lambda must have same dtype as datacube."
    assert datacube.flags["C_CONTIGUOUS"], \
        "datacube must be C-contiguous. Try np.ascontiguousarray(datacube) on the array"
    if mu.shape != (3,) or lam.shape != (3,):
        raise ValueError("mu and lam must have 3 entries")
    if reference_data is not None and (reference_data.shape != datacube.shape or reference_data.dtype != datacube.dtype):
        raise ValueError("reference_data must match the datacube's shape and dtype")

    lam_mu = (lam / mu).astype(datacube.dtype)
    # upstream's message says 1/8 while testing 1/16 (cyTVDN.py:306-308); the test is what counts
    assert np.all(lam_mu <= (1.0 / 16.0)) & np.all(lam_mu > 0), "Parameters must satisfy 0 < λ/μ <= 1/16"
    if not quiet:
        print("λ/μ ≈ [" + ", ".join(f"1/{m/l:.0f}" for m, l in zip(mu, lam)) + "]")
        fr = _hbm_free(0 if device is None else (device[0] if isinstance(device, (list, tuple)) else device))
        if fr:
            print(f"Available HBM: {_fmt_bytes(fr[0])} of {_fmt_bytes(fr[1])}", flush=True)

    return _run(3, datacube, mu, lam, iterations, FISTA, stopping_relative_change, reference_data, BC_mode, quiet,
                device)


def check_memory(datacube, device: int = 0):
    """Print whether each algorithm's state fits in this GPU's HBM (reference check_memory,
    cyTVDN.py:438-467, which does the same sum for host RAM)."""
    shape, dtype = tuple(datacube.shape), datacube.dtype
    if len(shape) not in (3, 4):
        raise TypeError("No matching signature found")
    fr = _hbm_free(device)
    avail = fr[0] if fr else None
    rows = []
    for name, fista in (("Anisotropic Unaccelerated", False), ("Anisotropic FISTA", True)):
        p = hbm_plan(shape, dtype, fista)
        ok = "?" if avail is None else ("✅" if p["bytes"] < avail else "❌")
        rows.append([name, _fmt_bytes(p["bytes"]), f"{p['arrays']} arrays", ok])
    print(f"Datacube size is {_fmt_bytes(datacube.nbytes)} with dtype {dtype}")
    if avail is not None:
        print(f"Free HBM on device {device}: {_fmt_bytes(avail)}")
        for fista in (True, False):
            p = plan_run(shape, dtype, fista, 1, device=device)
            extra = f", {p['chunk_rows']}-row chunks, {p['k']} iterations per pass" if p["k"] else ""
            print(f"denoise{len(shape)}D ({'FISTA' if fista else 'unaccelerated'}) will run: {p['mode']}{extra} -- {p['why']}")
    try:
        from tabulate import tabulate
        print(tabulate(rows, ["Algorithm", "HBM Needed", "State", "OK?"]))
    except Exception:  # pragma: no cover
        for r in rows:
            print("  ".join(r))
    return rows


__all__ = ["denoise4D", "denoise3D", "check_memory"]
