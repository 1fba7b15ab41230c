"""Capacity planner for HBM: which engine runs a cube, with how many slabs.

The reference's `check_memory` (cyTVDN/cyTVDN.py:438-467) adds up host RAM for its four algorithm variants and
prints a table; the choice is left to the user.  Here the same sum is done for HBM and the choice is made:

    plan = plan_run(shape, dtype, FISTA=True, n_gpus=8)
    plan["mode"]      "in-core"            one GPU holds the whole state: one fused sweep per iteration
                      "slabs"              axis 0 cut into plan["n_slabs"] slabs, one GPU each, halo rows over RCCL
                      "wavefront"          one GPU, state in pinned host memory, streamed plan["k"] iterations per
                                           pass in plan["chunk_rows"]-row chunks (tvdn_run, csrc/tvdn_stream.hip); with
                                           a stopping rule k = 1: a decision after every iteration
                      "slabs+wavefront"    every GPU streams its own slab from pinned host memory (BASELINE config 5)
                      "does-not-fit"       not even one chunk window fits

`denoise3D/4D` call it with n_gpus = 1 for the choice between resident and streamed (driver._run); the streamed run's own
shape -- chunk height, depth, rows kept resident in HBM -- is then the library's (tvdn_stream_plan); `check_memory` prints it.
The HBM figure is the free memory of the device (`torch.cuda.mem_get_info`) unless `hbm_bytes` is given or the
environment variable TVDN_HBM_LIMIT (bytes; suffixes K/M/G/T = KiB.. allowed, e.g. "48G") caps it -- the knob that
lets a test, or a cautious user sharing a GPU, drive the automatic out-of-core branch on a small cube.
"""
from __future__ import annotations

import os

import numpy as np

HEADROOM = 0.9            # fraction of the free HBM an in-core state may take
STAGING_FRACTION = 0.85   # fraction of the free HBM the level windows + staging boxes of a streamed run may take (as
                          # csrc/tvdn_stream.hip choose_stream_shape: on PCIe-bound planes depth is speed)
MAX_DEPTH = 128           # iterations per pass beyond which the sweeps, not PCIe, set the pace


def _parse_bytes(text: str) -> int:
    t = text.strip().upper().rstrip("B").rstrip("I")
    mult = {"K": 2 ** 10, "M": 2 ** 20, "G": 2 ** 30, "T": 2 ** 40}
    if t and t[-1] in mult:
        return int(float(t[:-1]) * mult[t[-1]])
    return int(float(t))


def _torch_cache_idle(torch, device: int) -> int:
    """Bytes torch's caching allocator holds on `device` without using them.  (torch.cuda.memory_reserved / memory_allocated
    flatten the allocator's whole statistics tree on every call -- 0.2 ms each, a third of what denoise3D spends on a
    64 x 64 x 256 cube outside its iterations; the two counters are read from the tree directly where that is possible.)"""
    try:
        st = torch._C._cuda_memoryStats(int(device))
        return max(0, int(st["reserved_bytes"]["all"]["current"]) - int(st["allocated_bytes"]["all"]["current"]))
    except Exception:
        return max(0, int(torch.cuda.memory_reserved(device)) - int(torch.cuda.memory_allocated(device)))


def hbm_available(device: int = 0, hbm_bytes: int = None, enough: int = None):
    """Bytes of HBM the planner may count on: explicit > min(TVDN_HBM_LIMIT, free) > free > None (no GPU).  `free` counts
    what torch's caching allocator holds but does not use: a process that has just denoised one cube keeps that cube's
    state block cached, and the next call either reuses it (resident runs allocate through torch) or has it released first
    (driver.py empties the cache before it hands a run to the library's own allocations) -- without this the second large
    cube of a process would be sent to the streamed engines although it fits.
    `enough`: the caller's question is only whether this many bytes are there; when what the driver reports free already
    answers it, that figure is returned without asking torch and the library what they hold (a lower bound of the whole)."""
    if hbm_bytes is not None:
        return int(hbm_bytes)
    free = None
    try:
        import torch
        if torch.cuda.is_available():
            free = int(torch.cuda.mem_get_info(device)[0])
            if enough is not None and free >= enough:
                cap = os.environ.get("TVDN_HBM_LIMIT")
                return free if not cap else min(_parse_bytes(cap), free)
            free += _torch_cache_idle(torch, device)
            # ... and the block the library's last run on this device kept for the next one (tvdn_release_cache): a streamed
            # run keeps its rings and resident rows, ~85 % of the HBM -- counted as used, the next plan saw 15 % free, sent
            # a cube that fits to the streamed engine or refused one that streams (planner.py:53)
            from . import _lib
            free += _lib.state_kept_bytes(device)
    except Exception:
        free = None
    cap = os.environ.get("TVDN_HBM_LIMIT")
    if cap:
        cap = _parse_bytes(cap)
        return cap if free is None else min(cap, free)
    return free


def host_available():
    """Bytes of host memory a streamed run may pin: what the kernel calls available, never more than the machine has,
    never more than the memory limit of the process's control group (the limit, not limit minus usage: usage counts
    page cache the kernel would give back); TVDN_HOST_LIMIT caps it (tests).  None: unknown.
    Same arithmetic as the library's own check (csrc/tvdn_stream.hip, host_available_bytes)."""
    avail = None
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    avail = int(line.split()[1]) * 1024
                    break
    except OSError:
        pass
    try:
        physical = os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
        if physical > 0 and (avail is None or avail > physical):
            avail = physical
    except (ValueError, OSError):
        pass

    def num(path):
        try:
            with open(path) as f:
                return int(f.read().split()[0])
        except (OSError, ValueError, IndexError):        # "max" does not parse: no limit
            return None

    for lim_p in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        lim = num(lim_p)
        if lim is not None and lim > 0:
            avail = lim if avail is None else min(avail, lim)
            break
    cap = os.environ.get("TVDN_HOST_LIMIT")
    if cap:
        cap = _parse_bytes(cap)
        avail = cap if avail is None else min(avail, cap)
    return avail


HOST_FRACTION = 0.8       # of host_available() a streamed run's pinned state may take


def check_host_fits(plan: dict, ranks_on_host: int = 1) -> None:
    """Raise MemoryError when the pinned host state of a streamed plan cannot fit this host: page-locked memory
    cannot swap, and a host driven out of memory takes every process on it down.  In-core plans always pass."""
    need = plan.get("host_bytes_per_rank")
    if not need:
        return
    need = (need + (plan.get("host_beside_per_rank") or 0)) * max(1, int(ranks_on_host))
    avail = host_available()
    if avail is not None and need > HOST_FRACTION * avail:
        raise MemoryError(f"streaming this cube needs {need / 2 ** 30:.1f} GiB of page-locked host memory "
                          f"({plan['mode']}, k = {plan['k']}), which exceeds {HOST_FRACTION:.0%} of the "
                          f"{avail / 2 ** 30:.1f} GiB this host has available: use more nodes (more slabs), or a "
                          f"shallower k")


def state_arrays(ndim: int, fista: bool) -> int:
    """Arrays of the fused engine's state: orig, recon x2, and per axis three rotating d arrays (FISTA) or two b."""
    return 3 + ndim * (3 if fista else 2)


def wavefront_windows(ndim: int, rows: int, k: int) -> int:
    """Rows of HBM a streamed run keeps resident for chunk height `rows` and depth k with no row of the cube resident
    (csrc/tvdn_stream.hip stream_planes, without an MSE trace: recon rings for levels 0..k, accumulator rings for levels -1..k
    per axis, the input ring, in/out boxes, the planes kept aside for an exact wrap)."""
    return (((k + 1) + (k + 2) * ndim) * (rows + 2) + (rows + k + 3) + 2 * (3 + 4 * ndim) * rows
            + 2 * (1 + 2 * ndim) + 2 * (k + 1) + 1)     # out boxes hold rows + 1; the planes of an exact wrap; the plane of zeros


def plan_run(shape, dtype, FISTA: bool = True, n_gpus: int = 1, hbm_bytes: int = None, stop: bool = False,
             device: int = 0, host_bytes: int = None, swap_through_host: bool = False) -> dict:
    """Pick the engine and the slab count for a cube of `shape` / `dtype` on `n_gpus` GPUs of `hbm_bytes` each.

    Returns a dict: mode, n_slabs, arrays, bytes_per_gpu (HBM the chosen mode needs per GPU), state_bytes (whole
    state), hbm_bytes (what was assumed available), chunk_rows / k for streamed modes, host_bytes_per_rank for them
    (page-locked) and host_beside_per_rank (what a rank of denoise_slabs holds beside that in the same memory: its slab, the
    result, and with `swap_through_host` -- gloo -- the rows of one swap in flight), min_slabs_in_core (smallest slab count
    whose slab state fits, or None), and a one-line `why`."""
    shape = tuple(int(s) for s in shape)
    nd = len(shape)
    if nd not in (3, 4):
        raise TypeError("No matching signature found")
    item = np.dtype(dtype).itemsize
    plane = int(np.prod(shape[1:])) * item
    n0 = shape[0]
    n_arr = state_arrays(nd, FISTA)
    # (one GPU: the only question is whether the whole state fits -- for a small cube the driver's free figure says so at once)
    avail = hbm_available(device, hbm_bytes, enough=int(n_arr * n0 * plane / HEADROOM) + 1 if n_gpus <= 1 else None)
    out = dict(arrays=n_arr, state_bytes=n_arr * n0 * plane, hbm_bytes=avail, n_gpus=int(n_gpus), n_slabs=1,
               chunk_rows=None, k=None, host_bytes_per_rank=None, min_slabs_in_core=None)
    if avail is None:                       # no GPU visible: nothing to plan against, say what would be needed
        out.update(mode="in-core", bytes_per_gpu=out["state_bytes"], why="no GPU visible: HBM need only")
        return out

    def slab_bytes(s):                      # tallest slab of an s-way split, with its halo rows
        rows = -(-n0 // s) + (2 if s > 1 else 0)
        return n_arr * rows * plane

    max_slabs = max(1, min(int(n_gpus), n0))
    for s in range(1, n0 + 1):
        if slab_bytes(s) <= HEADROOM * avail:
            out["min_slabs_in_core"] = s
            break
    if slab_bytes(1) <= HEADROOM * avail and n_gpus <= 1:
        out.update(mode="in-core", bytes_per_gpu=slab_bytes(1), why=f"{n_arr} arrays fit in {HEADROOM:.0%} of the HBM")
        return out
    if n_gpus > 1 and slab_bytes(max_slabs) <= HEADROOM * avail:
        out.update(mode="slabs", n_slabs=max_slabs, bytes_per_gpu=slab_bytes(max_slabs),
                   why=f"one slab per GPU ({-(-n0 // max_slabs)} rows + halo rows) fits; fewest slabs that would: "
                       f"{out['min_slabs_in_core']}")
        return out
    # streamed from pinned host memory: deepest temporal blocking whose windows fit
    s = max_slabs
    rows_own = -(-n0 // s)
    budget = int(STAGING_FRACTION * avail // plane)            # planes of HBM the rings, boxes and resident rows may take
    n_state = 2 if FISTA else 1
    n_in, n_out, moved = 2 + nd * n_state, 1 + nd * n_state, 3 + nd * (n_state + 1)
    host_arrays = 2 + nd * n_state                             # data term, recon, accumulator state: updated in place
    if s == 1:
        # One GPU: depth first.  PCIe traffic per iteration falls as 1/k, and a streamed run is PCIe-bound until k ~ 100 (measured
        # on config-2 planes: k 32 -> 36, k 64 -> 55-58, k 128 -> 60 Gvoxel-iters/s).  For every chunk height the deepest k whose
        # level windows fit is taken (the need is linear in k); among those the deepest wins, taller chunks on ties.  (This
        # decides resident against streamed; the streamed run's own shape is then the library's: tvdn_stream_plan.)
        best = None
        for rows in (32, 16, 8, 4, 2):
            rows = min(rows, max(2, rows_own))
            slope = wavefront_windows(nd, rows, 2) - wavefront_windows(nd, rows, 1)
            k = (budget - wavefront_windows(nd, rows, 0)) // slope if slope > 0 else 0
            k = int(min(k, 1 if stop else MAX_DEPTH, max(1, rows_own)))     # a stopping rule decides after every iteration
            if k >= 1 and wavefront_windows(nd, rows, k) <= budget and (best is None or k > best[0]):
                best = (k, rows)
        if best is not None:
            k, rows = best
            out.update(mode="wavefront", n_slabs=1, chunk_rows=rows, k=k, resident_rows_per_rank=0,
                       bytes_per_gpu=wavefront_windows(nd, rows, k) * plane,
                       host_bytes_per_rank=host_arrays * (rows_own + 2 * k) * plane,   # in-place host state
                       why=f"state of {out['state_bytes'] / 2 ** 30:.1f} GiB exceeds {avail / 2 ** 30:.1f} GiB of HBM: "
                           f"streamed from pinned host memory")
            # What the library itself would do with this much HBM on a long run (tvdn_stream_plan, pure arithmetic): it may keep
            # rows in HBM between the passes -- all of them, swept in place, where ten arrays and a few rings fit (a cube up to
            # 1.36 x what fits resident) -- and then page-locks only the rows it streams.
            try:
                import ctypes as C
                from . import _lib
                a = _lib.RunArgs(dtype=_lib.dtype_code(np.dtype(dtype)), ndim=nd, bc_mode=2, device=int(device), n_fista=1024 if FISTA else 0,
                                 n_plain=0 if FISTA else 1024, use_stop=int(bool(stop)), stream_rows=-1, stream_k=-1, stream_resident=-1)
                for i, v in enumerate(shape):
                    a.shape[i] = int(v)
                po = _lib.StreamPlanOut()
                if _lib.lib().tvdn_stream_plan(C.byref(a), int(avail), C.byref(po)) == 0:
                    out.update(chunk_rows=int(po.rows), k=int(po.k), resident_rows_per_rank=int(po.resident_rows),
                               bytes_per_gpu=int(po.hbm_bytes), host_bytes_per_rank=host_arrays * (n0 - int(po.resident_rows)) * plane)
                    if po.resident_rows >= n0:
                        out["why"] += ": every row kept in HBM between the passes and swept in place (10 arrays + rings), nothing page-locked"
                    elif po.resident_rows > 0:
                        out["why"] += f", {int(po.resident_rows)} of {n0} rows kept in HBM between the passes"
            except Exception:
                pass            # (no library: the arithmetic above stands)
            if host_bytes is not None and out["host_bytes_per_rank"] > host_bytes:
                out["why"] += " (WARNING: the pinned host state does not fit in the host memory given)"
            return out
    else:
        # One slab per GPU, every rank streaming its own (denoise_slabs(staged=(rows, k)) = tvdn_run with a tvdn_slab_io).  A rank
        # page-locks host_arrays x (own rows + 2 k halo rows - rows it keeps resident in HBM); interior rows -- none of the k
        # at a face shared with a neighbour -- stay resident as far as the HBM beside the rings allows.  The fastest (rows, k)
        # by the library's model (csrc/tvdn_stream.hip choose_stream_shape: a row crosses the busy link in max(up / 55,
        # down / 48 GB/s: both by the DMA engines), 68 GB/s both ways when rows are kept; ring sweeps at 0.82 x 5.6 TB/s, 0.77 x in one-row chunks;
        # device copies at 7.1 TB/s) AMONG those whose page-locked state fits 80 % of the host memory n_gpus ranks share.
        # What the ranks hold is what distributed._check_hosts_hold_the_slabs will add up before any of them pins: the
        # page-locked arrays, the slab and its result as the caller holds them, the rows of a swap that goes through host memory.
        host = host_bytes if host_bytes is not None else host_available()
        host_cap = None if host is None else HOST_FRACTION * host / s

        def beside(k):
            return (2 * rows_own + (2 * k * (1 + nd * n_state) if swap_through_host else 0)) * plane
        rb = float(plane)
        row_step = max(n_in * rb / 55e9, n_out * rb / 48e9)
        best = fallback = None
        for rows in (2, 1) if not stop else (2,):
            for k in range(1, (1 if stop else min(MAX_DEPTH, rows_own)) + 1):
                planes = wavefront_windows(nd, rows, k)
                if planes > budget:
                    break
                interior = max(0, rows_own - 2 * k)
                for res in ({0, min(interior, (budget - planes) // n_in)} if rows > 1 else {0}):
                    streamed = rows_own - res
                    # (rows kept: csrc/tvdn_stream_plan.hip choose_stream_shape (b) -- 68 GB/s over the link both ways together;
                    #  beside the sweeps a kept row's store <-> ring copies at 7.1 TB/s and 10 ms per streamed row of 256 MiB
                    #  planes whose transfers are hidden)
                    link = streamed * (n_in + n_out) * rb
                    t_pcie = link / 68e9 if res else (streamed + 0.5 * min(k, streamed)) * row_step
                    t_gpu = rows_own * k * moved * rb / (5.6e12 * (0.80 if rows == 1 else 0.82))
                    if res:
                        t_gpu += res * (2 * host_arrays - 1) * 2 * rb / 7.1e12 + link / 460e9
                        if streamed:
                            t_gpu += 0.2 * (n_in * rb / 55e9 + n_out * rb / 42.5e9)
                    cand = (max(t_pcie, t_gpu) / k, k, rows, res, host_arrays * (rows_own + 2 * k - res) * plane)
                    if fallback is None or cand[4] < fallback[4]:
                        fallback = cand                                  # the plan that page-locks least, should none fit
                    if (host_cap is None or cand[4] + beside(k) <= host_cap) and (best is None or cand[0] < best[0]):
                        best = cand
        fits_host = best is not None
        if best is None:
            best = fallback
        if best is not None:
            t, k, rows, res, host_need = best
            out.update(mode="slabs+wavefront", n_slabs=s, chunk_rows=rows, k=k, resident_rows_per_rank=int(res),
                       bytes_per_gpu=(wavefront_windows(nd, rows, k) + res * n_in) * plane, host_bytes_per_rank=int(host_need),
                       host_beside_per_rank=int(beside(k)),
                       seconds_per_iteration_model=t,
                       why=f"state of {out['state_bytes'] / 2 ** 30:.1f} GiB exceeds {s} x {avail / 2 ** 30:.1f} GiB of HBM: every "
                           f"rank streams its slab from pinned host memory, {res} of its {rows_own} rows resident in HBM")
            if not fits_host:
                out["why"] += " (WARNING: not even the plan that page-locks least fits the host memory these ranks share)"
            return out
    out.update(mode="does-not-fit", bytes_per_gpu=wavefront_windows(nd, 2, 1) * plane,
               why="not even a 2-row chunk window fits in HBM")
    return out


__all__ = ["plan_run", "hbm_available", "host_available", "check_host_fits", "state_arrays", "wavefront_windows"]
