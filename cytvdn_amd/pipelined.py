"""In-core runs from host memory: the first iterations under the upload, the last ones over the download.

`denoise3D/4D` take a NumPy array and return one (cyTVDN/cyTVDN.py:19-31, :244-247), so a call pays for 2 x the cube over
PCIe around its iterations: 2 x 4 GiB at ~50 GB/s = 0.17 s of the 0.75 s a 50-iteration denoise4D of BASELINE config 2
takes.  Both transfers can hide under iterations if those iterations do not wait for the whole cube:

  start   the cube goes up in chunks of R rows; when chunk c has arrived, iteration level j advances rows
          [c R - (j+1), (c+1) R - (j+1)) for j = 0 .. K-1 -- the wavefront of cytvdn_amd/wavefront.py (level j+1 trails
          level j by one row, every row of every level swept once), here on the RESIDENT arrays with partial sweeps
          (tvdn_iter_args.sweep_lo / sweep_hi) instead of rings: K iterations are done when the last chunk is in;
  end     the last K iterations run the same way, and every chunk's rows of the final level go home while the
          chunks after it are still being swept.

Two recon buffers and three rotating accumulator arrays per axis suffice for K levels in flight because level j+2, which
writes the buffer level j+1 reads, stays two rows behind it.  The sweeps at the cube's top face use TVDN_EDGE_ZERO (the
wrapped axis-0 accumulator of a Jia-Zhao run is identically zero while row 0 is finite; tvdn.h), since row 0 of the
buffers belongs to a later level by then: the caller (driver._run) takes this path only for Jia-Zhao runs whose first row
is finite.  The launches are tvdn_iterate_fused launches with the library's own role binding, so the bits are those of the
plain loop (tests/test_gpu_pipelined.py checks them against the CPU restatement of the reference)."""
from __future__ import annotations

import ctypes as C
import threading

import numpy as np
import torch

from . import _lib
from .engine import fista_ratios


def _levels(be, ratios):
    """Role snapshot per iteration of `ratios` (None = unaccelerated), and the snapshot after the last one."""
    m = _lib.ManyArgs()
    C.memmove(C.byref(m), C.byref(be._roles), C.sizeof(_lib.ManyArgs))
    snaps = []
    for tk in ratios:
        s = _lib.ManyArgs()
        C.memmove(C.byref(s), C.byref(m), C.sizeof(_lib.ManyArgs))
        snaps.append(s)
        _lib.check(_lib.lib().tvdn_roles_advance(C.byref(m), int(tk is not None), float(tk or 0.0)))
    return snaps, m


def _launch(be, snap, tk, slot, a, b):
    """Rows [a, b) of one iteration level with the roles of that level; sums accumulate in slot `slot`."""
    args = be._args
    _lib.check(_lib.lib().tvdn_roles_bind(C.byref(snap), int(tk is not None), float(tk or 0.0), C.byref(args)))
    args.sweep_lo, args.sweep_hi, args.accumulate = int(a), int(b), 1
    _lib.check(_lib.lib().tvdn_iterate_fused(be.ctx, C.byref(args), C.c_void_p(be.sums[slot].data_ptr()),
                                             _lib.current_stream(be.device)))


def _wavefront(be, ratios, slot0, n_chunks, R, before_chunk=None, after_chunk=None):
    """The K = len(ratios) levels over the whole cube, chunk by chunk; leaves the backend's roles after the last level."""
    N0 = be.layout.shape[0]
    snaps, after = _levels(be, ratios)
    hi_mode = be._args.hi_mode
    be._args.hi_mode = _lib.EDGE_ZERO
    try:
        for c in range(n_chunks):
            if before_chunk is not None:
                before_chunk(c)
            for j, tk in enumerate(ratios):
                a, b = max(0, c * R - (j + 1)), min(N0, (c + 1) * R - (j + 1))
                if a < b:
                    _launch(be, snaps[j], tk, slot0 + j, a, b)
            if after_chunk is not None:
                after_chunk(c)
    finally:
        be._args.hi_mode = hi_mode
    C.memmove(C.byref(be._roles), C.byref(after), C.sizeof(_lib.ManyArgs))


def plan(n_rows: int, n_total: int, cube_bytes: int):
    """(chunk rows, levels at the start, levels at the end), or None when the call is too small to gain."""
    if n_total < 4 or n_rows < 32 or cube_bytes < (256 << 20):
        return None
    R = max(8, -(-n_rows // 8))                  # eight chunks: ~12 ms of PCIe each for a 4 GiB cube
    K = min(8, n_total // 2)                     # as many iterations as one chunk's transfer pays for
    return R, K, min(8, n_total - K)


def run(be, runner, x: np.ndarray, n_fista: int, n_plain: int, R: int, k_start: int, k_end: int) -> np.ndarray:
    """n_fista FISTA iterations then n_plain unaccelerated ones on the (fresh) backend `be`, input `x` still on the host;
    returns the reconstruction as a new host array.  `runner` (engine.SlabRunner) runs the iterations in between."""
    import os
    import time
    timing = os.environ.get("TVDN_PIPE_TIMING")
    t0 = time.perf_counter()
    # Transfers that run beside the sweeps' launches: six staging lanes instead of eight.  With eight host threads copying
    # through their pinned buffers the launching thread falls behind and the overlap is lost (config 2, 50 iterations:
    # 0.72-0.76 s with 8 lanes = no gain, 0.62-0.64 s with 4-6; profiles/r03_e2e_pipelined.txt).  TVDN_IO_LANES overrides.
    lanes_set = "TVDN_IO_LANES" not in os.environ
    if lanes_set:
        os.environ["TVDN_IO_LANES"] = "6"
    try:
        return _run(be, runner, x, n_fista, n_plain, R, k_start, k_end, timing, t0)
    finally:
        if lanes_set:
            os.environ.pop("TVDN_IO_LANES", None)


def _run(be, runner, x, n_fista, n_plain, R, k_start, k_end, timing, t0):
    import time
    L = _lib.lib()
    N0 = x.shape[0]
    n_total = n_fista + n_plain
    ratios = [float(r) for r in fista_ratios(n_fista)] + [None] * n_plain
    dev = be.device
    main = torch.cuda.current_stream(dev)
    main.synchronize()
    row_bytes = x[0].nbytes
    n_up = -(-N0 // R)

    # ---- start: chunks go up on a helper thread (the library's pinned multi-lane staging is synchronous) -------------------
    arrived = [threading.Event() for _ in range(n_up)]
    err = []

    def uploader():
        try:
            for c in range(n_up):
                a, b = c * R, min((c + 1) * R, N0)
                _lib.check(L.tvdn_copy_to_device(C.c_void_p(be.orig[a:b].data_ptr()), C.c_void_p(x[a:b].ctypes.data),
                                                 (b - a) * row_bytes, dev))
                arrived[c].set()
        except Exception as e:      # pragma: no cover - surfaces in the main thread
            err.append(e)
            for ev in arrived:
                ev.set()

    th = threading.Thread(target=uploader, daemon=True)
    th.start()
    cur0 = be.cur

    waited = [0.0]

    def before(c):
        if c < n_up:
            tw = time.perf_counter()
            arrived[c].wait()
            waited[0] += time.perf_counter() - tw
            if err:
                raise err[0]
            a, b = c * R, min((c + 1) * R, N0)
            be.recon[cur0][a:b].copy_(be.orig[a:b], non_blocking=True)     # recon = datacube.copy() (cyTVDN.py:145)

    _wavefront(be, ratios[:k_start], 0, -(-(N0 + k_start) // R), R, before_chunk=before)
    th.join()
    if err:
        raise err[0]
    runner.ran.extend(range(k_start))
    runner.iter = k_start
    if timing:
        main.synchronize()
        t1 = time.perf_counter()

    # ---- middle: whole sweeps ------------------------------------------------------------------------------------------------
    mid = n_total - k_start - k_end
    mid_f = max(0, min(n_fista - k_start, mid))
    if mid > 0:
        runner.run(mid_f, mid - mid_f, None, first_fista=k_start if mid_f else 0)

    if timing:
        main.synchronize()
        t2 = time.perf_counter()
    # ---- end: the last levels as a wavefront, finished rows go home chunk by chunk ---------------------------------------------
    out = np.empty(x.shape, x.dtype)
    if k_end <= 0:
        main.synchronize()
        _lib.check(L.tvdn_copy_to_host(C.c_void_p(out.ctypes.data), C.c_void_p(be.recon_tensor().data_ptr()), out.nbytes, dev))
        return out
    import queue
    first = n_total - k_end
    final = be.recon[be.cur ^ (k_end % 2)]        # the buffer the last level writes (level j writes recon[cur ^ ((j+1) % 2)])
    jobs = queue.Queue()

    def downloader():
        try:
            while True:
                job = jobs.get()
                if job is None:
                    return
                ev, a, b = job
                ev.synchronize()                   # the last level has written rows [a, b)
                _lib.check(L.tvdn_copy_to_host(C.c_void_p(out[a:b].ctypes.data), C.c_void_p(final[a:b].data_ptr()),
                                               (b - a) * row_bytes, dev))
        except Exception as e:      # pragma: no cover - surfaces in the main thread
            err.append(e)

    th = threading.Thread(target=downloader, daemon=True)
    th.start()

    def after(c):
        a, b = max(0, c * R - k_end), min(N0, (c + 1) * R - k_end)
        if a < b:
            ev = torch.cuda.Event()
            ev.record(main)
            jobs.put((ev, a, b))

    try:
        _wavefront(be, ratios[first:], first, -(-(N0 + k_end) // R), R, after_chunk=after)
    finally:
        jobs.put(None)
        th.join()
    if err:
        raise err[0]
    runner.ran.extend(range(first, n_total))
    runner.iter = n_total
    main.synchronize()
    if timing:
        import sys
        t3 = time.perf_counter()
        print(f"pipelined: start ({k_start} levels under the upload) {t1 - t0:.3f} s (of which {waited[0]:.3f} s waiting for rows), middle {t2 - t1:.3f} s, "
              f"end ({k_end} levels over the download) {t3 - t2:.3f} s", file=sys.stderr)
    return out
