#!/usr/bin/env python3
"""Config-5-style measurement on ONE GPU: a cube in host memory, streamed through the GPU by the C entry point
(tvdn_run with stream_rows / stream_k, csrc/tvdn_stream.hip).  Reported separately from bench.py's headline: PCIe-bound.
tools/stream_rates.py is the round-4 form of this measurement (run statistics from the library instead of a parsed line).

    python tools/bench_outofcore.py --shape 128x256x128x128 --rows 64 --k 16 --iters 32
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _default_host_limit():
    """Unless the caller says otherwise, a measurement may page-lock at most half of what the host has available
    (TVDN_HOST_LIMIT is honoured by the streamed runs): a mistyped shape gets an error, not the box."""
    if "TVDN_HOST_LIMIT" in os.environ:
        return
    try:
        with open("/proc/meminfo") as f:
            kb = next(int(line.split()[1]) for line in f if line.startswith("MemAvailable:"))
        os.environ["TVDN_HOST_LIMIT"] = str(kb * 1024 // 2)
    except (OSError, StopIteration, ValueError):
        os.environ["TVDN_HOST_LIMIT"] = "32G"


_default_host_limit()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="128x256x128x128")
    ap.add_argument("--rows", type=int, default=64)
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--iters", type=int, default=32)
    ap.add_argument("--engine", default="native", choices=["native"], help="kept for old command lines: there is one engine")
    ap.add_argument("--check", action="store_true", help="also run in-core and compare bit for bit")
    a = ap.parse_args()
    import numpy as np
    import torch
    from cytvdn_amd import _lib, synth
    shape = tuple(int(v) for v in a.shape.split("x"))
    nd = len(shape)
    dt = np.dtype(np.float32)
    _lib.ctx(0)
    # synthesise on the device in slices, land in host memory
    t0 = time.perf_counter()
    x = np.empty(shape, dt)
    step = max(1, (1 << 28) // int(np.prod(shape[1:])))
    buf = torch.empty((step,) + shape[1:], dtype=torch.float32, device="cuda")
    for r in range(0, shape[0], step):
        n = min(step, shape[0] - r)
        _lib.check(_lib.lib().tvdn_synth_fill(0, nd, _lib.shape_arr(shape), synth.SEED_4D if nd == 4 else synth.SEED_3D,
                                              r, n, buf.data_ptr(), _lib.current_stream(0)))
        x[r:r + n] = buf[:n].cpu().numpy()
    t_syn = time.perf_counter() - t0
    mu = np.array([1.0, 1.0, 0.5, 0.5][:nd] if nd == 4 else [1.0, 1.0, 0.5], dt)
    lam = mu / dt.type(32.0 if nd == 4 else 16.0)
    if True:
        # the C entry point (tvdn_run, stream_rows / stream_k): page-locks x and the result in place, allocates the
        # pinned state, streams, returns.  TVDN_STREAM_TIMING makes the library report set-up and passes apart (stderr).
        import ctypes as C
        import re
        import tempfile
        recon = np.empty_like(x)
        ra = _lib.RunArgs(dtype=0, ndim=nd, bc_mode=2, device=0, n_fista=a.iters, n_plain=0, stream_rows=a.rows, stream_k=a.k)
        for i, s in enumerate(shape):
            ra.shape[i] = s
        for q in range(nd):
            ra.clip[q] = float((1.0 / lam)[q])
            ra.lambda_mu[q] = float((lam / mu).astype(dt)[q])
        sums = np.zeros((a.iters, 3))
        ra.data, ra.recon_out, ra.sums_out = x.ctypes.data, recon.ctypes.data, sums.ctypes.data
        os.environ["TVDN_STREAM_TIMING"] = "1"
        with tempfile.TemporaryFile(mode="w+") as tf:      # the library writes its timing line to fd 2
            saved = os.dup(2)
            os.dup2(tf.fileno(), 2)
            try:
                t0 = time.perf_counter()
                _lib.check(_lib.lib().tvdn_run(C.byref(ra)))
                whole = time.perf_counter() - t0
            finally:
                os.dup2(saved, 2)
                os.close(saved)
            tf.seek(0)
            m = re.search(r"set-up ([0-9.]+) s, passes ([0-9.]+) s", tf.read())
        setup_s, pass_s = (float(m.group(1)), float(m.group(2))) if m else (None, whole)
        vox = float(np.prod(shape))
        out = {"metric": "Gvoxel-iters/s (4D aniso FISTA, out-of-core single GPU)", "value": round(vox * a.iters / pass_s / 1e9, 3),
               "unit": "Gvoxel-iters/s", "shape": list(shape), "block_rows": a.rows, "iters_per_pass": a.k, "iters": a.iters,
               "engine": "native (tvdn_run, csrc/tvdn_stream.hip)", "seconds": round(pass_s, 3), "setup_s": setup_s,
               "whole_call_s": round(whole, 3), "h2d_GBps": None, "d2h_GBps": None, "b_norm_last": float(sums[-1, 0])}
        if a.check:
            import cytvdn_amd as tv
            want = tv.denoise4D(x, mu, a.iters, quiet=True)[0]
            out["bit_identical_to_in_core"] = bool(want.tobytes() == recon.tobytes())
        print(json.dumps(out))
        return


if __name__ == "__main__":
    main()
