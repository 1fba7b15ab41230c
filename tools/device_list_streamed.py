#!/usr/bin/env python3
"""BASELINE configs[4] in structure inside ONE process: tvdn_run with a device list and stream_rows / stream_k -- every slab
streamed through its own GPU from page-locked host arrays the slabs share, no launcher, no messages.  One JSON line.

    python tools/device_list_streamed.py --shape 256x256x128x128 --devices 0,1,2,3,4,5,6,7 --rows 8 --k 24 --iters 48 [--check]

On a one-GPU box the same device may be named several times (--devices 0,0,0: a rehearsal of the schedule, not of the rate).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="256x256x128x128")
    ap.add_argument("--devices", default="0,0")
    ap.add_argument("--rows", type=int, default=8)
    ap.add_argument("--k", type=int, default=24)
    ap.add_argument("--iters", type=int, default=48)
    ap.add_argument("--check", action="store_true", help="also run on the first device alone and compare bit for bit (the cube must fit it)")
    a = ap.parse_args()
    import numpy as np
    from cytvdn_amd import _lib
    shape = tuple(int(v) for v in a.shape.split("x"))
    devs = [int(v) for v in a.devices.split(",")]
    nd = len(shape)
    x = bench.synth_host(shape, devs[0])
    mu = np.array([1.0, 1.0, 0.5, 0.5][:nd] if nd == 4 else [1.0, 1.0, 0.5], np.float32)
    lam = mu / np.float32(32.0 if nd == 4 else 16.0)

    def run(devices, stream):
        ra = _lib.RunArgs(dtype=0, ndim=nd, bc_mode=2, device=devices[0], n_fista=a.iters, n_plain=0, stream_rows=stream[0],
                          stream_k=stream[1], n_devices=len(devices) if len(devices) > 1 else 0)
        for i, d in enumerate(devices):
            ra.devices[i] = d
        for i, s in enumerate(shape):
            ra.shape[i] = s
        for q in range(nd):
            ra.clip[q] = float((1.0 / lam)[q])
            ra.lambda_mu[q] = float((lam / mu).astype(np.float32)[q])
        recon, sums, st = np.empty_like(x), np.zeros((a.iters, 3)), _lib.RunStats()
        ra.data, ra.recon_out, ra.sums_out, ra.stats = x.ctypes.data, recon.ctypes.data, sums.ctypes.data, C.addressof(st)
        t0 = time.perf_counter()
        _lib.check(_lib.lib().tvdn_run(C.byref(ra)))
        return recon, sums, st, time.perf_counter() - t0

    recon, sums, st, whole = run(devs, (a.rows, a.k))
    vox = float(np.prod(shape))
    out = {"metric": "Gvoxel-iters/s (4D aniso FISTA, streamed device list in one process)", "value": round(vox * a.iters / st.loop_s / 1e9, 3),
           "unit": "Gvoxel-iters/s", "value_whole_call": round(vox * a.iters / whole / 1e9, 3), "shape": list(shape), "devices": devs,
           "distinct_devices": len(set(devs)), "stream_rows": st.stream_rows, "stream_k": st.stream_k, "passes": st.n_passes, "iters": a.iters,
           "passes_s": round(st.loop_s, 3), "setup_s": round(st.setup_s, 3), "whole_call_s": round(whole, 3),
           "h2d_GBps_all": round(st.h2d_bytes / st.loop_s / 1e9, 2), "d2h_GBps_all": round(st.d2h_bytes / st.loop_s / 1e9, 2),
           "b_norm_last": float(sums[-1, 0])}
    if a.check:
        want = run(devs[:1], (0, 0))[0]
        out["bit_identical_to_one_device_resident"] = bool(want.tobytes() == recon.tobytes())
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
