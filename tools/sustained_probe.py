#!/usr/bin/env python3
"""Time series of the config-2 sweep kernel over a long run: one JSON line per block of steps (wall-clock offset, mean /
min / max kernel time from HIP events), to lay beside tools/clocks_during.py's samples of the same seconds.

    python tools/clocks_during.py gpurun_out/clocks.txt -- python tools/sustained_probe.py --seconds 60 > series.jsonl
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--block", type=int, default=50)
    ap.add_argument("--shape", default="256x256x128x128")
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--plain", action="store_true")
    ap.add_argument("--idle", type=float, default=0.0, help="seconds to sleep between blocks (does a pause reset the state?)")
    a = ap.parse_args()
    import numpy as np
    import torch
    from cytvdn_amd import _lib, synth
    from cytvdn_amd.engine import HipBackend, SlabLayout, fista_ratios
    shape = tuple(int(v) for v in a.shape.split("x"))
    nd = len(shape)
    dtype = np.float32 if a.dtype == "f32" else np.float64
    fista = not a.plain
    t_start = time.time()
    be = HipBackend(SlabLayout(shape, 0, 1, 2), dtype, fista, device=0, max_iters=a.block)
    mu = np.array([1.0, 1.0, 0.5, 0.5][:nd] if nd == 4 else [1.0, 1.0, 0.5], dtype)
    lam = mu / dtype(32.0 if nd == 4 else 16.0)
    be.set_params(1.0 / lam, (lam / mu).astype(dtype))
    _lib.check(_lib.lib().tvdn_synth_fill(be.code, nd, _lib.shape_arr(shape), synth.SEED_4D if nd == 4 else synth.SEED_3D,
                                          0, shape[0], be.orig.data_ptr(), _lib.current_stream(0)))
    be.recon[be.cur].copy_(be.orig)
    ratios = fista_ratios(4096)
    torch.cuda.synchronize()
    print(json.dumps({"t": round(time.time() - t_start, 2), "event": "state ready", "shape": shape}), flush=True)
    it = 0
    each = (C.c_double * (a.block + 8))()
    nl = C.c_int64()
    while time.time() - t_start < a.seconds:
        _lib.check(_lib.lib().tvdn_ctx_timing_enable(be.ctx, 1))
        t0 = time.time()
        for i in range(a.block):
            be.step(float(ratios[min(it, 4095)]) if fista else None, i)
            it += 1
        torch.cuda.synchronize()
        _lib.check(_lib.lib().tvdn_ctx_timing_read_each(be.ctx, each, a.block + 8, C.byref(nl)))
        v = np.array(each[:nl.value])
        print(json.dumps({"t": round(t0 - t_start, 2), "steps": int(nl.value), "mean_ms": round(float(v.mean()), 4),
                          "min_ms": round(float(v.min()), 4), "max_ms": round(float(v.max()), 4)}), flush=True)
        if a.idle:
            time.sleep(a.idle)


if __name__ == "__main__":
    main()
