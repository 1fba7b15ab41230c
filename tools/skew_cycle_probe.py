#!/usr/bin/env python3
"""The stagger between the state's arrays against a WHOLE cycle of their roles (round 6: which arrays are written decides a sweep's
time by +-1.5 %, period 6 -- tools/layout_probe.py of round 3 took the best of three sweeps and could not see that).  One block on
granules per allocation, every stagger carved from the same pages, alternating rounds, 12 timed sweeps each (two cycles): mean and
the twelve times.  One JSON line per (allocation, stagger)."""
import argparse, ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cytvdn_amd import _lib
from cytvdn_amd.engine import HipBackend, SlabLayout

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="256x256x128x128")
ap.add_argument("--dtype", default="float32")
ap.add_argument("--plain", action="store_true")
ap.add_argument("--allocs", type=int, default=2)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--skews", default="4096,0,256,1024,2048,8192,12288,16384,20480,36864,69632,135168,266240,1052672,2101248,4198400")
a = ap.parse_args()
shape = tuple(int(v) for v in a.shape.split("x"))
dt = np.dtype(a.dtype)
tdt = torch.float32 if dt == np.float32 else torch.float64
fista = not a.plain
nd = len(shape)
skews = [int(v) for v in a.skews.split(",")]
n_arr = 3 + nd * (3 if fista else 2)
n_el = int(np.prod(shape))
need = n_arr * (-(-(n_el * dt.itemsize) // 256) * 256 + max(skews)) + 4096
lay = SlabLayout(shape, 0, 1, 2)
L = _lib.lib()
for alloc in range(a.allocs):
    blk = _lib.DeviceBlock(need, 0)
    big = blk.tensor(tdt)
    res = {s: [] for s in skews}
    for rnd in range(a.rounds):
        for skew in skews:
            os.environ["TVDN_ARRAY_SKEW"] = str(skew)
            be = HipBackend(lay, dt, fista, device=0, max_iters=1, slab=big)
            n = 13
            for q in range(nd):
                be._args.clip[q], be._args.lambda_mu[q] = 1.0, 1.0 / 32.0
            be.orig.zero_(); be.recon[0].zero_()
            _lib.check(L.tvdn_ctx_timing_enable(be.ctx, 1))
            for i in range(n):
                be.step(0.5 if fista else None, 0)
            torch.cuda.synchronize()
            each = (C.c_double * (n + 4))(); nl = C.c_int64()
            _lib.check(L.tvdn_ctx_timing_read_each(be.ctx, each, n + 4, C.byref(nl)))
            _lib.check(L.tvdn_ctx_timing_enable(be.ctx, 0))
            res[skew].append([round(float(v), 3) for v in each[1:nl.value]])
            del be
    for skew, runs in res.items():
        means = [round(float(np.mean(r)), 4) for r in runs]
        print(json.dumps({"alloc": alloc, "kind": blk.kind, "skew": skew, "cycle_mean_ms": means, "mean": round(float(np.mean(means)), 4), "sweeps": runs[0]}), flush=True)
    del big
    blk.free()
