#!/usr/bin/env python3
"""Timing of the one-pass (kernel-level) entry points on device-resident arrays (measurement aid)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cytvdn_amd import _lib

shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "256x256x128x128").split("x"))
nd = len(shape)
L, ctx, st = _lib.lib(), _lib.ctx(0), _lib.current_stream(0)
for dtn, tdt, code in (("f32", torch.float32, 0), ("f64", torch.float64, 1)):
    a, b, d, o = (torch.rand(shape, dtype=tdt, device="cuda") for _ in range(4))
    bs = [torch.rand(shape, dtype=tdt, device="cuda") for _ in range(nd)]
    out = torch.zeros(4, dtype=torch.float64, device="cuda")
    item = 4 if code == 0 else 8
    n = float(np.prod(shape))
    def timeit(fn, passes, label):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(json.dumps({"op": label, "dtype": dtn, "ms": round(ms, 3), "GBps": round(passes * n * item / ms / 1e6, 1)}))
    sh = _lib.shape_arr(shape)
    for ax in (0, nd - 1):
        timeit(lambda: _lib.check(L.tvdn_accumulator_update(ctx, code, nd, sh, a.data_ptr(), b.data_ptr(), d.data_ptr(), 0.3, ax, 0.9, 2, out.data_ptr(), st)), 5, f"accumulator_update FISTA ax={ax}")
    timeit(lambda: _lib.check(L.tvdn_accumulator_update(ctx, code, nd, sh, a.data_ptr(), b.data_ptr(), None, 0.0, 1, 0.9, 2, out.data_ptr(), st)), 3, "accumulator_update plain ax=1")
    bp = (C.c_void_p * nd)(*[x.data_ptr() for x in bs]); lm = (C.c_double * nd)(*([0.03] * nd))
    timeit(lambda: _lib.check(L.tvdn_datacube_update(ctx, code, nd, sh, o.data_ptr(), a.data_ptr(), bp, lm, 2, out.data_ptr(), st)), nd + 3, "datacube_update")
    timeit(lambda: _lib.check(L.tvdn_sum_square_error(ctx, code, nd, sh, a.data_ptr(), b.data_ptr(), out.data_ptr(), st)), 2, "sum_square_error")
    del a, b, d, o, bs
    torch.cuda.empty_cache()
