#!/usr/bin/env python3
"""Timing of the one-pass (kernel-level) entry points on device-resident arrays (measurement aid)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cytvdn_amd import _lib

shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "256x256x128x128").split("x"))
nd = len(shape)
L, ctx, st = _lib.lib(), _lib.ctx(0), _lib.current_stream(0)
# TVDN_LIB_B=<path of another build>: every op is timed with both libraries alternately on the SAME arrays (the one-pass
# kernels' speed depends on the placement of their arrays as the fused sweep's does: only such pairs are comparable)
LIBS = [("", L, ctx)]
if os.environ.get("TVDN_LIB_B"):
    LB = C.CDLL(os.environ["TVDN_LIB_B"])
    LB.tvdn_ctx_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    for name in ("tvdn_accumulator_update", "tvdn_datacube_update", "tvdn_sum_square_error"):
        getattr(LB, name).argtypes = getattr(L, name).argtypes
    hb = C.c_void_p()
    assert LB.tvdn_ctx_create(C.byref(hb), 0) == 0
    LIBS.append((" [B]", LB, hb))
for dtn, tdt, code in (("f32", torch.float32, 0), ("f64", torch.float64, 1)):
    a, b, d, o = (torch.rand(shape, dtype=tdt, device="cuda") for _ in range(4))
    bs = [torch.rand(shape, dtype=tdt, device="cuda") for _ in range(nd)]
    out = torch.zeros(4, dtype=torch.float64, device="cuda")
    item = 4 if code == 0 else 8
    n = float(np.prod(shape))
    def timeit(make, passes, label):
        res = {tag: [] for tag, _, _ in LIBS}
        for rnd in range(3 if len(LIBS) > 1 else 1):
            for tag, Lx, cx in LIBS:
                fn = make(Lx, cx)
                fn(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): fn()
                e1.record(); torch.cuda.synchronize()
                res[tag].append(e0.elapsed_time(e1) / 5)
        for tag, v in res.items():
            ms = sum(v) / len(v)
            print(json.dumps({"op": label + tag, "dtype": dtn, "ms": round(ms, 3), "GBps": round(passes * n * item / ms / 1e6, 1)}))
    sh = _lib.shape_arr(shape)
    def chk(rc): assert rc == 0, rc
    for ax in (0, nd - 1):
        timeit(lambda Lx, cx: (lambda: chk(Lx.tvdn_accumulator_update(cx, code, nd, sh, a.data_ptr(), b.data_ptr(), d.data_ptr(), 0.3, ax, 0.9, 2, out.data_ptr(), st))), 5, f"accumulator_update FISTA ax={ax}")
    timeit(lambda Lx, cx: (lambda: chk(Lx.tvdn_accumulator_update(cx, code, nd, sh, a.data_ptr(), b.data_ptr(), None, 0.0, 1, 0.9, 2, out.data_ptr(), st))), 3, "accumulator_update plain ax=1")
    bp = (C.c_void_p * nd)(*[x.data_ptr() for x in bs]); lm = (C.c_double * nd)(*([0.03] * nd))
    timeit(lambda Lx, cx: (lambda: chk(Lx.tvdn_datacube_update(cx, code, nd, sh, o.data_ptr(), a.data_ptr(), bp, lm, 2, out.data_ptr(), st))), nd + 3, "datacube_update")
    timeit(lambda Lx, cx: (lambda: chk(Lx.tvdn_sum_square_error(cx, code, nd, sh, a.data_ptr(), b.data_ptr(), out.data_ptr(), st))), 2, "sum_square_error")
    del a, b, d, o, bs
    torch.cuda.empty_cache()
