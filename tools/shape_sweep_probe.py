#!/usr/bin/env python3
"""The fused sweep by shape: kernel time (HIP events, mean of 24 sweeps after 6) of FISTA f32 on cubes whose extents are / are not
multiples of what the kernel likes (16-byte packs along the last axis, 256-thread tiles, 8-row marches).  One JSON line per shape."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cytvdn_amd import _lib
from cytvdn_amd.engine import HipBackend, SlabLayout

shapes = sys.argv[1:] or ["128x128x512", "128x128x513", "128x128x510", "127x129x512", "128x128x500", "100x100x1000", "100x100x1024",
                          "64x64x124x124", "64x64x125x125", "64x64x126x126", "63x65x124x124", "64x64x128x128", "61x67x128x128", "64x64x96x96", "64x64x100x100"]
L = _lib.lib()
for sh in shapes:
    shape = tuple(int(v) for v in sh.split("x"))
    nd = len(shape)
    be = HipBackend(SlabLayout(shape, 0, 1, 2), np.float32, True, device=0, max_iters=1)
    for q in range(nd):
        be._args.clip[q], be._args.lambda_mu[q] = 1.0, 1.0 / 32.0
    be.orig.zero_(); be.recon[0].zero_()
    for i in range(6):
        be.step(0.5, 0)
    n = 24
    _lib.check(L.tvdn_ctx_timing_enable(be.ctx, 1))
    for i in range(n):
        be.step(0.5, 0)
    torch.cuda.synchronize()
    each = (C.c_double * (n + 4))(); nl = C.c_int64()
    _lib.check(L.tvdn_ctx_timing_read_each(be.ctx, each, n + 4, C.byref(nl)))
    _lib.check(L.tvdn_ctx_timing_enable(be.ctx, 0))
    ms = float(np.mean(each[:nl.value]))
    vox = float(np.prod(shape))
    moved = (15 if nd == 4 else 12) * 4
    print(json.dumps({"shape": shape, "kernel_ms": round(ms, 4), "Gvoxel_iters_per_s": round(vox / ms / 1e6, 2), "moved_TBps": round(vox * moved / ms / 1e9, 3)}), flush=True)
    del be
    torch.cuda.empty_cache()
