#!/usr/bin/env python3
"""Granule size of the state's block, A/B in ONE process: one block per size held side by side (each a random subset of its own
pool, as the product draws it), the config-2 state carved from each, swept in alternating rounds of 12 sweeps (two cycles of the
roles).  Fresh processes compare different boxes' memory states (a block released a moment ago is still being cleared); this does not."""
import argparse, ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cytvdn_amd import _lib
from cytvdn_amd.engine import HipBackend, SlabLayout

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="256x256x128x128")
ap.add_argument("--sizes", default="1024,256,64")
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--plain", action="store_true")
a = ap.parse_args()
shape = tuple(int(v) for v in a.shape.split("x"))
nd = len(shape)
fista = not a.plain
n_arr = 3 + nd * (3 if fista else 2)
need = n_arr * (int(np.prod(shape)) * 4 + 4096 + 256)
lay = SlabLayout(shape, 0, 1, 2)
L = _lib.lib()
blocks = {}
for g in [int(v) for v in a.sizes.split(",")]:
    os.environ["TVDN_GRANULE_MIB"] = str(g)
    blocks[g] = _lib.DeviceBlock(need, 0)
os.environ.pop("TVDN_GRANULE_MIB")
res = {g: [] for g in blocks}
for rnd in range(a.rounds):
    for g, blk in blocks.items():
        be = HipBackend(lay, np.float32, fista, device=0, max_iters=1, slab=blk.tensor(torch.float32))
        for q in range(nd):
            be._args.clip[q], be._args.lambda_mu[q] = 1.0, 1.0 / 32.0
        be.orig.zero_(); be.recon[0].zero_()
        n = 13
        _lib.check(L.tvdn_ctx_timing_enable(be.ctx, 1))
        for i in range(n):
            be.step(0.5 if fista else None, 0)
        torch.cuda.synchronize()
        each = (C.c_double * (n + 4))(); nl = C.c_int64()
        _lib.check(L.tvdn_ctx_timing_read_each(be.ctx, each, n + 4, C.byref(nl)))
        _lib.check(L.tvdn_ctx_timing_enable(be.ctx, 0))
        res[g].append(round(float(np.mean(each[1:nl.value])), 4))
        del be
for g in blocks:
    print(json.dumps({"shape": shape, "granule_MiB": g, "kind": blocks[g].kind, "round_means_ms": res[g], "mean_ms": round(float(np.mean(res[g])), 4)}), flush=True)
