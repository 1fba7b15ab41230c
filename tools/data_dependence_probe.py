#!/usr/bin/env python3
"""Does the sweep's time depend on the VALUES it moves?  One state of BASELINE config 2 on one allocation, swept (a) all zeros (what
the placement audition times), (b) from the synthetic cube, iteration by iteration, (c) all zeros again, (d) from white noise.
The kernel is branch-free, so a difference is the memory system's (or the clocks').  One JSON line."""
import ctypes as C
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cytvdn_amd import _lib, synth
from cytvdn_amd.engine import HipBackend, SlabLayout, SlabRunner, fista_ratios

shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "256x256x128x128").split("x"))
nd = len(shape)
dtype = np.float32
lay = SlabLayout(shape, 0, 1, 2)
be = HipBackend(lay, dtype, True, device=0, max_iters=64)
mu = np.array([1.0, 1.0, 0.5, 0.5][:nd] if nd == 4 else [1.0, 1.0, 0.5], dtype)
lam = mu / dtype(32.0 if nd == 4 else 16.0)
L = _lib.lib()


def timed(n, first_slot=0):
    """per-iteration sweep-kernel ms of n FISTA iterations continuing from the current state"""
    runner = SlabRunner(be)
    ratios = fista_ratios(first_slot + n)
    _lib.check(L.tvdn_ctx_timing_enable(be.ctx, 1))
    for i in range(first_slot, first_slot + n):
        runner._step(float(ratios[i]), i)
    runner.finish()
    torch.cuda.synchronize()
    each = (C.c_double * (n + 8))()
    nl = C.c_int64()
    _lib.check(L.tvdn_ctx_timing_read_each(be.ctx, each, n + 8, C.byref(nl)))
    _lib.check(L.tvdn_ctx_timing_enable(be.ctx, 0))
    return [round(float(v), 4) for v in each[:nl.value]]


out = {"shape": shape, "state_mem": be.state_mem}
out["zeros_1"] = round(be.probe_ms(8), 4)
be.set_params(1.0 / lam, (lam / mu).astype(dtype))
seed = synth.SEED_4D if nd == 4 else synth.SEED_3D
_lib.check(L.tvdn_synth_fill(be.code, nd, _lib.shape_arr(shape), seed, 0, shape[0], be.orig.data_ptr(), _lib.current_stream(0)))
be.recon[be.cur].copy_(be.orig)
out["synthetic_by_iteration"] = timed(40)
out["zeros_2"] = round(be.probe_ms(8), 4)
be.set_params(1.0 / lam, (lam / mu).astype(dtype))
be.orig.normal_(mean=20.0, std=5.0)
be.recon[be.cur].copy_(be.orig)
out["white_noise_by_iteration"] = timed(40)
out["zeros_3"] = round(be.probe_ms(8), 4)
print(json.dumps(out), flush=True)
