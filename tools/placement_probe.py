#!/usr/bin/env python3
"""Does the config-2 sweep time depend on where its 60 GiB of state land?  Several allocations in ONE process, with and
without churning the allocator in between; prints the state's base address and the mean sweep-kernel time of 10 steps."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cytvdn_amd import _lib, synth
from cytvdn_amd.engine import HipBackend, SlabLayout, SlabRunner, fista_ratios

shape = (256, 256, 128, 128)
dt = np.float32
mu = np.array([1, 1, .5, .5], dt); lam = mu / dt(32)


def once(tag):
    be = HipBackend(SlabLayout(shape, 0, 1, 2), dt, True, device=0, max_iters=16)
    be.set_params(1.0 / lam, (lam / mu).astype(dt))
    _lib.check(_lib.lib().tvdn_synth_fill(0, 4, _lib.shape_arr(shape), synth.SEED_4D, 0, shape[0], be.orig.data_ptr(), _lib.current_stream(0)))
    be.recon[be.cur].copy_(be.orig)
    r = fista_ratios(16)
    for i in range(3):
        be.step(float(r[i]), i)
    torch.cuda.synchronize()
    _lib.check(_lib.lib().tvdn_ctx_timing_enable(be.ctx, 1))
    for i in range(3, 13):
        be.step(float(r[i]), i)
    torch.cuda.synchronize()
    ms, n = C.c_double(), C.c_int64()
    _lib.check(_lib.lib().tvdn_ctx_timing_read(be.ctx, C.byref(ms), C.byref(n)))
    _lib.check(_lib.lib().tvdn_ctx_timing_enable(be.ctx, 0))
    base = be._slab.data_ptr()
    print(json.dumps({"tag": tag, "kernel_ms": round(ms.value / n.value, 4), "base": hex(base), "base_mod_1GiB_MiB": (base % (1 << 30)) >> 20,
                      "base_mod_2MiB": base % (1 << 21)}), flush=True)
    del be
    torch.cuda.empty_cache()


for t in range(3):
    once(f"fresh{t}")
junk = [torch.empty(40 << 30, dtype=torch.uint8, device="cuda") for _ in range(5)]     # churn: 200 GiB in five pieces
del junk[1], junk[2]
once("after-churn-holes")
del junk
torch.cuda.empty_cache()
for t in range(2):
    once(f"after-free{t}")
pad = torch.empty((1 << 30) + (37 << 20), dtype=torch.uint8, device="cuda")           # shift the next allocation
once("shifted")
del pad
