#!/usr/bin/env python3
"""One fresh process, one JSON line: the config-2 sweep (10 timed steps) on a state that is
  first       the first large allocation of the process,
  pad N       allocated behind a pad of N GiB that stays allocated,
  arena N     carved out of one allocation of 61 + N GiB at offset N GiB.
Round 4: inside one allocation the sweep time is a smooth function of the offset (profiles/r04_arena_offset_probe.jsonl: 12.5 ms at
the start of both arenas seen, 11.0 ms 32 GiB in) -- does keeping the first pages of a process's device memory out of the state help?"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cytvdn_amd import _lib, synth
from cytvdn_amd.engine import HipBackend, SlabLayout, fista_ratios

mode = sys.argv[1] if len(sys.argv) > 1 else "first"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 0
shape = (256, 256, 128, 128)
dt = np.float32
mu = np.array([1, 1, .5, .5], dt)
lam = mu / dt(32)
r = fista_ratios(16)
keep = None
slab = None
if mode == "pad":
    keep = torch.empty(n << 30, dtype=torch.uint8, device="cuda")
elif mode == "arena":
    keep = torch.empty((61 + n) * (1 << 28), dtype=torch.float32, device="cuda")
    slab = keep[n * (1 << 28):]
be = HipBackend(SlabLayout(shape, 0, 1, 2), dt, True, device=0, max_iters=16, slab=slab)
be.set_params(1.0 / lam, (lam / mu).astype(dt))
_lib.check(_lib.lib().tvdn_synth_fill(0, 4, _lib.shape_arr(shape), synth.SEED_4D, 0, shape[0], be.orig.data_ptr(), _lib.current_stream(0)))
be.recon[be.cur].copy_(be.orig)
for i in range(3):
    be.step(float(r[i]), i)
torch.cuda.synchronize()
_lib.check(_lib.lib().tvdn_ctx_timing_enable(be.ctx, 1))
for i in range(3, 13):
    be.step(float(r[i]), i)
torch.cuda.synchronize()
ms, cnt = C.c_double(), C.c_int64()
_lib.check(_lib.lib().tvdn_ctx_timing_read(be.ctx, C.byref(ms), C.byref(cnt)))
print(json.dumps({"mode": mode, "GiB": n, "kernel_ms": round(ms.value / cnt.value, 4), "state_base": hex(be._slab.data_ptr())}), flush=True)
