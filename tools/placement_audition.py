#!/usr/bin/env python3
"""Is the "box state" of the config-2 sweep (11.2 vs 12.6 ms with identical clocks, profiles/r03_clocks_*.txt) a property
of WHERE in HBM the 60 GiB of state landed?  Holds several states at once in one process (so they are certainly in
different physical places), times each in turn, twice round-robin; then frees all and allocates again.  One JSON line per
measurement.  Also reads the clocks around it (sysfs of the busy card) so the comparison is at equal clocks."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cytvdn_amd import _lib, synth
from cytvdn_amd.engine import HipBackend, SlabLayout, fista_ratios

shape = (256, 256, 128, 128)
dt = np.float32
mu = np.array([1, 1, .5, .5], dt)
lam = mu / dt(32)
R = fista_ratios(64)


def make():
    be = HipBackend(SlabLayout(shape, 0, 1, 2), dt, True, device=0, max_iters=16)
    be.set_params(1.0 / lam, (lam / mu).astype(dt))
    _lib.check(_lib.lib().tvdn_synth_fill(0, 4, _lib.shape_arr(shape), synth.SEED_4D, 0, shape[0], be.orig.data_ptr(), _lib.current_stream(0)))
    be.recon[be.cur].copy_(be.orig)
    be.n_done = 0
    return be


def timed(be, tag, steps=12):
    for i in range(2):
        be.step(float(R[min(be.n_done, 63)]), i)
        be.n_done += 1
    torch.cuda.synchronize()
    _lib.check(_lib.lib().tvdn_ctx_timing_enable(be.ctx, 1))
    for i in range(steps):
        be.step(float(R[min(be.n_done, 63)]), 2 + i)
        be.n_done += 1
    torch.cuda.synchronize()
    ms, n = C.c_double(), C.c_int64()
    _lib.check(_lib.lib().tvdn_ctx_timing_read(be.ctx, C.byref(ms), C.byref(n)))
    _lib.check(_lib.lib().tvdn_ctx_timing_enable(be.ctx, 0))
    base = be._slab.data_ptr() if be._slab is not None else be.orig.data_ptr()
    print(json.dumps({"tag": tag, "kernel_ms": round(ms.value / n.value, 4), "base": hex(base), "t": round(time.time() - T0, 1)}), flush=True)


T0 = time.time()
n_hold = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for rnd in range(2):
    held = [make() for _ in range(n_hold)]
    for rep in range(2):
        for j, be in enumerate(held):
            timed(be, f"round{rnd}.state{j}.rep{rep}")
    # free in reverse order except the first, re-time the first alone
    while len(held) > 1:
        b = held.pop()
        del b
    torch.cuda.empty_cache()
    timed(held[0], f"round{rnd}.state0.alone")
    del held
    torch.cuda.empty_cache()


def per_array(be, tag, scratch):
    """Plain streaming over each array of the state, one at a time: copy into a fixed scratch buffer (GB/s of the pair)."""
    n_el = int(np.prod(shape))
    item = 4
    arrs = [t for S in be.S for t in S] + [be.recon[0], be.recon[1], be.orig]
    out = []
    for t in arrs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        scratch.copy_(t.view(-1))
        e0.record()
        for _ in range(3):
            scratch.copy_(t.view(-1))
        e1.record()
        torch.cuda.synchronize()
        out.append(round(3 * 2 * n_el * item / (e0.elapsed_time(e1) * 1e-3) / 1e9))
    print(json.dumps({"tag": tag, "copy_GBps_per_array": out}), flush=True)


if len(sys.argv) > 2 and sys.argv[2] == "arrays":
    scratch = torch.empty(int(np.prod(shape)), dtype=torch.float32, device="cuda")
    held = [make() for _ in range(n_hold)]
    for j, be in enumerate(held):
        timed(be, f"diag.state{j}")
        per_array(be, f"diag.state{j}", scratch)
    del held
    torch.cuda.empty_cache()
    held = [make() for _ in range(n_hold)]
    for j, be in enumerate(held):
        timed(be, f"diag2.state{j}")
        per_array(be, f"diag2.state{j}", scratch)


if len(sys.argv) > 2 and sys.argv[2] == "separate":
    # one hipMalloc per array instead of one for the whole state: is the sweep's time then the same for every state?
    for mode in ("separate", "one", "separate"):
        os.environ["TVDN_ALLOC"] = mode
        held = [make() for _ in range(n_hold)]
        for j, be in enumerate(held):
            timed(be, f"alloc-{mode}.state{j}")
        del held, be
        torch.cuda.empty_cache()
