#!/usr/bin/env python3
"""Where inside ONE big allocation does the config-2 state run fast?  An arena of ARENA GiB is allocated once; the 60 GiB state is
bound at offsets 0, STEP, 2 STEP, ... of it (HipBackend(slab=...)), 10 sweeps timed at each (kernel time from the library's
events), the whole series twice.  If the sweep time follows the offset, placement can be had by construction (a kept arena and
a chosen offset) instead of by audition; if it does not, the lottery is in the pages behind the arena, not in the position.

    python tools/arena_offset_probe.py [ARENA_GIB=240] [STEP_GIB=20] [ROUNDS=2]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cytvdn_amd import _lib, synth
from cytvdn_amd.engine import HipBackend, SlabLayout, fista_ratios

arena_gib = int(sys.argv[1]) if len(sys.argv) > 1 else 240
step_gib = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
shape = (256, 256, 128, 128)
dt = np.float32
mu = np.array([1, 1, .5, .5], dt)
lam = mu / dt(32)
arena = torch.empty(arena_gib * (1 << 28), dtype=torch.float32, device="cuda")        # float32 elements: 2^28 per GiB
need_gib = 61
r = fista_ratios(16)


def timed(off_gib):
    sl = arena[off_gib * (1 << 28):(off_gib + need_gib) * (1 << 28)]
    be = HipBackend(SlabLayout(shape, 0, 1, 2), dt, True, device=0, max_iters=16, slab=sl)
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
    ms, n = C.c_double(), C.c_int64()
    _lib.check(_lib.lib().tvdn_ctx_timing_read(be.ctx, C.byref(ms), C.byref(n)))
    _lib.check(_lib.lib().tvdn_ctx_timing_enable(be.ctx, 0))
    del be
    return ms.value / n.value


print(json.dumps({"arena_GiB": arena_gib, "arena_base": hex(arena.data_ptr()), "state_GiB": need_gib}), flush=True)
for rd in range(rounds):
    for off in range(0, arena_gib - need_gib + 1, step_gib):
        print(json.dumps({"round": rd, "offset_GiB": off, "kernel_ms": round(timed(off), 4)}), flush=True)
# and the same state in allocations of its own, for comparison with the lottery
del arena
torch.cuda.empty_cache()
for t in range(3):
    own = torch.empty(need_gib * (1 << 28), dtype=torch.float32, device="cuda")
    arena = own
    print(json.dumps({"own_allocation": t, "base": hex(own.data_ptr()), "kernel_ms": round(timed(0), 4)}), flush=True)
    del own, arena
    torch.cuda.empty_cache()
