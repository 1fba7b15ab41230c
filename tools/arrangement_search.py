#!/usr/bin/env python3
"""GPU box: is a search over ARRANGEMENTS of a state's granules worth its time?  

The config-2 state (60 GiB, 15 arrays) on one block of granules; the block is re-dealt K times (tvdn_mem_resize at its own size:
same granules, new random order, new address -- 2 ms of remapping + a TLB flush) and every arrangement is timed for `--sweeps`
fused sweeps after one untimed one.  Then the same with a FRESH random subset of a pool per deal (free + allocate).  One JSON
line: every time, first / mean / best-of-K, and what the best would have gained over taking the first."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cytvdn_amd import _lib
from cytvdn_amd.engine import HipBackend, SlabLayout, ARRAY_SKEW

ap = argparse.ArgumentParser()
ap.add_argument("--deals", type=int, default=8)
ap.add_argument("--sweeps", type=int, default=4)
ap.add_argument("--shape", default="256x256x128x128")
ap.add_argument("--fresh", action="store_true", help="a new random subset of a new pool per deal instead of a re-deal of the same granules")
a = ap.parse_args()
shape = tuple(int(v) for v in a.shape.split("x"))
lay = SlabLayout(shape, 0, 1, 2)
n_el = int(np.prod(shape))
stride_el = (-(-(n_el * 4) // 256) * 256 + ARRAY_SKEW) // 4
n_arr = 3 + len(shape) * 3
nbytes = n_arr * stride_el * 4
blk = _lib.DeviceBlock(nbytes, 0)
times, remap_ms = [], []
for k in range(a.deals):
    be = HipBackend(lay, np.float32, True, device=0, max_iters=1, slab=blk.tensor(torch.float32), private_ctx=True)
    times.append(round(be.probe_ms(sweeps=a.sweeps), 4))
    del be
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if a.fresh:
        blk.free()
        blk = _lib.DeviceBlock(nbytes, 0)
    else:
        blk.resize()
    remap_ms.append(round(1e3 * (time.perf_counter() - t0), 2))
st = _lib.mem_status(0)
blk.free()
print(json.dumps({"what": "fresh subsets of fresh pools" if a.fresh else "re-deals of one block's granules", "shape": list(shape), "kind": blk.kind, "deals": a.deals,
                  "sweep_ms": times, "first": times[0], "mean": round(float(np.mean(times)), 4), "best": min(times), "worst": max(times),
                  "best_over_first": round(min(times) / times[0], 4), "best_over_mean": round(min(times) / float(np.mean(times)), 4),
                  "redeal_ms": remap_ms, "pool": st["last_pool"], "granules": st["last_granules"], "faults": st["faults"]}), flush=True)
