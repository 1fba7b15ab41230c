#!/usr/bin/env python3
"""Does the LAYOUT of the state inside its allocation (offset of the first array, stagger between arrays) move the
sweep's time the way a fresh allocation does (11.1-12.6 ms for config 2 by placement)?  One
allocation at a time, every layout carved from the very same pages and timed in alternating rounds; then the next
allocation.  If layouts rank the same on every allocation, placement can be had by construction instead of by audition.

    python tools/layout_probe.py [--shape 256x256x128x128] [--allocs 3] [--rounds 3] > profiles/r03_layout_probe.jsonl
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cytvdn_amd.engine import HipBackend, SlabLayout

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="256x256x128x128")
ap.add_argument("--dtype", default="float32")
ap.add_argument("--plain", action="store_true")
ap.add_argument("--allocs", type=int, default=3)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--layouts", default="0:4096,0:0,0:256,0:1024,0:2048,0:8192,0:16384,0:65536,0:1052672,0:2101248,"
                                     "4096:4096,65536:4096,1048576:4096,2097152:4096,16777216:4096,1073741824:4096",
                help="comma-separated offset:skew pairs in bytes")
a = ap.parse_args()
shape = tuple(int(v) for v in a.shape.split("x"))
dt = np.dtype(a.dtype)
tdt = torch.float32 if dt == np.float32 else torch.float64
fista = not a.plain
nd = len(shape)
layouts = [tuple(int(v) for v in p.split(":")) for p in a.layouts.split(",")]
n_arr = 3 + nd * (3 if fista else 2)
n_el = int(np.prod(shape))
need = max(off + n_arr * (-(-(n_el * dt.itemsize) // 256) * 256 + skew) for off, skew in layouts)
lay = SlabLayout(shape, 0, 1, 2)
for alloc in range(a.allocs):
    big = torch.empty(need // dt.itemsize + 64, dtype=tdt, device="cuda")
    res = {l: [] for l in layouts}
    for rnd in range(a.rounds):
        for off, skew in layouts:
            os.environ["TVDN_ARRAY_SKEW"] = str(skew)
            be = HipBackend(lay, dt, fista, device=0, max_iters=1, slab=big[off // dt.itemsize:])
            res[(off, skew)].append(round(be.probe_ms(3), 4))
            del be
    for (off, skew), ms in res.items():
        print(json.dumps({"alloc": alloc, "base": hex(big.data_ptr()), "offset": off, "skew": skew, "ms": ms, "best": min(ms)}), flush=True)
    del big
    torch.cuda.empty_cache()
