#!/usr/bin/env python3
"""Small cubes are bound by the host side of an iteration, not by the sweep: per-iteration wall time of denoise3D/4D
with the loop in Python (one ctypes call + role rotation per iteration) against the native loop (tvdn_iterate_many) and the whole call behind tvdn_run (the default: hipMalloc per call, no torch cache)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cytvdn_amd as tv
from cytvdn_amd import synth

for shape, iters in (((64, 64, 256), 2000), ((128, 128, 512), 400), ((32, 32, 64, 64), 1000), ((64, 64, 64, 64), 400)):
    nd = len(shape)
    x = synth.cube(shape, dtype=np.float32)
    mu = np.array([1, 1, .5, .5][:nd] if nd == 4 else [1, 1, .5], np.float32)
    fn = tv.denoise4D if nd == 4 else tv.denoise3D
    out = {"shape": shape, "iters": iters}
    for mode in ("python", "native", "run"):
        os.environ["TVDN_LOOP"] = mode
        fn(x, mu, 10, FISTA=True, quiet=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(x, mu, iters, FISTA=True, quiet=True)
        dt = time.perf_counter() - t0
        out[f"{mode}_us_per_iter"] = round(dt / iters * 1e6, 1)
        out[f"{mode}_Gvoxel_iters_per_s"] = round(np.prod(shape) * iters / dt / 1e9, 2)
        out[f"{mode}_check"] = float(r[1][-1])
    print(json.dumps(out), flush=True)
