#!/usr/bin/env python3
"""What a call costs beyond its iterations: denoise3D/4D (NumPy -> NumPy) on small cubes at 0 / 1 / 10 / 100 iterations, best of 5
calls each, after a warm-up call.  One JSON line per shape; TVDN_RUN_TIMING=1 adds tvdn_run's own phase breakdown on stderr."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cytvdn_amd as tv
from cytvdn_amd import synth

for shape in ((32, 32, 128), (64, 64, 256), (128, 128, 512), (256, 256, 256), (32, 32, 64, 64), (64, 64, 64, 64)):
    nd = len(shape)
    x = synth.cube(shape, dtype=np.float32)
    mu = np.array([1, 1, .5, .5][:nd] if nd == 4 else [1, 1, .5], np.float32)
    fn = tv.denoise4D if nd == 4 else tv.denoise3D
    out = {"shape": shape, "MiB": round(x.nbytes / 2 ** 20, 1)}
    fn(x, mu, 10, FISTA=True, quiet=True)
    for iters in (0, 1, 10, 100):
        for tag, kw in (("", {}), ("_rule", {"stopping_relative_change": 1e-30})):
            best = None
            for _ in range(5):
                t0 = time.perf_counter()
                fn(x, mu, iters, FISTA=True, quiet=True, **kw)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            out[f"us_{iters}{tag}"] = round(best * 1e6, 1)
    print(json.dumps(out), flush=True)
