#!/usr/bin/env python3
"""What a stopping rule costs per iteration: denoise3D/4D with `stopping_relative_change` set so low that it never fires
against the same call without one (cyTVDN.py:189-195 reads delta_recon[i] after every iteration; here that is a read of
three f64 sums from the device per iteration).  One JSON line per shape; TVDN_STOP_LAG=0 is the blocking form."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cytvdn_amd as tv
from cytvdn_amd import synth

shapes = (((64, 64, 256), 2000), ((128, 128, 512), 1000), ((256, 256, 256), 400), ((512, 512, 512), 100),
          ((32, 32, 64, 64), 1000), ((64, 64, 64, 64), 400), ((64, 64, 128, 128), 100))
for shape, iters in shapes:
    nd = len(shape)
    x = synth.cube(shape, dtype=np.float32)
    mu = np.array([1, 1, .5, .5][:nd] if nd == 4 else [1, 1, .5], np.float32)
    fn = tv.denoise4D if nd == 4 else tv.denoise3D
    out = {"shape": shape, "iters": iters}
    for name, kw in (("no_rule", {}), ("rule", {"stopping_relative_change": 1e-30})):
        for fista in (True, False):
            fn(x, mu, 10, FISTA=fista, quiet=True, **kw)
            torch.cuda.synchronize()
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                r = fn(x, mu, iters, FISTA=fista, quiet=True, **kw)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            tag = f"{name}_{'fista' if fista else 'plain'}"
            out[f"{tag}_us_per_iter"] = round(best / iters * 1e6, 1)
            out[f"{tag}_check"] = float(r[2][-1])
    print(json.dumps(out), flush=True)
