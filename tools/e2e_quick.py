#!/usr/bin/env python3
"""denoise4D from NumPy, config 2 by default, N calls, wall time of each (TVDN_RUN_TIMING=1 adds tvdn_run's own phases)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cytvdn_amd as tv
from cytvdn_amd import _lib, synth

shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "256x256x128x128").split("x"))
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
_lib.ctx(0)
buf = torch.empty(shape, dtype=torch.float32, device="cuda")
_lib.check(_lib.lib().tvdn_synth_fill(0, 4, _lib.shape_arr(shape), synth.SEED_4D, 0, shape[0], buf.data_ptr(), _lib.current_stream(0)))
x = buf.cpu().numpy(); del buf; torch.cuda.empty_cache()
mu = np.array([1, 1, .5, .5], np.float32)
for rep in range(reps):
    t0 = time.perf_counter()
    recon, bn, dl = tv.denoise4D(x, mu, iters, quiet=True)
    t = time.perf_counter() - t0
    print(json.dumps({"shape": shape, "iters": iters, "seconds": round(t, 3), "env": {k: v for k, v in os.environ.items() if k.startswith("TVDN_")},
                      "Gvoxel_iters_per_s_end_to_end": round(np.prod(shape) * iters / t / 1e9, 2), "b_norm_last": float(bn[-1])}), flush=True)
    del recon
