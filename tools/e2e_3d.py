#!/usr/bin/env python3
"""denoise3D from NumPy on a 1024^3 cube (4 GiB, 12 state arrays), 50 FISTA iterations: pipelined transfers against the plain
order, alternating; same bits.  (TVDN_RUN_TIMING=1 adds tvdn_run's phases.)"""
import hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cytvdn_amd as tv
from cytvdn_amd import _lib, synth

shape = (1024, 1024, 1024)
_lib.ctx(0)
buf = torch.empty(shape, dtype=torch.float32, device="cuda")
_lib.check(_lib.lib().tvdn_synth_fill(0, 3, _lib.shape_arr(shape), synth.SEED_3D, 0, shape[0], buf.data_ptr(), _lib.current_stream(0)))
x = buf.cpu().numpy(); del buf; torch.cuda.empty_cache()
mu = np.array([1, 1, .5], np.float32)
for pipe in ("1", "0", "1", "0"):
    os.environ["TVDN_PIPELINE"] = pipe
    t0 = time.perf_counter(); r = tv.denoise3D(x, mu, 50, FISTA=True, quiet=True); t = time.perf_counter() - t0
    print(json.dumps({"pipelined": pipe == "1", "seconds": round(t, 3), "Gvoxel_iters_per_s_end_to_end": round(2 ** 30 * 50 / t / 1e9, 1),
                      "recon_sha1": hashlib.sha1(r[0].tobytes()).hexdigest()[:12], "b_norm_last": float(r[1][-1])}), flush=True)
