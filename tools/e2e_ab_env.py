#!/usr/bin/env python3
"""denoise4D from NumPy, ONE process and so one placement of the state (the kept block), calls alternating between two settings of
an environment variable the library reads per call:   python tools/e2e_ab_env.py VAR=VALUE ITERS PAIRS"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cytvdn_amd as tv
from cytvdn_amd import _lib, synth

var, val = sys.argv[1].split("=")
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
pairs = int(sys.argv[3]) if len(sys.argv) > 3 else 4
shape = (256, 256, 128, 128)
_lib.ctx(0)
buf = torch.empty(shape, dtype=torch.float32, device="cuda")
_lib.check(_lib.lib().tvdn_synth_fill(0, 4, _lib.shape_arr(shape), synth.SEED_4D, 0, shape[0], buf.data_ptr(), _lib.current_stream(0)))
x = buf.cpu().numpy(); del buf; torch.cuda.empty_cache()
mu = np.array([1, 1, .5, .5], np.float32)
tv.denoise4D(x, mu, 4, quiet=True)
for rep in range(pairs):
    for setting in (None, val):
        if setting is None:
            os.environ.pop(var, None)
        else:
            os.environ[var] = setting
        t0 = time.perf_counter()
        recon, bn, dl = tv.denoise4D(x, mu, iters, quiet=True)
        t = time.perf_counter() - t0
        print(json.dumps({"iters": iters, var: setting, "seconds": round(t, 4), "Gvoxel_iters_per_s_end_to_end": round(np.prod(shape) * iters / t / 1e9, 2),
                          "b_norm_last": float(bn[-1])}), flush=True)
        del recon
os.environ.pop(var, None)
