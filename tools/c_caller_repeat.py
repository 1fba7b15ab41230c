#!/usr/bin/env python3
"""tvdn_run as a C program would call it -- no workspace -- N times on the config-2 cube: with the kept state block (default)
and with TVDN_KEEP_STATE=0 (a hipMalloc and a hipFree of 60 GiB per call)."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cytvdn_amd import _lib, synth

shape = (256, 256, 128, 128)
n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 50
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
_lib.ctx(0)
buf = torch.empty(shape, dtype=torch.float32, device="cuda")
_lib.check(_lib.lib().tvdn_synth_fill(0, 4, _lib.shape_arr(shape), synth.SEED_4D, 0, shape[0], buf.data_ptr(), _lib.current_stream(0)))
x = buf.cpu().numpy(); del buf; torch.cuda.empty_cache()
mu = np.array([1, 1, .5, .5], np.float32); lam = mu / np.float32(32)
recon, sums = np.empty_like(x), np.zeros((n_it, 3))
a = _lib.RunArgs(dtype=0, ndim=4, bc_mode=2, device=0, n_fista=n_it, n_plain=0)
for i, s in enumerate(shape):
    a.shape[i] = s
for q in range(4):
    a.clip[q] = float((1 / lam)[q]); a.lambda_mu[q] = float((lam / mu)[q])
a.data, a.recon_out, a.sums_out = x.ctypes.data, recon.ctypes.data, sums.ctypes.data
for keep in ("1", "0"):
    os.environ["TVDN_KEEP_STATE"] = keep
    _lib.lib().tvdn_release_cache()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); _lib.check(_lib.lib().tvdn_run(C.byref(a))); t.append(round(time.perf_counter() - t0, 3))
    print(json.dumps({"TVDN_KEEP_STATE": keep, "iterations": n_it, "seconds_per_call": t, "b_norm_last": float(sums[-1, 0])}), flush=True)
_lib.lib().tvdn_release_cache()
