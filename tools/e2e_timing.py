#!/usr/bin/env python3
"""End-to-end denoise4D timing from NumPy (PCIe-inclusive), config 2 by default, with the breakdown of what is not
sweeping: host->HBM, allocation + fill, HBM->host.  Never the bench `value`; reported in DESIGN.md."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cytvdn_amd as tv
from cytvdn_amd import _lib, synth
from cytvdn_amd.engine import HipBackend, SlabLayout

shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "256x256x128x128").split("x"))
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
_lib.ctx(0)
buf = torch.empty(shape, dtype=torch.float32, device="cuda")
_lib.check(_lib.lib().tvdn_synth_fill(0, 4, _lib.shape_arr(shape), synth.SEED_4D, 0, shape[0], buf.data_ptr(), _lib.current_stream(0)))
x = buf.cpu().numpy(); del buf; torch.cuda.empty_cache()
mu = np.array([1, 1, .5, .5], np.float32)
gb = x.nbytes / 1e9
for rep in range(3):
    # the pieces, timed separately
    torch.cuda.synchronize(); t0 = time.perf_counter()
    be = HipBackend(SlabLayout(shape, 0, 1, 2), np.float32, True, device=0, max_iters=1)
    torch.cuda.synchronize(); t_alloc = time.perf_counter() - t0
    t0 = time.perf_counter(); be.set_input(x); torch.cuda.synchronize(); t_up = time.perf_counter() - t0
    t0 = time.perf_counter(); r = be.recon_to_host(); t_down = time.perf_counter() - t0
    t0 = time.perf_counter(); r2 = be.recon_tensor().cpu().numpy(); t_down_torch = time.perf_counter() - t0
    assert r.tobytes() == x.tobytes()
    del be, r, r2; torch.cuda.empty_cache()
    # the call a user makes
    t0 = time.perf_counter()
    recon, bn, dl = tv.denoise4D(x, mu, iters, quiet=True)
    t = time.perf_counter() - t0
    print(json.dumps({"shape": shape, "iters": iters, "seconds": round(t, 3),
                      "Gvoxel_iters_per_s_end_to_end": round(np.prod(shape) * iters / t / 1e9, 2),
                      "alloc_fill_s": round(t_alloc, 3), "upload_s": round(t_up, 3), "upload_GBps": round(gb / t_up, 1),
                      "download_s": round(t_down, 3), "download_GBps": round(gb / t_down, 1),
                      "download_via_torch_cpu_s": round(t_down_torch, 3), "b_norm_last": float(bn[-1])}), flush=True)
    del recon
