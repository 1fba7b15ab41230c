#!/usr/bin/env python3
"""GPU box, fresh process: what the FIRST denoise4D of a process pays over the second.  `--warm` names what is
done before the first call: nothing; `lanes` = one 512 MiB upload + download through the library's pinned staging lanes into a
scratch tensor (csrc/tvdn_hostio.hip io_init: 16 hipHostMalloc of 16 MiB, 8 streams); `clock` = 0.3 s of sweeps on a small
state (clocks up); `pool` = a 60 GiB granule block allocated and freed (memory the driver has touched); `api` = the product's
own `cytvdn_amd.warm_up()` (lanes, first priority streams, code object, canary).  One JSON line per call;
TVDN_RUN_TIMING=1 adds tvdn_run's phases on stderr."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cytvdn_amd as tv
from cytvdn_amd import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--warm", default="", help="comma list of lanes, clock, pool")
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
shape = (256, 256, 128, 128)
_lib.ctx(0)
buf = torch.empty(shape, dtype=torch.float32, device="cuda")
_lib.check(_lib.lib().tvdn_synth_fill(0, 4, _lib.shape_arr(shape), synth.SEED_4D, 0, shape[0], buf.data_ptr(), _lib.current_stream(0)))
x = buf.cpu().numpy(); del buf; torch.cuda.empty_cache()
mu = np.array([1, 1, .5, .5], np.float32)
warm = [w for w in a.warm.split(",") if w]
t_warm = {}
for w in warm:
    t0 = time.perf_counter()
    if w == "lanes":
        t = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
        _lib.copy_to_device(x.reshape(-1).view(np.uint8)[:512 << 20], t)
        _lib.copy_to_host(t, np.uint8)
        del t; torch.cuda.empty_cache()
    elif w == "clock":
        s = torch.zeros(64 << 20, dtype=torch.float32, device="cuda")
        t1 = time.perf_counter()
        while time.perf_counter() - t1 < 0.3:
            s.add_(1.0)
        torch.cuda.synchronize(); del s; torch.cuda.empty_cache()
    elif w == "api":
        tv.warm_up(0)
    elif w == "pool":
        b = _lib.DeviceBlock(61 << 30, 0); b.free()
    t_warm[w] = round(time.perf_counter() - t0, 3)
for rep in range(a.reps):
    t0 = time.perf_counter()
    recon, bn, dl = tv.denoise4D(x, mu, a.iters, quiet=True)
    t = time.perf_counter() - t0
    print(json.dumps({"call": rep, "warm": warm, "warm_s": t_warm, "iters": a.iters, "seconds": round(t, 3),
                      "Gvoxel_iters_per_s_end_to_end": round(np.prod(shape) * a.iters / t / 1e9, 2), "mem": {k: v for k, v in _lib.mem_status(0).items() if k in ("last_granules", "last_pool", "flushes", "faults")}}), flush=True)
    del recon
