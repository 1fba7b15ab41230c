#!/usr/bin/env python3
"""Repeated calls must not leak: N calls of each kind of denoise4D call on a 512 MiB cube -- resident (pipelined transfers on),
resident plain order, device list on one GPU, streamed -- with the free HBM (driver's view, torch's cache emptied) and the
process's resident set before / after each series."""
import gc, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cytvdn_amd as tv
from cytvdn_amd import synth

def rss_mib():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 2 ** 20

def free_mib():
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    return torch.cuda.mem_get_info(0)[0] / 2 ** 20

shape = (64, 32, 256, 256)          # 512 MiB
x = synth.cube(shape, seed=1, dtype=np.float32) + np.float32(0.25)
mu = np.array([1, 1, .5, .5], np.float32)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
kinds = {"resident, pipelined": ({}, {}), "resident, plain order": ({"TVDN_PIPELINE": "0"}, {}),
         "device list [0, 0]": ({}, {"device": [0, 0]}), "streamed 8 rows x 3": ({"TVDN_WAVEFRONT": "8,3"}, {}),
         "loop in Python (TVDN_LOOP=native)": ({"TVDN_LOOP": "native"}, {}),
         # round 4: rows resident in HBM beside the streamed ones; a device list whose slabs are streamed; a stopping rule
         "streamed 4 rows x 3, rows resident": ({"TVDN_WAVEFRONT": "4,3"}, {}),
         "streamed device list [0, 0, 0]": ({"TVDN_WAVEFRONT": "4,3"}, {"device": [0, 0, 0]}),
         "streamed with a stopping rule": ({"TVDN_STAGED": "8,1"}, {"stopping_relative_change": 1e-9}),
         # round 6: the resident run with a rule (host-visible sums mirror and two events per run), also over slabs
         "resident with a stopping rule": ({}, {"stopping_relative_change": 1e-9}),
         "device list [0, 0] with a stopping rule": ({}, {"device": [0, 0], "stopping_relative_change": 1e-9})}
for name, (env, kw) in kinds.items():
    os.environ.update(env)
    for _ in range(3):
        tv.denoise4D(x, mu, 6, quiet=True, **kw)
    gc.collect()
    f0, r0 = free_mib(), rss_mib()
    for _ in range(n):
        tv.denoise4D(x, mu, 6, quiet=True, **kw)
    gc.collect()
    f1, r1 = free_mib(), rss_mib()
    for k in env:
        del os.environ[k]
    print(json.dumps({"kind": name, "calls": n, "hbm_leak_MiB": round(f0 - f1, 1), "rss_growth_MiB": round(r1 - r0, 1),
                      "rss_growth_per_call_MiB": round((r1 - r0) / n, 2)}), flush=True)
