import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, cytvdn_amd as tv
from cytvdn_amd import synth
shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "64x64x256").split("x"))
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
kw = {"stopping_relative_change": 1e-30} if len(sys.argv) > 3 and sys.argv[3] == "rule" else {}
nd = len(shape)
x = synth.cube(shape, dtype=np.float32); mu = np.array([1, 1, .5, .5][:nd], np.float32)
fn = tv.denoise4D if nd == 4 else tv.denoise3D
fn(x, mu, 10, FISTA=True, quiet=True, **kw)
t0 = time.perf_counter(); fn(x, mu, iters, FISTA=True, quiet=True, **kw); dt = time.perf_counter() - t0
print(shape, iters, kw, "us per iteration", round(dt / iters * 1e6, 2))
