import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cytvdn_amd import _lib
_lib.ctx(0)
for gib in (4, 30, 60, 124):
    t0 = time.perf_counter()
    b = _lib.DeviceBlock(gib << 30, 0)
    t1 = time.perf_counter()
    b.free()
    t2 = time.perf_counter()
    print(f"block of {gib} GiB: alloc {t1 - t0:.3f} s, free {t2 - t1:.3f} s", flush=True)
