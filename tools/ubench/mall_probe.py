#!/usr/bin/env python3
"""Does data that was just touched come back faster than HBM (the 256 MB Infinity Cache)?  Copy and read-only rates over
working sets from 16 MiB to 4 GiB, repeated in place.  If small sets run well above the 4 GiB rate, a schedule that keeps
one level's freshly written rows hot for the next level (two iterations per trip through HBM) has something to gain."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cytvdn_amd import _lib

dev = torch.device("cuda", 0)
L = _lib.lib()
_lib.ctx(0)
for mib in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 4096):
    n = mib * (1 << 20) // 4
    a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    reps = max(4, min(400, (64 << 30) // (mib << 20) // 8))
    out = {"working_set_MiB_per_array": mib, "reps": reps}
    # copy: 1 read + 1 write per byte; working set 2 arrays
    for _ in range(3):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    out["copy_GBps_moved"] = round(2 * n * 4 / ms / 1e6, 1)
    # read-only: sum
    for _ in range(3):
        a.sum()
    e0.record()
    for _ in range(reps):
        a.sum()
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    out["read_GBps"] = round(n * 4 / ms / 1e6, 1)
    print(json.dumps(out), flush=True)
    del a, b
