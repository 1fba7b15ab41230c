#!/usr/bin/env python3
"""When does a 60 GiB device allocation take seconds?  Allocate / release cycles through torch with the cache emptied after
every release (so each cycle is one hipMalloc and one hipFree), timing both; then the same with a second block held all
the while; then with a pause between the release and the next allocation."""
import json, sys, time
import torch

gib = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n = gib << 30
torch.cuda.init(); torch.empty(1, device="cuda"); torch.cuda.synchronize()

def cycle(tag, reps, pause=0.0, touch=True):
    rows = []
    for i in range(reps):
        t0 = time.perf_counter(); t = torch.empty(n, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize(); ta = time.perf_counter() - t0
        if touch:
            t.zero_(); torch.cuda.synchronize()
        t0 = time.perf_counter(); del t; torch.cuda.empty_cache(); torch.cuda.synchronize(); tf = time.perf_counter() - t0
        rows.append((round(ta * 1e3, 1), round(tf * 1e3, 1)))
        if pause:
            time.sleep(pause)
    print(json.dumps({"case": tag, "GiB": gib, "alloc_ms,free_ms": rows}), flush=True)

cycle("allocate, zero, free, again at once", 10)
cycle("allocate, free (never touched)", 6, touch=False)
cycle("... with 2 s between free and the next allocation", 6, pause=2.0)
held = torch.empty(n, dtype=torch.uint8, device="cuda"); held.zero_()
cycle("... while another block of the same size is held", 6)
