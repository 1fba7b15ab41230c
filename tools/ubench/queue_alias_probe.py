#!/usr/bin/env python3
"""Do HIP streams share hardware queues?  One stream gets ~90 ms of kernels queued; then a tiny device-to-host copy goes
onto each of 12 other streams and the host notes when each copy's event completes.  A stream that sits on the same hardware
queue as the busy one sees its copy complete only when the queued kernels have drained.  Twice: busy stream of normal
priority, busy stream of high priority (the fix in csrc/tvdn_common.hpp make_stream)."""
import json, time
import torch

dev = torch.device("cuda", 0)
big = torch.ones(1 << 30, dtype=torch.float32, device=dev)          # 4 GiB: one in-place multiply ~ 1.5 ms
src = torch.ones(1024, dtype=torch.float32, device=dev)
dst = [torch.empty(1024, dtype=torch.float32).pin_memory() for _ in range(12)]
for prio in (0, -1):
    busy = torch.cuda.Stream(device=dev, priority=prio)
    others = [torch.cuda.Stream(device=dev) for _ in range(12)]
    torch.cuda.synchronize()
    with torch.cuda.stream(busy):
        for _ in range(4):
            big.mul_(1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done_busy = torch.cuda.Event()
    with torch.cuda.stream(busy):
        for _ in range(60):
            big.mul_(1.0)
        done_busy.record(busy)
    evs = []
    for s, d in zip(others, dst):
        with torch.cuda.stream(s):
            d.copy_(src, non_blocking=True)
            e = torch.cuda.Event()
            e.record(s)
            evs.append(e)
    t_sub = time.perf_counter() - t0
    lat = [None] * len(evs)
    busy_ms = None
    while any(v is None for v in lat) or busy_ms is None:
        now = (time.perf_counter() - t0) * 1e3
        for i, e in enumerate(evs):
            if lat[i] is None and e.query():
                lat[i] = round(now, 2)
        if busy_ms is None and done_busy.query():
            busy_ms = round(now, 2)
    print(json.dumps({"busy_stream_priority": prio, "submit_ms": round(t_sub * 1e3, 2), "busy_stream_done_ms": busy_ms,
                      "tiny_copy_done_ms_per_stream": lat,
                      "streams_that_waited_for_the_busy_one": sum(1 for v in lat if v > 0.5 * busy_ms)}), flush=True)
    del busy, others
