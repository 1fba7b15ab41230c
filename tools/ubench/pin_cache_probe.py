#!/usr/bin/env python3
"""GPU box, harmless: how does the runtime move a caller's PAGEABLE array?  Run under AMD_LOG_LEVEL=4 and filter the runtime's own
log for its two paths -- "HSA Async Copy staged H2D / D2H" (through its staging buffer) and "HSA Copy Using Pinned resource size N"
(the caller's pages pinned IN PLACE, the pinned object kept in a cache by address and size) -- per copy size, for torch's copies
(hipMemcpy from / to pageable memory) and for the library's (tvdn_copy_to_device / _to_host: its own pinned lanes).  The mechanism
behind the GPU memory-access fault of profiles/r06_abort_found.txt; nothing is freed under the runtime's feet: no fault is provoked.

    AMD_LOG_LEVEL=4 python3 tools/ubench/pin_cache_probe.py 2> log; python3 tools/ubench/pin_cache_probe.py --digest log"""
import json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

if len(sys.argv) > 2 and sys.argv[1] == "--digest":
    cur, out = None, {}
    for line in open(sys.argv[2], errors="replace"):
        m = re.search(r"PROBE (\S+) (\d+)", line)
        if m:
            cur = f"{m.group(1)} {m.group(2)} bytes"
            out[cur] = {"staged": 0, "pinned_in_place": 0, "pinned_sizes": []}
            continue
        if cur is None:
            continue
        if "Copy staged" in line:
            out[cur]["staged"] += 1
        m = re.search(r"Copy Using Pinned resource size (\d+)", line)
        if m:
            out[cur]["pinned_in_place"] += 1
            out[cur]["pinned_sizes"].append(int(m.group(1)))
    for k, v in out.items():
        v["pinned_sizes"] = sorted(set(v["pinned_sizes"]))[:6]
        print(json.dumps({"copy": k, **v}))
    sys.exit(0)

import numpy as np
import torch
from cytvdn_amd import _lib

_lib.ctx(0)
for nbytes in (2048, 64 << 10, 2600 << 10, 40 << 20):
    dev = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    x = np.full(nbytes, 7, np.uint8)
    torch.cuda.synchronize()
    print(f"PROBE runtime_h2d {nbytes}", file=sys.stderr, flush=True)
    dev.copy_(torch.from_numpy(x))
    torch.cuda.synchronize()
    print(f"PROBE runtime_d2h {nbytes}", file=sys.stderr, flush=True)
    back = dev.cpu()
    print(f"PROBE library_h2d {nbytes}", file=sys.stderr, flush=True)
    _lib.copy_to_device(x, dev)
    print(f"PROBE library_d2h {nbytes}", file=sys.stderr, flush=True)
    y = _lib.copy_to_host(dev, np.uint8)
    print(f"PROBE end {nbytes}", file=sys.stderr, flush=True)
    assert int(y[0]) == 7 and int(back[0]) == 7
