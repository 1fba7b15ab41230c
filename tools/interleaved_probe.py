#!/usr/bin/env python3
"""A 10-read / 5-write stream over 15 separate 4 GiB arrays against the same bytes read and written TILE-INTERLEAVED (the
15 tiles of a workgroup back to back: one sequential stream), alternating on one allocation, several allocations:
would a tile-interleaved state take the placement lottery out of the sweep and what would it stream at?"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cytvdn_amd import _lib

L = _lib.lib()
_lib.ctx(0)
n_bytes = 4 << 30
stream = _lib.current_stream(0)
def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return round(best, 4)
held = []
for alloc in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    buf = torch.zeros(15 * (n_bytes + 4096) // 4, dtype=torch.float32, device="cuda")
    base = buf.data_ptr()
    ptr = [base + i * (n_bytes + 4096) for i in range(15)]
    pi, po = (C.c_void_p * 10)(*ptr[:10]), (C.c_void_p * 5)(*ptr[10:])
    def flat():
        os.environ.pop("TVDN_MIX_INTERLEAVED", None)
        _lib.check(L.tvdn_stream_mix(10, pi, 5, po, n_bytes, stream))
    def inter():
        os.environ["TVDN_MIX_INTERLEAVED"] = "1"
        _lib.check(L.tvdn_stream_mix(10, pi, 5, po, n_bytes, stream))
    flat(); inter()
    out = {"alloc": alloc, "base": hex(base), "separate_arrays_ms": [], "tile_interleaved_ms": []}
    for _ in range(3):
        out["separate_arrays_ms"].append(timed(flat))
        out["tile_interleaved_ms"].append(timed(inter))
    os.environ.pop("TVDN_MIX_INTERLEAVED", None)
    print(json.dumps(out), flush=True)
    held.append(buf)
    if len(held) == 3:
        held.clear(); torch.cuda.empty_cache()
