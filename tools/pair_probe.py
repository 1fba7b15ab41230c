#!/usr/bin/env python3
"""Is the placement lottery a matter of PAIRS of arrays?  For several allocations of the config-2
state held in turn: the sweep's time, then a pure 1-read / 1-write stream (tvdn_stream_mix) between every ordered pair
i -> j of the 15 arrays, then the 5-read / 4-write mix over random 9-subsets.  If a slow allocation shows a few slow pairs
(or subsets that are slow exactly when they hold one array), a state could be assembled from compatible arrays instead
of drawn whole.

    python tools/pair_probe.py [--allocs 4] > profiles/r03_pair_probe.jsonl
"""
import argparse, ctypes as C, json, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cytvdn_amd import _lib
from cytvdn_amd.engine import HipBackend, SlabLayout

ap = argparse.ArgumentParser()
ap.add_argument("--allocs", type=int, default=4)
ap.add_argument("--hold", type=int, default=2, help="allocations held at once (placements differ more between held states)")
a = ap.parse_args()
shape, dt = (256, 256, 128, 128), np.dtype(np.float32)
L = _lib.lib()
lay = SlabLayout(shape, 0, 1, 2)
n_bytes = int(np.prod(shape)) * 4
stride = -(-n_bytes // 256) * 256 + 4096
n_arr = 15
stream = _lib.current_stream(0)


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def mix(ins, outs):
    pi = (C.c_void_p * len(ins))(*ins)
    po = (C.c_void_p * len(outs))(*outs)
    _lib.check(L.tvdn_stream_mix(len(ins), pi, len(outs), po, n_bytes, stream))


done = 0
rng = random.Random(7)
while done < a.allocs:
    held = []
    for _ in range(min(a.hold, a.allocs - done)):
        held.append(HipBackend(lay, dt, True, device=0, max_iters=1))
    for be in held:
        sweep = be.probe_ms(3)
        base = be._slab.data_ptr()
        ptr = [base + i * stride for i in range(n_arr)]
        mix(ptr[:1], ptr[1:2])                                   # warm the kernel
        pair = np.zeros((n_arr, n_arr))
        for i in range(n_arr):
            for j in range(n_arr):
                if i != j:
                    pair[i, j] = timed(lambda: mix([ptr[i]], [ptr[j]]), reps=2)
        subs = []
        for _ in range(24):
            pick = rng.sample(range(n_arr), 9)
            subs.append((sorted(pick), round(timed(lambda: mix([ptr[k] for k in pick[:5]], [ptr[k] for k in pick[5:]]), reps=2), 4)))
        full = timed(lambda: mix(ptr[:10], ptr[10:]))
        off = pair[~np.eye(n_arr, dtype=bool)]
        worst = sorted(((round(float(pair[i, j]), 4), i, j) for i in range(n_arr) for j in range(n_arr) if i != j), reverse=True)[:8]
        print(json.dumps({"base": hex(base), "sweep_ms": round(sweep, 4), "mix_10r5w_ms": round(full, 4),
                          "pair_1r1w_ms": {"min": round(float(off.min()), 4), "median": round(float(np.median(off)), 4),
                                           "max": round(float(off.max()), 4), "slowest": worst},
                          "per_array_mean_as_source": [round(float(pair[i][pair[i] > 0].mean()), 4) for i in range(n_arr)],
                          "per_array_mean_as_dest": [round(float(pair[:, j][pair[:, j] > 0].mean()), 4) for j in range(n_arr)],
                          "subsets_5r4w": subs}), flush=True)
        done += 1
    del held
    torch.cuda.empty_cache()
