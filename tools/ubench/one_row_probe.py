#!/usr/bin/env python3
"""GPU box: what a launch of the streamed engine's RING sweep costs by its height, with and without the axis-0 accumulator handed
across the cut between consecutive launches (tvdn.h TVDN_SWEEP_*).  One level of a 4-D FISTA pass on rings of
R + 2 rows of config-5 planes (1024x256x256 f32 = 256 MiB): launches of R rows follow each other over `--rows` rows, timed with
the context's events.  Reported per plane (row) so that heights compare: ms per row, plane moves per row, GB/s of what is moved."""
import argparse, ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from cytvdn_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--plane", default="1024x256x256")
ap.add_argument("--rows", type=int, default=96)
ap.add_argument("--heights", default="1,2,4,8")
ap.add_argument("--mode", type=int, default=2)
a = ap.parse_args()
plane = tuple(int(v) for v in a.plane.split("x"))
nd = 1 + len(plane)
L, ctx = _lib.lib(), _lib.new_ctx(0)
N0 = a.rows + 2
out = []
for R in [int(v) for v in a.heights.split(",")]:
    cap = R + 2
    def ring(n=cap):
        return torch.zeros((n,) + plane, dtype=torch.float32, device="cuda")
    t = {"orig": ring(R + 1), "r_in": ring(), "r_out": ring(), "s1": [ring() for _ in range(nd)], "s2": [ring() for _ in range(nd)], "o": [ring() for _ in range(nd)]}
    sums = torch.zeros(3, dtype=torch.float64, device="cuda")
    for chained in (False, True, False, True):
        it = _lib.IterArgs(dtype=0, ndim=nd, row_lo=0, row_hi=N0, lo_mode=_lib.EDGE_BC, hi_mode=_lib.EDGE_ZERO, bc_mode=2, mode=a.mode, tk=0.4, tk_prev=0.3, accumulate=1)
        it.shape[0] = N0
        for i, s in enumerate(plane):
            it.shape[1 + i] = s
        for q in range(nd):
            it.clip[q], it.lambda_mu[q] = 1.0, 1.0 / 32.0
            it.b_in[q] = it.b_out[q] = it.d_in[q] = it.d_out[q] = it.dprev_in[q] = None
            if a.mode == 2:
                it.dprev_in[q], it.d_in[q], it.d_out[q] = t["s1"][q].data_ptr(), t["s2"][q].data_ptr(), t["o"][q].data_ptr()
            else:
                it.b_in[q], it.b_out[q] = t["s1"][q].data_ptr(), t["o"][q].data_ptr()
        it.orig, it.recon_in, it.recon_out = t["orig"].data_ptr(), t["r_in"].data_ptr(), t["r_out"].data_ptr()
        it.ring_rows, it.orig_ring_rows = cap, R + 1
        launches = [(c0, min(c0 + R, 1 + a.rows)) for c0 in range(1, 1 + a.rows, R)]
        def go(timed):
            if timed:
                _lib.check(L.tvdn_ctx_timing_enable(ctx, 1))
            for i, (c0, c1) in enumerate(launches):
                it.sweep_lo, it.sweep_hi = c0, c1
                it.chain = ((_lib.SWEEP_CHAIN_LO if i > 0 else 0) | (_lib.SWEEP_STORE_AHEAD if c1 < N0 else 0)) if chained else 0
                _lib.check(L.tvdn_iterate_fused(ctx, C.byref(it), sums.data_ptr(), _lib.current_stream(0)))
            torch.cuda.synchronize()
            if timed:
                each = (C.c_double * (len(launches) + 8))()
                nl = C.c_int64()
                _lib.check(L.tvdn_ctx_timing_read_each(ctx, each, len(launches) + 8, C.byref(nl)))
                _lib.check(L.tvdn_ctx_timing_enable(ctx, 0))
                return np.array(each[:nl.value])
        go(False)
        ms = go(True)
        full = ms[1:-1] if len(ms) > 2 else ms           # (the first launch is never chained, the last may be short)
        per_row = float(np.mean(full)) / R
        pass_arr = 15 if a.mode == 2 else 11
        extra = (5.0 if chained else 6.0) if a.mode == 2 else (3.0 if chained else 4.0)   # plane moves per launch beyond R x the passes
        moves = pass_arr + extra / R
        row_bytes = float(np.prod(plane)) * 4
        out.append({"rows_per_launch": R, "handover": chained, "ms_per_launch": round(float(np.mean(full)), 4), "ms_per_row": round(per_row, 4),
                    "plane_moves_per_row": round(moves, 3), "moved_TBps": round(moves * row_bytes / (per_row * 1e-3) / 1e12, 3), "launches": len(ms)})
        print(json.dumps(out[-1]), flush=True)
    del t
    torch.cuda.empty_cache()
