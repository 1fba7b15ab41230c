#!/usr/bin/env python3
"""A/B of the fused sweep's launch-time knobs INSIDE one process on ONE allocation of the state: the knobs are environment
variables the library reads at every launch (TVDN_CHUNK, TVDN_XCD, TVDN_PATCH, TVDN_PATCH_AFAST, ...), so variants can
alternate step by step on the very same physical pages.  A process per variant (rounds 1 and 2) gives every variant
another placement of its 60 GiB -- a +-6 % lottery (profiles/r03_placement_audition_*.jsonl) that drowned every effect
smaller than that in rounds 1 and 2.

    python tools/ab_inproc.py --config 2 --rounds 3 "base:" "afast:TVDN_PATCH=8,8;TVDN_PATCH_AFAST=1" ...
"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cytvdn_amd import _lib, synth
from cytvdn_amd.engine import HipBackend, SlabLayout, fista_ratios

CONFIGS = {"2": ((256, 256, 128, 128), np.float32, True), "3": ((256, 256, 128, 128), np.float64, False),
           "plain32": ((256, 256, 128, 128), np.float32, False), "3d": ((512, 512, 512), np.float32, True),
           "3dplain": ((512, 512, 512), np.float32, False), "c1": ((128, 128, 512), np.float32, True),
           "slab": ((66, 512, 256, 256), np.float32, True), "f64fista": ((256, 256, 128, 128), np.float64, True),
           # one interior slab of BASELINE configs[3] exactly as bench.py --slab-of 8 runs it: halo edges, two 8-row edge
           # launches, halo rows refreshed on a side stream, then the 48-row interior launch
           "slab8": ((512, 512, 256, 256), np.float32, True)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="2", choices=sorted(CONFIGS))
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--audition", type=int, default=1)
    ap.add_argument("--shape", default=None, help="e.g. 1024x512x512: overrides the config's shape (dtype and FISTA stay)")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    shape, dt, fista = CONFIGS[a.config]
    if a.shape:
        shape = tuple(int(v) for v in a.shape.lower().split("x"))
    dt = np.dtype(dt)
    nd = len(shape)
    mu = np.array([1.0, 1.0, 0.5, 0.5][:nd] if nd == 4 else [1.0, 1.0, 0.5], dt)
    lam = mu / dt.type(32.0 if nd == 4 else 16.0)
    R = fista_ratios(4096)
    L = _lib.lib()
    lay = SlabLayout(shape, 4, 8, 2) if a.config == "slab8" else SlabLayout(shape, 0, 1, 2)
    be = HipBackend.best_of(a.audition, lay, dt, fista, device=0, max_iters=a.steps + 4)
    be.set_params(1.0 / lam, (lam / mu).astype(dt))
    _lib.check(L.tvdn_synth_fill(be.code, nd, _lib.shape_arr(shape), synth.SEED_4D if nd == 4 else synth.SEED_3D,
                                 lay.g0 - lay.halo_lo, lay.local_shape[0], be.orig.data_ptr(), _lib.current_stream(0)))
    be.recon[be.cur].copy_(be.orig)
    variants = []
    for v in a.variants:
        label, _, envs = v.partition(":")
        variants.append((label, dict(e.split("=", 1) for e in envs.split(";") if e)))
    # a variant may name another build of the library (TVDN_LIB=path): it is loaded beside the default one and given a
    # context of its own; the state arrays (plain device pointers) are shared
    libs = {}
    for _, env in variants:
        path = env.pop("TVDN_LIB", None)
        env["_lib"] = path
        if path and path not in libs:
            Lx = C.CDLL(path)
            Lx.tvdn_ctx_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
            Lx.tvdn_ctx_timing_enable.argtypes = [C.c_void_p, C.c_int]
            Lx.tvdn_ctx_timing_read_each.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_int64)]
            Lx.tvdn_iterate_fused.argtypes = [C.c_void_p, C.POINTER(_lib.IterArgs), C.c_void_p, C.c_void_p]
            Lx.tvdn_last_error.restype = C.c_char_p
            h = C.c_void_p()
            assert Lx.tvdn_ctx_create(C.byref(h), 0) == 0, Lx.tvdn_last_error()
            libs[path] = (Lx, h)
    libs[None] = (L, be.ctx)

    emu = None
    if a.config == "slab8":
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        emu = bench.EmulatedNeighbours(be)

    def step(Lx, ctx, tk, slot):
        if emu is not None:                 # three launches per iteration, default library only
            assert Lx is L, "slab8 runs the default library"
            emu._step(tk, slot)
            return
        be._bind(tk)
        be._args.sweep_lo = be._args.sweep_hi = 0
        be._args.accumulate = 0
        rc = Lx.tvdn_iterate_fused(ctx, C.byref(be._args), C.c_void_p(be.sums[slot].data_ptr()), _lib.current_stream(0))
        assert rc == 0, Lx.tvdn_last_error()
        be.flip()

    keys = sorted({k for _, env in variants for k in env if k != "_lib"})
    base_env = {k: os.environ.get(k) for k in keys}
    res = {label: [] for label, _ in variants}
    it = 0
    for rnd in range(a.rounds):
        for label, env in variants:
            for k in keys:
                if k in env:
                    os.environ[k] = env[k]
                elif base_env[k] is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = base_env[k]
            Lx, ctx = libs[env["_lib"]]
            for i in range(2):
                step(Lx, ctx, float(R[min(it, 4095)]) if fista else None, i)
                it += 1
            torch.cuda.synchronize()
            assert Lx.tvdn_ctx_timing_enable(ctx, 1) == 0
            for i in range(a.steps):
                step(Lx, ctx, float(R[min(it, 4095)]) if fista else None, 2 + i)
                it += 1
            torch.cuda.synchronize()
            each = (C.c_double * (3 * a.steps + 4))()
            nl = C.c_int64()
            assert Lx.tvdn_ctx_timing_read_each(ctx, each, 3 * a.steps + 4, C.byref(nl)) == 0
            assert Lx.tvdn_ctx_timing_enable(ctx, 0) == 0
            res[label].append(float(np.sum(each[:nl.value])) / a.steps)     # per iteration (slab8: three launches each)
    base = np.mean(res[variants[0][0]])
    for label, env in variants:
        v = res[label]
        print(json.dumps({"config": a.config, "shape": list(shape), "variant": label, "env": {k: v for k, v in env.items() if v is not None}, "mean_ms": round(float(np.mean(v)), 4),
                          "min_ms": round(float(np.min(v)), 4), "vs_first": round(float(np.mean(v)) / base, 4),
                          "rounds": [round(x, 4) for x in v], "audition": getattr(be, "audition", [])}), flush=True)


if __name__ == "__main__":
    main()
