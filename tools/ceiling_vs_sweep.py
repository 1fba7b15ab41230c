#!/usr/bin/env python3
"""The fused sweep next to a PURE stream of the same read/write mix over the SAME arrays (the same physical pages), for
several placements of the state held at once in one process: how much of the sweep's time is the chip's own ceiling for
10 reads + 5 writes of this size in this placement, and how much is the sweep's (neighbour re-reads, arithmetic, the
reductions).  One JSON line per placement; VERDICT r2 item 6 asked for exactly this comparison "on the same box in the
same interleaved run".

    python tools/ceiling_vs_sweep.py [--config 2|3|plain32|3d] [--hold 3]
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
           "3dplain": ((512, 512, 512), np.float32, False)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="2", choices=sorted(CONFIGS))
    ap.add_argument("--hold", type=int, default=3)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--drop-recon", action="store_true",
                    help="config 2: also time the 9-read / 4-write mix of a sweep that would NOT store recon (orig + the d pairs in, "
                         "d_k+1 out): what 'fewer bytes' could reach before its wider stencil is paid for")
    a = ap.parse_args()
    shape, dt, fista = CONFIGS[a.config]
    dt = np.dtype(dt)
    nd = len(shape)
    mu = np.array([1.0, 1.0, 0.5, 0.5][:nd] if nd == 4 else [1.0, 1.0, 0.5], dt)
    lam = mu / dt.type(32.0 if nd == 4 else 16.0)
    R = fista_ratios(64)
    L = _lib.lib()
    n_bytes = int(np.prod(shape)) * dt.itemsize
    held = []
    for j in range(a.hold):
        free, _ = torch.cuda.mem_get_info(0)
        need = (3 + nd * (3 if fista else 2)) * n_bytes
        if held and need > 0.85 * free:
            break
        be = HipBackend(SlabLayout(shape, 0, 1, 2), dt, fista, device=0, max_iters=32)
        be.set_params(1.0 / lam, (lam / mu).astype(dt))
        _lib.check(L.tvdn_synth_fill(be.code, nd, _lib.shape_arr(shape), synth.SEED_4D if nd == 4 else synth.SEED_3D, 0, shape[0],
                                     be.orig.data_ptr(), _lib.current_stream(0)))
        be.recon[be.cur].copy_(be.orig)
        held.append(be)

    def sweep_ms(be, it0):
        for i in range(2):
            be.step(float(R[it0 + i]) if fista else None, i)
        torch.cuda.synchronize()
        _lib.check(L.tvdn_ctx_timing_enable(be.ctx, 1))
        for i in range(a.steps):
            be.step(float(R[it0 + 2 + i]) if fista else None, 2 + i)
        torch.cuda.synchronize()
        each = (C.c_double * (a.steps + 4))()
        nl = C.c_int64()
        _lib.check(L.tvdn_ctx_timing_read_each(be.ctx, each, a.steps + 4, C.byref(nl)))
        _lib.check(L.tvdn_ctx_timing_enable(be.ctx, 0))
        v = np.array(each[:nl.value])
        return float(v.mean()), float(v.min())

    self_march = {}

    def mix_ms(be):
        # the arrays the NEXT sweep would read and write, in its roles
        if fista:
            ins = [be.orig, be.recon[be.cur]] + [t for S in be.S for t in (S[be.i_d], S[be.i_prev])]
            outs = [be.recon[be.cur ^ 1]] + [S[be.i_out] for S in be.S]
        else:
            ins = [be.orig, be.recon[be.cur]] + [S[be.i_b] for S in be.S]
            outs = [be.recon[be.cur ^ 1]] + [S[be.i_bout] for S in be.S]
        pi = (C.c_void_p * len(ins))(*[t.data_ptr() for t in ins])
        po = (C.c_void_p * len(outs))(*[t.data_ptr() for t in outs])
        ts = []
        for i in range(a.steps + 2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(L.tvdn_stream_mix(len(ins), pi, len(outs), po, n_bytes, _lib.current_stream(0)))
            e1.record()
            torch.cuda.synchronize()
            if i >= 2:
                ts.append(e0.elapsed_time(e1))
        tm = {}
        if dt.itemsize == 4 and (len(ins), len(outs)) in ((10, 5), (6, 5)):
            for chunk in (8, 1, 32):      # the same stream walked like the sweep: 4 KiB tiles marching `chunk` rows
                tt = []
                for i in range(a.steps + 2):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    _lib.check(L.tvdn_stream_mix_march(len(ins), pi, len(outs), po, n_bytes, shape[0], chunk, _lib.current_stream(0)))
                    e1.record()
                    torch.cuda.synchronize()
                    if i >= 2:
                        tt.append(e0.elapsed_time(e1))
                tm[f"march{chunk}_ms"] = round(float(np.mean(tt)), 4)
        if a.drop_recon and fista and nd == 4 and dt.itemsize == 4:
            ins9 = [be.orig] + [t for S in be.S for t in (S[be.i_d], S[be.i_prev])]
            outs4 = [S[be.i_out] for S in be.S]
            p9 = (C.c_void_p * 9)(*[t.data_ptr() for t in ins9])
            p4 = (C.c_void_p * 4)(*[t.data_ptr() for t in outs4])
            tt = []
            for i in range(a.steps + 2):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _lib.check(L.tvdn_stream_mix(9, p9, 4, p4, n_bytes, _lib.current_stream(0)))
                e1.record()
                torch.cuda.synchronize()
                if i >= 2:
                    tt.append(e0.elapsed_time(e1))
            tm["mix_9R4W_ms"] = round(float(np.mean(tt)), 4)
            tm["mix_9R4W_GBps"] = round(13 * n_bytes / (float(np.mean(tt)) * 1e-3) / 1e9)
        self_march.update(tm)
        return float(np.mean(ts)), float(np.min(ts)), len(ins), len(outs)

    for rep in range(2):
        for j, be in enumerate(held):
            s_mean, s_min = sweep_ms(be, 12 * rep)
            m_mean, m_min, nr, nw = mix_ms(be)
            moved = (nr + nw) * n_bytes
            print(json.dumps({"config": a.config, "placement": j, "rep": rep, "sweep_ms": round(s_mean, 4), "sweep_min_ms": round(s_min, 4),
                              "stream_mix": f"{nr}R/{nw}W", "mix_ms": round(m_mean, 4), "mix_min_ms": round(m_min, 4),
                              "sweep_over_mix": round(s_mean / m_mean, 4),
                              "mix_GBps": round(moved / (m_mean * 1e-3) / 1e9), "sweep_moved_GBps": round(moved / (s_mean * 1e-3) / 1e9),
                              **self_march}),
                  flush=True)


if __name__ == "__main__":
    main()
