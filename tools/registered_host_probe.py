#!/usr/bin/env python3
"""Does host memory registered with the GPU slow the resident sweep?  Config-2 state, the sweep timed in blocks of 20 with
nothing registered, with N GiB of anonymous memory page-locked (hipHostRegister; huge pages asked for or not), and after it
has been released again.  One JSON line per phase.  (Round 4: a result array page-locked during a resident run cost the
middle iterations 9 %: profiles/r04_e2e_result_pinned_in_place.txt.)"""
import ctypes as C
import json
import mmap
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    from cytvdn_amd import _lib
    from cytvdn_amd.engine import HipBackend, SlabLayout, fista_ratios
    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    shape = (256, 256, 128, 128)
    dt = np.dtype(np.float32)
    be = HipBackend(SlabLayout(shape, 0, 1, 2), dt, True, device=0, max_iters=64)
    mu = np.array([1.0, 1.0, 0.5, 0.5], dt)
    lam = mu / dt.type(32.0)
    be.set_params(1.0 / lam, (lam / mu).astype(dt))
    be.orig.normal_()
    be.recon[0].copy_(be.orig)
    ratios = fista_ratios(64)
    hip = C.CDLL("libamdhip64.so")
    hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
    hip.hipHostUnregister.argtypes = [C.c_void_p]

    def block(label, n=20):
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        ev[0].record()
        for i in range(n):
            be.step(float(ratios[(i + 3) % 64]), i % 64)
            be.flip()
            ev[i + 1].record()
        torch.cuda.synchronize()
        ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
        print(json.dumps({"phase": label, "sweep_ms_mean": round(sum(ms) / n, 4), "min": round(min(ms), 4), "max": round(max(ms), 4)}), flush=True)

    block("warm-up")
    block("nothing registered")
    block("nothing registered (again)")
    n = int(gib * 2 ** 30)
    for huge in (True, False):
        m = mmap.mmap(-1, n, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
        addr = C.addressof(C.c_char.from_buffer(m))
        if huge:
            m.madvise(mmap.MADV_HUGEPAGE)
        np.frombuffer(m, np.uint8)[::4096] = 1
        t0 = time.perf_counter()
        rc = hip.hipHostRegister(addr, n, 0)
        t_reg = time.perf_counter() - t0
        block(f"{gib:g} GiB registered (rc {rc}, {t_reg:.3f} s, huge pages {'asked for' if huge else 'not asked for'})")
        block("... still registered")
        hip.hipHostUnregister(addr)
        block("unregistered")
        del addr
        try:
            m.close()
        except BufferError:
            pass
    t = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    block(f"{gib:g} GiB of torch pinned memory (hipHostMalloc)")
    del t
    block("released")


if __name__ == "__main__":
    main()
