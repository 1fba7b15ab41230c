"""D2H / H2D rates of pinned-memory copies by size and count per burst, alone and with the other direction running:
looks for the 4x-slower mode some bursts of the wavefront engine's downloads fall into (profiles/r02_wavefront_timeline.txt)."""
import sys
import time

import torch

dev = torch.device("cuda", 0)
MiB = 1 << 20
total = 576 * MiB
host_d = torch.empty(total, dtype=torch.uint8).pin_memory()
host_u = torch.empty(total + 64 * MiB, dtype=torch.uint8).pin_memory()
dbuf_d = torch.empty(total, dtype=torch.uint8, device=dev)
dbuf_u = torch.empty(total + 64 * MiB, dtype=torch.uint8, device=dev)
s_down, s_up = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
big = torch.empty(1 << 28, dtype=torch.float32, device=dev)


def burst(n_pieces, with_up, with_compute, reps=12):
    piece = total // n_pieces
    times = []
    for r in range(reps):
        torch.cuda.synchronize()
        if with_compute:
            for _ in range(6):
                big.mul_(1.0001)
        if with_up:
            with torch.cuda.stream(s_up):
                for i in range(10):
                    dbuf_u[i * 64 * MiB:(i + 1) * 64 * MiB].copy_(host_u[i * 64 * MiB:(i + 1) * 64 * MiB], non_blocking=True)
        with torch.cuda.stream(s_down):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(s_down)
            for i in range(n_pieces):
                host_d[i * piece:(i + 1) * piece].copy_(dbuf_d[i * piece:(i + 1) * piece], non_blocking=True)
            e1.record(s_down)
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    times.sort()
    return times


for n_pieces in (1, 3, 9, 18, 36):
    for with_up in (False, True):
        for with_compute in (False, True):
            t = burst(n_pieces, with_up, with_compute)
            print(f"D2H 576 MiB in {n_pieces:2d} copies  up={int(with_up)} compute={int(with_compute)}  "
                  f"min {t[0]:6.1f} med {t[len(t) // 2]:6.1f} max {t[-1]:6.1f} ms  ({total / t[len(t) // 2] / 1e6:5.1f} GB/s median)", flush=True)
