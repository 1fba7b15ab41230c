#!/usr/bin/env python3
"""Config-5 rehearsal on ONE GPU: N ranks (gloo, all on cuda:0), each with its slab of the cube in pinned host memory,
streamed through the GPU by the library's own loop, k rows of state swapped between neighbouring ranks per pass
(`denoise_slabs(..., staged=(rows, k))`).  Reported separately from bench.py: this mode is PCIe-bound and the ranks
share one GPU and one PCIe link here.

    python tools/bench_staged_slabs.py --shape 256x256x128x128 --ranks 2 --rows 16 --k 32 --iters 64
"""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _default_host_limit():
    """Unless the caller says otherwise, a measurement may page-lock at most half of what the host has available
    (TVDN_HOST_LIMIT is honoured by the streamed runs): a mistyped shape gets an error, not the box."""
    if "TVDN_HOST_LIMIT" in os.environ:
        return
    try:
        with open("/proc/meminfo") as f:
            kb = next(int(line.split()[1]) for line in f if line.startswith("MemAvailable:"))
        os.environ["TVDN_HOST_LIMIT"] = str(kb * 1024 // 2)
    except (OSError, StopIteration, ValueError):
        os.environ["TVDN_HOST_LIMIT"] = "32G"


_default_host_limit()


def worker(rank, world, port, a, q):
    import numpy as np
    import torch
    import torch.distributed as dist
    from cytvdn_amd import _lib, synth
    from cytvdn_amd.distributed import slab_rows
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = rank % torch.cuda.device_count() if a.gpu_per_rank else 0
    torch.cuda.set_device(dev)
    shape = tuple(int(v) for v in a.shape.split("x"))
    nd = len(shape)
    dt = np.dtype(np.float32)
    g0, g1 = slab_rows(shape, rank, world)
    own = np.empty((g1 - g0,) + shape[1:], dt)
    step = max(1, (1 << 28) // int(np.prod(shape[1:])))
    buf = torch.empty((step,) + shape[1:], dtype=torch.float32, device="cuda")
    for r in range(g0, g1, step):
        n = min(step, g1 - r)
        _lib.check(_lib.lib().tvdn_synth_fill(0, nd, _lib.shape_arr(shape), synth.SEED_4D if nd == 4 else synth.SEED_3D,
                                              r, n, buf.data_ptr(), _lib.current_stream(dev)))
        own[r - g0:r - g0 + n] = buf[:n].cpu().numpy()
    del buf
    mu = np.array([1.0, 1.0, 0.5, 0.5][:nd] if nd == 4 else [1.0, 1.0, 0.5], dt)
    lam = mu / dt.type(32.0 if nd == 4 else 16.0)
    from cytvdn_amd.distributed import denoise_slabs
    dist.barrier()
    t0 = time.perf_counter()
    # the whole call of every rank: page-locking its slab, the passes of the library's streamed loop (tvdn_run with a
    # tvdn_slab_io, csrc/tvdn_stream.hip run_streamed_rank) with the k-row state swaps between them, the release
    # rows kept resident in HBM: as many as fit when every rank has its own GPU; none when the ranks share one (each would
    # count the same free memory as its own) unless --resident says how many
    res = a.resident if a.resident is not None else (-1 if a.gpu_per_rank else 0)
    recon, b_norm, _ = denoise_slabs(own, shape, mu, a.iters, FISTA=True, lam=lam, device=dev, staged=(a.rows, a.k, res))
    dist.barrier()
    dt_run = time.perf_counter() - t0
    if rank == 0:
        q.put({"seconds": round(dt_run, 3), "b_norm_last": float(b_norm[a.iters - 1])})
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="256x256x128x128")
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--rows", type=int, default=16)
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--iters", type=int, default=64)
    ap.add_argument("--gpu-per-rank", action="store_true", help="rank r on GPU r (a multi-GPU node) instead of all ranks on GPU 0")
    ap.add_argument("--resident", type=int, default=None, help="rows of every slab kept in HBM (-1: as many interior rows as fit)")
    a = ap.parse_args()
    import numpy as np
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    t0 = time.perf_counter()
    procs = [ctx.Process(target=worker, args=(r, a.ranks, port, a, q)) for r in range(a.ranks)]
    for p in procs:
        p.start()
    res = q.get()
    for p in procs:
        p.join()
    shape = tuple(int(v) for v in a.shape.split("x"))
    vox = float(np.prod(shape))
    res.update({"metric": "Gvoxel-iters/s (4D aniso FISTA, staged slabs, %d ranks %s over gloo)" % (a.ranks, "one GPU each" if a.gpu_per_rank else "on ONE GPU"),
                "value": round(vox * a.iters / res["seconds"] / 1e9, 2), "unit": "Gvoxel-iters/s", "shape": list(shape),
                "ranks": a.ranks, "chunk_rows": a.rows, "k": a.k, "iters": a.iters, "resident_rows_asked": a.resident,
                "state_GiB_compact": round(15 * vox * 4 / 2 ** 30, 1), "wall_s": round(time.perf_counter() - t0, 1)})
    print(json.dumps(res))


if __name__ == "__main__":
    main()
