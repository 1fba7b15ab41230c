#!/usr/bin/env python3
"""Headline benchmark: Gvoxel-iters/s of the 4-D anisotropic FISTA iteration (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one full TV iteration (all four accumulator updates + the reconstruction update, with
its convergence reductions) over the device-resident synthetic 4D-STEM cube.
  N = 1   BASELINE.json configs[1]: denoise4D FISTA, float32, 256x256x128x128 (2^30 voxels).
  N > 1   the configs[3] family, weak scaling: float32, (64*N)x512x256x256, one 64-row slab
          (2^31 voxels) per GPU, RCCL halo exchange each step (driver launches one rank per GPU).
Inputs are synthesised in HBM before the timed region (cytvdn_amd.synth on the device).
Rank 0 prints ONE JSON line.  `roofline.achieved` = algorithmic bytes per launch (19 array passes
x 4 B = 76 B per voxel) / mean duration of the fused sweep kernel, taken from HIP
events recorded around that kernel on its own stream during the timed steps.
`cpu_baseline` (N = 1 only) times the reference's own compiled kernels (oracle/_ref, kind
"reference") -- or, when they are absent, this repo's C restatement (kind "port") -- on the
host cores of the same box over a bounded sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_VOXEL_ITER = {("f32", True, 4): 76, ("f64", True, 4): 152, ("f32", False, 4): 44, ("f64", False, 4): 88,
                        ("f32", True, 3): 60, ("f32", False, 3): 36}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--shape", type=str, default=None, help="override the global shape, e.g. 64x64x128x128")
    ap.add_argument("--dtype", type=str, default="f32", choices=["f32", "f64"])
    ap.add_argument("--plain", action="store_true", help="unaccelerated iteration instead of FISTA")
    ap.add_argument("--state", type=str, default="compact", choices=["compact", "reference"],
                    help="accumulator state in HBM: compact = rotating d arrays (15 passes per 4-D FISTA iteration), "
                         "reference = the reference's (b, d) pairs (19 passes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline duration")
    return ap.parse_args()


def host_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) // int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(target_s):
    """Reference OpenMP kernels (or the port) on a bounded sample: 32x128x128x128 f32 FISTA."""
    import numpy as np
    cores = host_cores()
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    os.environ.setdefault("OMP_PROC_BIND", "spread")
    os.environ.setdefault("OMP_PLACES", "cores")
    from oracle import oracle
    from cytvdn_amd import synth
    shape = (32, 128, 128, 128)               # 2^26 voxels, 2.5 GiB of state: far beyond the host caches
    x = synth.stem4d(shape, dtype=np.float32)
    mu = np.array([1.0, 1.0, 0.5, 0.5], np.float32)
    lam = mu / np.float32(32.0)
    lam_inv, lam_mu = 1.0 / lam, (lam / mu).astype(np.float32)
    kind = "port"
    k = None
    if oracle.have_reference_kernels():
        try:
            k = oracle.load_reference_kernels()
            kind = "reference"
        except Exception:
            k = None
    if k is None:
        oracle.build()
        oracle.set_threads(int(os.environ["OMP_NUM_THREADS"]))
        k = oracle
    acc = [np.zeros_like(x) for _ in range(4)]
    dd = [np.zeros_like(x) for _ in range(4)]
    recon = x.copy()
    ratios = oracle.fista_schedule(256)

    def one(i):  # the reference's loop body, cyTVDN/cyTVDN.py:153-184
        for ax in range(4):
            k.accumulator_update_4D_FISTA(recon, acc[ax], dd[ax], ratios[i], ax, lam_inv[ax], BC_mode=2)
        k.datacube_update_4D(x, recon, acc[0], acc[1], acc[2], acc[3], lam_mu, BC_mode=2)

    one(0)                                   # first touch of the state arrays (page faults), untimed
    t0 = time.perf_counter()
    one(1)
    one(2)
    t1 = (time.perf_counter() - t0) / 2
    n = int(max(5, min(250, target_s / max(t1, 1e-3))))
    t0 = time.perf_counter()
    for i in range(3, n + 3):
        one(i)
    dt = time.perf_counter() - t0
    vox = float(np.prod(shape))
    return dict(value=vox * n / dt / 1e9, unit="Gvoxel-iters/s", cores=int(os.environ["OMP_NUM_THREADS"]), kind=kind,
                sample=f"denoise4D FISTA f32 {'x'.join(map(str, shape))} synthetic 4D-STEM, {n} iterations, "
                       f"{dt:.1f} s, OMP_NUM_THREADS={os.environ['OMP_NUM_THREADS']}")


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    cpu = None
    if world == 1 and rank == 0 and not a.no_cpu_baseline:
        cpu = cpu_baseline(a.cpu_seconds)   # before the GPU is touched: libgomp reads OMP_* at load

    import numpy as np
    import torch
    import torch.distributed as dist
    from cytvdn_amd import _lib, synth
    if not os.path.exists(_lib.LIB_PATH):       # fresh checkout: compile the HIP library in-tree (hipcc, ~20 s)
        if rank == 0:
            _lib.build()
        while not os.path.exists(_lib.LIB_PATH):
            time.sleep(1.0)
    from cytvdn_amd.engine import HipBackend, SlabLayout, SlabRunner, fista_ratios

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: cytvdn_amd has no CPU fallback")
    backend = os.environ.get("TVDN_DIST_BACKEND", "nccl")   # "gloo": rehearsal with host-staged halo rows
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()             # several ranks may then share one GPU
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            try:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
                dist.barrier()  # first collective with every rank taking part (batched P2P must not be the first one)
                torch.cuda.synchronize()
            except Exception as e:  # RCCL unusable on this node: still produce a (slower) number over gloo
                print(f"[bench] RCCL initialisation failed ({e!r}); falling back to gloo with host-staged halo rows",
                      file=sys.stderr, flush=True)
                try:
                    dist.destroy_process_group()
                except Exception:
                    pass
                backend = "gloo"
                dist.init_process_group("gloo")
                dist.barrier()
        else:
            dist.init_process_group(backend)
            dist.barrier()

    dtype = np.float32 if a.dtype == "f32" else np.float64
    if a.shape:
        shape = tuple(int(v) for v in a.shape.lower().split("x"))
    elif world == 1:
        shape = (256, 256, 128, 128)
    else:
        shape = (64 * world, 512, 256, 256)
    nd = len(shape)
    fista = not a.plain
    workload = (f"denoise{nd}D anisotropic {'FISTA' if fista else 'unaccelerated'} {a.dtype} "
                f"{'x'.join(map(str, shape))} synthetic {'4D-STEM' if nd == 4 else 'EELS'}"
                + (f", {world} slabs along axis 0" if world > 1 else ""))

    lay = SlabLayout(shape, rank, world, 2)
    be = HipBackend(lay, dtype, fista, device=local_rank, max_iters=a.steps + a.warmup, state=a.state)
    mu = np.array([1.0, 1.0, 0.5, 0.5][:nd] if nd == 4 else [1.0, 1.0, 0.5], dtype)
    lam = mu / dtype(32.0 if nd == 4 else 16.0)
    be.set_params(1.0 / lam, (lam / mu).astype(dtype))
    # synthesise this slab (halo rows included) directly in HBM: global rows g0-halo .. g1+halo
    g_first = lay.g0 - lay.halo_lo
    rows = lay.local_shape[0]
    seed = synth.SEED_4D if nd == 4 else synth.SEED_3D
    _lib.check(_lib.lib().tvdn_synth_fill(be.code, nd, _lib.shape_arr(shape), seed, g_first, rows,
                                          be.orig.data_ptr(), _lib.current_stream(local_rank)))
    be.recon[be.cur].copy_(be.orig)
    runner = SlabRunner(be)
    ratios = fista_ratios(a.steps + a.warmup)

    def step(i):
        # world > 1: edge rows first, their RCCL transfer on a side stream under the interior sweep
        runner._step(float(ratios[i]) if fista else None, i)

    def fence():
        runner.finish()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i)
    fence()
    _lib.check(_lib.lib().tvdn_ctx_timing_enable(be.ctx, 1))
    t0 = time.perf_counter()
    for i in range(a.warmup, a.warmup + a.steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    tot_ms, nl = C.c_double(), C.c_int64()
    _lib.check(_lib.lib().tvdn_ctx_timing_read(be.ctx, C.byref(tot_ms), C.byref(nl)))
    _lib.check(_lib.lib().tvdn_ctx_timing_enable(be.ctx, 0))
    # mean sweep-kernel time per iteration (with N > 1 an iteration is three launches: two edge rows + interior)
    kern_ms = tot_ms.value / max(a.steps, 1)

    t = torch.tensor([elapsed, kern_ms], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, kern_ms = float(t[0]), float(t[1])

    sums = runner.global_sums().cpu().numpy()
    total_vox = float(np.prod(shape))
    own_vox = float(lay.own_rows) * float(np.prod(shape[1:]))
    value = total_vox * a.steps / elapsed / 1e9
    bpv = BYTES_PER_VOXEL_ITER.get((a.dtype, fista, nd))
    achieved = own_vox * bpv / (kern_ms * 1e-3) / 1e9 if bpv else None
    # bytes the sweep really has to move per voxel-iteration with the chosen state representation
    item = 4 if a.dtype == "f32" else 8
    passes = 3 + nd * ((3 if a.state == "compact" else 4) if fista else 2)
    moved_bpv = passes * item
    traffic = None
    try:  # PMC-derived HBM bytes per launch, committed next to the rocprof CSVs they come from
        tr = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        traffic = tr.get(f"{workload}|{a.state}", {}).get("traffic_bytes")
    except Exception:
        pass
    if rank == 0:
        out = {
            "metric": "Gvoxel-iters/s (4D aniso FISTA)", "value": round(value, 3), "unit": "Gvoxel-iters/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": workload, "global_shape": list(shape), "bc_mode": 2,
                       "transport": ("rccl" if backend == "nccl" else backend) if world > 1 else None,
                       "state_arrays": be.n_arrays(), "state": a.state,
                       "parallelism": f"slab{world}" if world > 1 else "single"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1) if achieved else None, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
                         "traffic": traffic, "kernel": "fused_iter_kernel", "kernel_ms": round(kern_ms, 4),
                         "algorithmic_bytes_per_launch": own_vox * bpv if bpv else None,
                         "moved_bytes_per_voxel": moved_bpv,
                         "moved_GBps": round(own_vox * moved_bpv / (kern_ms * 1e-3) / 1e9, 1)},
            "cpu_baseline": cpu,
            "check": {"b_norm_last": float(sums[a.warmup + a.steps - 1, 0]),
                      "delta_last": float(sums[a.warmup + a.steps - 1, 1] / sums[a.warmup + a.steps - 1, 2])},
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
