#!/usr/bin/env python3
"""Headline benchmark: Gvoxel-iters/s of the 4-D anisotropic FISTA iteration (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one full TV iteration (all four accumulator updates + the reconstruction update, with
its convergence reductions) over the device-resident synthetic 4D-STEM cube.
  N = 1   BASELINE.json configs[1]: denoise4D FISTA, float32, 256x256x128x128 (2^30 voxels).
  N > 1   the configs[3] family, weak scaling: float32, (64*N)x512x256x256, one 64-row slab
          (2^31 voxels) per GPU, RCCL halo exchange each step under the interior sweep.  The driver
          launches one rank per GPU with torch.distributed.run; a plain `python bench.py --gpus N`
          starts those N ranks itself (as child processes, before anything touches a GPU).
Inputs are synthesised in HBM before the timed region (cytvdn_amd.synth on the device).
Rank 0 prints ONE JSON line.

roofline.achieved = ALGORITHMIC bytes per launch (every array of the reference's state read
once and written once = 19 passes x 4 B = 76 B per voxel for 4-D FISTA f32) / mean duration of the fused
sweep kernel, from HIP events recorded around that kernel on its own stream during the timed steps.
The compact state of this engine moves fewer bytes than that yardstick (15 passes = 60 B per voxel); what the
kernel really streams is reported next to it as roofline.moved_GBps / roofline.moved_frac.

The state lives where the product puts it: on a virtual range composed from 1 GiB physical granules (csrc/tvdn_devmem.hip;
`config.state_mem`).  On a plain hipMalloc block of that size the sweep's speed is a draw (11.0 ... 12.6 ms for config 2 by the
placement, stable for as long as the block is held: profiles/r03_placement_audition_*.jsonl, r05_placement_*.jsonl); on granules
it is 11.1 ... 11.45 ms whichever granules the block got, so there is nothing to audition and the first block is the only one.

At N = 1 the line also carries
  hipmalloc_placements  the headline workload once more on plain hipMalloc blocks (TVDN_VMM=0), the best of --audition-extra (4)
                  placements with every candidate's probe time listed: what rounds 1-4 ran on, and the lottery of THIS box
  sustained     config 2 again for >= 400 steps: what a long run sees
  also          the other single-GPU configurations on the same clock: BASELINE configs[2] (float64, unaccelerated), the
                configs[0] shape on the GPU (3-D FISTA 128x128x512), the unaccelerated f32 forms (denoise3D's default path on
                512^3, denoise4D on the config-2 cube), FISTA in float64 on the config-2 cube, ONE slab of configs[3] (66x512x256x256 local block, halo
                edges, edge rows first, halo rows refreshed by device copies of the size of the RCCL messages) = the per-GPU term
                of the weak-scaling curve; and the API level, PCIe included (never `value`): cytvdn_amd.denoise4D NumPy -> NumPy
                at 50 and 200 iterations, cytvdn_amd.denoise3D on BASELINE configs[0] (200 iterations, without and with a stopping
                rule), and tvdn_run streamed from page-locked host memory (tvdn_run_stats: passes-only rate,
                h2d / d2h GB/s, set-up and whole-call seconds, kept_in_place) -- half a rank slab of BASELINE configs[4] with the
                library's plan (every row kept in HBM and swept in place) and with every row streamed, the config-2 cube from
                host-resident state, and a cube BEYOND the resident engine (88 rows of 256 MiB planes: 330 GiB as resident state)
                that the streamed engine keeps in HBM as ten arrays and a few rings
  cpu_baseline  kind "port": the repo's reference-structured restatement (oracle/libtvdn_oracle_timed.so) called in the reference's
                order on the host cores of the same box, on config 2 itself when the host has the memory for it, in a CHILD
                process (the thread binding SURVEY 8d asks of it must not leak into the library's host threads).  Nothing built
                from the reference travels to the GPU box; `port_over_reference` is the port's speed relative to the reference's
                OWN compiled kernels (oracle/_ref), measured in the build container on identical arrays by
                tools/port_vs_reference.py and quoted from profiles/r06_port_vs_reference.json.
Every roofline object carries the per-step sweep-kernel time as mean, minimum, median and maximum (HIP events per launch).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


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
    ap.add_argument("--slab-of", type=int, default=0, metavar="N",
                    help="single GPU: run ONE interior slab of an N-slab job (halo edges, edge rows first, halo rows "
                         "refreshed by device copies) instead of the whole cube; --shape is then the GLOBAL shape")
    ap.add_argument("--audition", type=int, default=1, metavar="N",
                    help="placements of the state tried before the HEADLINE run, the fastest kept (engine.HipBackend.best_of); "
                         "1 = the first, which is what the product does: a state on granules sweeps at the same speed wherever "
                         "it lies (only with TVDN_VMM=0, on plain hipMalloc blocks, is there anything to choose)")
    ap.add_argument("--audition-extra", type=int, default=4, metavar="N",
                    help="the headline workload once more on plain hipMalloc blocks, the best of N placements (reported as "
                         "`hipmalloc_placements`; 0 = skip)")
    ap.add_argument("--no-api", action="store_true", help="skip the API-level entries (denoise4D from NumPy, streamed runs)")
    ap.add_argument("--no-also", action="store_true", help="skip the extra single-GPU configurations")
    ap.add_argument("--no-sustained", action="store_true", help="skip the >= 300-step repeat of the headline workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target duration of each timed CPU leg")
    ap.add_argument("--no-preflight", action="store_true", help="N > 1: skip the bit-exactness check of the exchange")
    ap.add_argument("--watchdog-s", type=float, default=1500.0,
                    help="N > 1: seconds after which a run that is stuck (a collective that never returns) says where, as a JSON line with "
                         "\"error\", and ends every rank -- instead of hanging until someone kills it and leaving no record (0: off)")
    ap.add_argument("--cpu-child", type=str, default=None, metavar="NPY",
                    help="internal: run ONLY the CPU baseline on the cube saved at NPY ('' = the 1/16 sample) and print its JSON")
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


def mem_available_gib():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) / 2 ** 20
                break
        else:
            return 0.0
        try:  # a cgroup limit may be tighter than the machine
            lim = open("/sys/fs/cgroup/memory.max").read().strip()
            if lim != "max":
                cur = int(open("/sys/fs/cgroup/memory.current").read())
                avail = min(avail, (int(lim) - cur) / 2 ** 30)
        except Exception:
            pass
        return avail
    except Exception:
        return 0.0


# --------------------------------------------------------------------------------------------------
# CPU baseline
# --------------------------------------------------------------------------------------------------
def cpu_baseline(target_s, x_host=None):
    """The reference's loop body (cyTVDN/cyTVDN.py:153-184) on the host cores: config 2 itself (2^30 voxels, 40 GiB
    of state in the reference's representation) when `x_host` is given, else a 1/16 sample of it."""
    import numpy as np
    from oracle import oracle
    from cytvdn_amd import synth
    cores = int(os.environ["OMP_NUM_THREADS"])
    oracle.build()
    oracle.set_threads(cores)
    if x_host is not None:
        x, what = x_host, "BASELINE config 2 in full"
        if not x.flags.writeable:      # the reference's kernels take writable buffers only, also in read-only roles
            x = np.array(x)
    else:
        x, what = synth.stem4d((16, 256, 128, 128), dtype=np.float32), "a 16-row slice (1/16) of config 2: host memory is short"
    shape = x.shape
    mu = np.array([1.0, 1.0, 0.5, 0.5], np.float32)
    lam = mu / np.float32(32.0)
    lam_inv, lam_mu = 1.0 / lam, (lam / mu).astype(np.float32)
    ratios = oracle.fista_schedule(4096)
    vox = float(np.prod(shape))

    def leg(k, budget_s):
        acc = [np.zeros_like(x) for _ in range(4)]
        dd = [np.zeros_like(x) for _ in range(4)]
        recon = x.copy()

        def one(i):
            for ax in range(4):
                k.accumulator_update_4D_FISTA(recon, acc[ax], dd[ax], ratios[i], ax, lam_inv[ax], BC_mode=2)
            k.datacube_update_4D(x, recon, acc[0], acc[1], acc[2], acc[3], lam_mu, BC_mode=2)

        one(0)                                   # first touch of the state arrays (page faults), untimed
        t0 = time.perf_counter()
        one(1)
        t1 = time.perf_counter() - t0
        n = int(max(5, min(500, budget_s / max(t1, 1e-3))))
        t0 = time.perf_counter()
        for i in range(2, n + 2):
            one(i)
        dt = time.perf_counter() - t0
        return vox * n / dt / 1e9, n, dt

    # What is timed is the tracked PORT (oracle/libtvdn_oracle_timed.so): nothing built from the reference's sources travels to
    # the GPU box (round 5 let the compiled kernels travel once and measured port / reference = 0.94-1.03 on the box
    # itself).  How the port compares with the reference's own compiled kernels is measured where the reference lives -- the
    # build container, tools/port_vs_reference.py on identical arrays -- and quoted here from the committed record.
    v, n, dt = leg(oracle.timed_kernels(), target_s)
    kind = "port"
    what_k = "oracle/libtvdn_oracle_timed.so = the reference's five passes, visiting order, dtype-width sums and serial boundary hyperslab"
    port = None
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "r06_port_vs_reference.json")))
        port = {"port_over_reference": rec["port_over_reference"], "recon_bit_identical": rec["recon_bit_identical"],
                "provenance": f"profiles/r06_port_vs_reference.json: {rec['what']}; {rec['workload']}, {rec['cores']} cores"}
        what_k += (f"; port / reference's own compiled kernels = {rec['port_over_reference']} (build container, "
                   "profiles/r06_port_vs_reference.json; 0.94-1.03 on a GPU box's host in round 5)")
    except Exception as e:
        port = {"port_over_reference": None, "provenance": f"profiles/r06_port_vs_reference.json unreadable: {e!r}"}
    out = dict(value=v, unit="Gvoxel-iters/s", cores=cores, kind=kind,
               sample=f"denoise4D FISTA f32 {'x'.join(map(str, shape))} synthetic 4D-STEM ({what}), {n} iterations, "
                      f"{dt:.1f} s, OMP_NUM_THREADS={cores}; {what_k}")
    out["port_over_reference"] = port["port_over_reference"]
    out["port_vs_reference"] = port
    return out


def cpu_baseline_in_child(target_s, x_host):
    """The CPU leg in a process of its own (thread binding: see main): the cube travels through /dev/shm when it fits there."""
    import numpy as np
    path = ""
    try:
        if x_host is not None:
            st = os.statvfs("/dev/shm")
            if st.f_bavail * st.f_frsize > x_host.nbytes + (1 << 30):
                path = f"/dev/shm/tvdn_bench_cube_{os.getpid()}.npy"
                np.save(path, x_host)
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-child", path, "--cpu-seconds", str(target_s)]
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        for line in reversed(out.stdout.splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"error": f"CPU child rc {out.returncode}: {out.stderr[-400:]}"}
    except Exception as e:
        return {"error": repr(e)}
    finally:
        if path and os.path.exists(path):
            os.remove(path)


# --------------------------------------------------------------------------------------------------
# GPU measurement of one workload
# --------------------------------------------------------------------------------------------------
def passes_algorithmic(nd, fista):
    """arrays of the reference's state, each read once and written once."""
    return (2 + 2 * nd) + (1 + 2 * nd) if fista else (2 + nd) + (1 + nd)


def passes_moved(nd, fista, state):
    return 3 + nd * ((3 if state == "compact" else 4) if fista else 2)


def workload_name(shape, dtype_name, fista, world=1, slab_of=0):
    nd = len(shape)
    s = (f"denoise{nd}D anisotropic {'FISTA' if fista else 'unaccelerated'} {dtype_name} "
         f"{'x'.join(map(str, shape))} synthetic {'4D-STEM' if nd == 4 else 'EELS'}")
    if world > 1:
        s += f", {world} slabs along axis 0"
    if slab_of:
        s += f", ONE interior slab of {slab_of} on a single GPU"
    return s


class EmulatedNeighbours:
    """Single GPU, one interior slab of a larger job: the same three launches per iteration as `step_overlapped`
    (two edge rows, then the interior) with the two halo rows refreshed on a side stream by device copies of the
    size of the RCCL messages (content: the slab's own edge rows; only the timing is meaningful)."""

    def __init__(self, be):
        import torch
        from cytvdn_amd.engine import edge_block
        self.be, self.torch, self.edge_block = be, torch, edge_block
        self.side = torch.cuda.Stream(device=be.device, priority=-1)   # as engine.SlabRunner: not on the sweeps' hardware queue

    def _step(self, tk, slot):
        torch, be = self.torch, self.be
        lay = be.layout
        lo, hi = lay.row_lo, lay.row_hi
        main = torch.cuda.current_stream(be.device)
        main.wait_stream(self.side)
        e = self.edge_block(hi - lo)
        be.step(tk, slot, rows=(lo, lo + e), accumulate=False)
        be.step(tk, slot, rows=(hi - e, hi), accumulate=True)
        done = torch.cuda.Event()
        done.record(main)
        r = be.recon_next()
        with torch.cuda.stream(self.side):
            self.side.wait_event(done)
            r[hi].copy_(r[lo], non_blocking=True)
            r[lo - 1].copy_(r[hi - 1], non_blocking=True)
        be.step(tk, slot, rows=(lo + e, hi - e), accumulate=True)
        be.flip()

    def finish(self):
        self.torch.cuda.current_stream(self.be.device).wait_stream(self.side)

    def global_sums(self):
        self.finish()
        return self.be.sums_tensor().clone()


def measure(shape, dtype_name, fista, state, steps, warmup, device, rank=0, world=1, group=None,
            slab_of=0, overlap=True, traffic_table=None, audition=1):
    """Runs warmup + steps iterations of one workload on this rank's slab; returns the result dict (same on all ranks).
    `group` is the data-plane group (RCCL; None = the default group); barriers and the timing reduction use the
    default group, which `init_groups` makes a gloo one (host memory)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from cytvdn_amd import _lib, synth
    from cytvdn_amd.engine import HipBackend, SlabLayout, SlabRunner, fista_ratios
    dtype = np.float32 if dtype_name == "f32" else np.float64
    nd = len(shape)
    if slab_of:
        lay = SlabLayout(shape, slab_of // 2, slab_of, 2)
    else:
        lay = SlabLayout(shape, rank, world, 2)
    be = HipBackend.best_of(audition, lay, dtype, fista, device=device, max_iters=steps + warmup, state=state)
    mu = np.array([1.0, 1.0, 0.5, 0.5] if nd == 4 else [1.0, 1.0, 0.5], dtype)
    lam = mu / dtype(32.0 if nd == 4 else 16.0)
    be.set_params(1.0 / lam, (lam / mu).astype(dtype))
    # synthesise this slab (halo rows included) directly in HBM: global rows g0-halo .. g1+halo
    seed = synth.SEED_4D if nd == 4 else synth.SEED_3D
    _lib.check(_lib.lib().tvdn_synth_fill(be.code, nd, _lib.shape_arr(shape), seed, lay.g0 - lay.halo_lo,
                                          lay.local_shape[0], be.orig.data_ptr(), _lib.current_stream(device)))
    be.recon[be.cur].copy_(be.orig)
    if slab_of:
        runner = EmulatedNeighbours(be)
    else:
        runner = SlabRunner(be, group)
        runner.overlap = overlap
    ratios = fista_ratios(steps + warmup)

    def step(i):
        runner._step(float(ratios[i]) if fista else None, i)

    def fence():
        runner.finish()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(warmup):
        step(i)
    fence()
    _lib.check(_lib.lib().tvdn_ctx_timing_enable(be.ctx, 1))
    t0 = time.perf_counter()
    for i in range(warmup, warmup + steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    nl = C.c_int64()
    cap = 4 * steps + 16
    each = (C.c_double * cap)()
    _lib.check(_lib.lib().tvdn_ctx_timing_read_each(be.ctx, each, cap, C.byref(nl)))
    _lib.check(_lib.lib().tvdn_ctx_timing_enable(be.ctx, 0))
    # sweep-kernel time per iteration (a slab iteration is three launches: two edge rows + interior)
    per_launch = np.array(each[:min(cap, nl.value)], np.float64)
    lps = max(1, int(nl.value // max(steps, 1)))
    per_step = per_launch[:lps * steps].reshape(steps, lps).sum(axis=1) if per_launch.size >= lps * steps else per_launch
    kern_ms = float(per_launch.sum()) / max(steps, 1)
    if world > 1:
        t = torch.tensor([elapsed, kern_ms], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)      # control-plane group (gloo, host memory)
        elapsed, kern_ms = float(t[0]), float(t[1])
    sums = runner.global_sums().cpu().numpy()      # all-reduced over the data-plane group when world > 1
    own_vox = float(lay.own_rows) * float(np.prod(shape[1:]))
    total_vox = own_vox if slab_of else float(np.prod(shape))
    item = 4 if dtype_name == "f32" else 8
    bpv = passes_algorithmic(nd, fista) * item
    moved_bpv = passes_moved(nd, fista, state) * item
    achieved = own_vox * bpv / (kern_ms * 1e-3) / 1e9
    moved = own_vox * moved_bpv / (kern_ms * 1e-3) / 1e9
    name = workload_name(shape, dtype_name, fista, world, slab_of)
    traffic = None
    if traffic_table:
        traffic = traffic_table.get(f"{name}|{state}", {}).get("traffic_bytes")
    last = warmup + steps - 1
    res = {
        "value": round(total_vox * steps / elapsed / 1e9, 3), "unit": "Gvoxel-iters/s",
        "ms_per_step": round(elapsed / steps * 1e3, 4), "dtype": dtype_name,
        "config": {"workload": name, "global_shape": list(shape), "local_block": list(lay.local_shape), "bc_mode": 2,
                   "state_arrays": be.n_arrays(), "state": state, "state_mem": be.state_mem,
                   "placement_audition_ms": getattr(be, "audition", []),
                   "timed_loop": "engine.SlabRunner: one tvdn_iterate_fused launch + fold per step, driven from Python on a state "
                                 "resident in HBM (the loop denoise3D/4D run by default is tvdn_run's, measured whole-call by the "
                                 "'denoise4D NumPy -> NumPy' entries of also[])",
                   "parallelism": f"slab{world}" if world > 1 else ("one slab of %d" % slab_of if slab_of else "single")},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "kernel": "fused_iter_kernel", "kernel_ms": round(kern_ms, 4),
                     "kernel_ms_min": round(float(per_step.min()), 4), "kernel_ms_median": round(float(np.median(per_step)), 4),
                     "kernel_ms_max": round(float(per_step.max()), 4),
                     "launches_per_step": int(nl.value // max(steps, 1)),
                     "traffic_source": "profiles/traffic.json: PMC passes (2 x FETCH_SIZE + WRITE_SIZE) of the same "
                                       "workload, collected in separate rocprofv3 --pmc runs (not in this run)",
                     "basis": f"SURVEY 8d algorithmic bytes: {passes_algorithmic(nd, fista)} array passes x {item} B = "
                              f"{bpv} B per voxel-iteration (the reference's state, each array read once and written once)",
                     "algorithmic_bytes_per_launch": own_vox * bpv,
                     "moved_bytes_per_voxel": moved_bpv, "moved_bytes_per_launch": own_vox * moved_bpv,
                     "moved_GBps": round(moved, 1), "moved_frac": round(moved / HBM_PEAK_GBS, 4)},
        "check": {"b_norm_last": float(sums[last, 0]), "delta_last": float(sums[last, 1] / sums[last, 2])},
    }
    del runner, be
    torch.cuda.empty_cache()
    return res


# --------------------------------------------------------------------------------------------------
# API-level entries: what a caller of denoise4D / tvdn_run sees, PCIe included (never `value` of the headline)
# --------------------------------------------------------------------------------------------------
def synth_host(shape, device=0):
    """The synthetic cube in ordinary host memory: synthesised on the device in slices, brought down by the library."""
    import numpy as np
    import torch
    from cytvdn_amd import _lib, synth
    nd = len(shape)
    x = np.empty(shape, np.float32)
    plane = int(np.prod(shape[1:]))
    step = max(1, min(shape[0], (1 << 30) // (plane * 4)))
    buf = torch.empty((step,) + tuple(shape[1:]), dtype=torch.float32, device=f"cuda:{device}")
    for r in range(0, shape[0], step):
        n = min(step, shape[0] - r)
        _lib.check(_lib.lib().tvdn_synth_fill(0, nd, _lib.shape_arr(shape), synth.SEED_4D if nd == 4 else synth.SEED_3D,
                                              r, n, buf.data_ptr(), _lib.current_stream(device)))
        torch.cuda.synchronize()
        _lib.check(_lib.lib().tvdn_copy_to_host(C.c_void_p(x[r:r + n].ctypes.data), C.c_void_p(buf.data_ptr()),
                                                n * plane * 4, device))
    del buf
    torch.cuda.empty_cache()
    return x


def api_denoise4d(x, iters_list, device=0):
    """cytvdn_amd.denoise4D from a NumPy cube to a NumPy cube (upload, iterations, download: the reference's call,
    cyTVDN/cyTVDN.py:19-247), wall time of the whole call.  The first call of a process also pins the staging buffers."""
    import numpy as np
    import cytvdn_amd as tv
    mu = np.array([1.0, 1.0, 0.5, 0.5], np.float32)
    vox = float(x.size)
    out = []
    tv.denoise4D(x, mu, 4, quiet=True, device=device)                 # first call of the process: untimed
    for n in iters_list:
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            r = tv.denoise4D(x, mu, n, quiet=True, device=device)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out.append({"config": {"workload": f"cytvdn_amd.denoise4D NumPy -> NumPy, FISTA f32 {'x'.join(map(str, x.shape))}, "
                                           f"{n} iterations, PCIe transfers included (best of 2 calls)"},
                    "value": round(vox * n / best / 1e9, 3), "unit": "Gvoxel-iters/s", "iterations": n,
                    "whole_call_s": round(best, 4), "pcie_inclusive": True,
                    "check": {"b_norm_last": float(r[1][-1])}})
        del r
    return out


def api_denoise3d_config1(device=0):
    """BASELINE configs[0] as a user calls it -- cytvdn_amd.denoise3D(cube, mu, 200, FISTA=True) on the 128 x 128 x 512 EELS cube, NumPy
    to NumPy (cyTVDN/cyTVDN.py:250-435) -- without and with a stopping rule that never fires (what `denoise3D` is usually given:
    the resident tvdn_run then looks at every iteration's sums one iteration behind, csrc/tvdn_run.hip).  Best of 3 calls."""
    import numpy as np
    import cytvdn_amd as tv
    from cytvdn_amd import synth
    shape, n = (128, 128, 512), 200
    x = synth.cube(shape, dtype=np.float32)
    mu = np.array([1.0, 1.0, 0.5], np.float32)
    tv.denoise3D(x, mu, 4, FISTA=True, quiet=True, device=device)
    out = []
    for tag, kw in (("", {}), (", stopping_relative_change set (never met)", {"stopping_relative_change": 1e-30})):
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            r = tv.denoise3D(x, mu, n, FISTA=True, quiet=True, device=device, **kw)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out.append({"config": {"workload": f"cytvdn_amd.denoise3D NumPy -> NumPy, FISTA f32 128x128x512 (BASELINE configs[0]), {n} iterations{tag}, "
                                           "PCIe transfers included (best of 3 calls)"},
                    "value": round(float(x.size) * n / best / 1e9, 3), "unit": "Gvoxel-iters/s", "iterations": n,
                    "whole_call_s": round(best, 5), "pcie_inclusive": True, "check": {"b_norm_last": float(r[1][-1])}})
    return out


def api_streamed(shape, rows, k, iters, what, x=None, device=0, force_stream=False, resident=-1):
    """tvdn_run (the C entry, csrc/tvdn_stream.hip) on a cube whose state stays in page-locked HOST memory: `rows`-row
    chunks, `k` iterations per PCIe round trip (-1 / -1: the library's own choice).  Reports the rate of the passes, the
    PCIe rates beside it (tvdn_run_stats), set-up and whole-call time.  Skips, visibly, when the host cannot hold the state."""
    import numpy as np
    import torch
    from cytvdn_amd import _lib
    nd = len(shape)
    entry = {"config": {"workload": what, "global_shape": list(shape), "engine": "tvdn_run streamed (csrc/tvdn_stream.hip)"}}
    a = _lib.RunArgs(dtype=0, ndim=nd, bc_mode=2, device=device, n_fista=iters, n_plain=0, stream_rows=rows, stream_k=k,
                     stream_resident=resident)
    for i, v in enumerate(shape):
        a.shape[i] = int(v)
    if force_stream and rows < 0:
        # the library's own (rows, k, resident rows) for a STREAMED run of this cube, even where the whole state would fit
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        free_b, total_b = torch.cuda.mem_get_info(device)
        entry["hbm_free_GiB_at_plan"] = round((free_b + _lib.state_kept_bytes(device)) / 2 ** 30, 2)
        po = _lib.StreamPlanOut()
        _lib.check(_lib.lib().tvdn_stream_plan(C.byref(a), 0, C.byref(po)))
        a.stream_rows, a.stream_k, a.stream_resident = int(po.rows), int(po.k), int(po.resident_rows)
        if iters < 0:                        # -n: n passes of the planned depth (whole PCIe round trips)
            iters = -iters * int(po.k)
            a.n_fista = iters
    mu = np.array([1.0, 1.0, 0.5, 0.5], np.float32)
    lam = mu / np.float32(32.0)
    for q in range(nd):
        a.clip[q] = float((1.0 / lam)[q])
        a.lambda_mu[q] = float((lam / mu).astype(np.float32)[q])
    need, avail = C.c_int64(), C.c_int64()
    rc = _lib.lib().tvdn_stream_host_need(C.byref(a), C.byref(need), C.byref(avail))      # (an upper bound: no rows kept in HBM)
    if a.stream_rows > 0 and a.stream_resident != 0:
        po = _lib.StreamPlanOut()
        if _lib.lib().tvdn_stream_plan(C.byref(a), 0, C.byref(po)) == 0 and a.stream_resident > 0:
            need.value = int(need.value * (1.0 - min(a.stream_resident, shape[0]) / shape[0]))
            rc = 0 if need.value <= 0.8 * avail.value else rc
    # the cube itself and the result are ordinary arrays of this process on top of what the library pins
    if rc != 0 or need.value > 0.8 * mem_available_gib() * 2 ** 30:
        entry["skipped"] = (f"host memory: the run page-locks {need.value / 2 ** 30:.0f} GiB, the host offers "
                            f"{min(avail.value / 2 ** 30, mem_available_gib()):.0f} GiB")
        return entry
    if x is None:
        x = synth_host(shape, device)
    recon = np.empty_like(x)
    sums = np.zeros((iters, 3))
    st = _lib.RunStats()
    a.data, a.recon_out, a.sums_out, a.stats = x.ctypes.data, recon.ctypes.data, sums.ctypes.data, C.addressof(st)
    torch.cuda.empty_cache()
    _lib.lib().tvdn_wait_background()        # an earlier streamed call's page-locked memory is back with the OS (not this call's time)
    t0 = time.perf_counter()
    _lib.check(_lib.lib().tvdn_run(C.byref(a)))
    whole = time.perf_counter() - t0
    t0 = time.perf_counter()
    _lib.lib().tvdn_wait_background()        # ... and this call's own: returned in the background, after the call
    released = time.perf_counter() - t0
    vox = float(np.prod(shape))
    entry.update({
        "value": round(vox * iters / st.loop_s / 1e9, 3), "unit": "Gvoxel-iters/s", "iterations": iters,
        "value_whole_call": round(vox * iters / whole / 1e9, 3),
        "stream_rows": st.stream_rows, "stream_k": st.stream_k, "resident_rows": st.resident_rows, "passes": st.n_passes,
        "passes_s": round(st.loop_s, 3), "setup_s": round(st.setup_s, 3), "whole_call_s": round(whole, 3),
        "h2d_GBps": round(st.h2d_bytes / st.loop_s / 1e9, 2), "d2h_GBps": round(st.d2h_bytes / st.loop_s / 1e9, 2),
        "h2d_GB": round(st.h2d_bytes / 1e9, 1), "d2h_GB": round(st.d2h_bytes / 1e9, 1),
        "kept_in_place": st.kept_in_place, "pinned_host_GiB": round(need.value / 2 ** 30, 1), "pcie_inclusive": True,
        "background_release_s": round(released, 3),
        "check": {"b_norm_last": float(sums[-1, 0])}})
    if st.n_passes > 1 and 0 < st.first_pass_s < st.loop_s and 0 < st.first_pass_iters < iters:
        # the first pass is the one the host state is page-locked under (seconds that depend on the box: huge pages at hand
        # or not) and, on a fresh cube, uploads the data term only; the passes after it are the steady state of a long run
        entry["first_pass_s"] = round(st.first_pass_s, 3)
        entry["value_later_passes"] = round(vox * (iters - st.first_pass_iters) / (st.loop_s - st.first_pass_s) / 1e9, 3)
    del recon
    return entry


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children (nothing here has touched a GPU)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] --gpus %d without a launcher: starting the ranks with: %s" % (a.gpus, " ".join(cmd)),
          file=sys.stderr, flush=True)
    return subprocess.call(cmd)


def init_groups(local_rank):
    """Control plane over gloo (decisions, timing reductions, barriers); data plane over RCCL when EVERY rank can
    bring it up -- the decision is taken collectively, so no rank is left behind in another backend.
    Returns (data_group_or_None, transport, fallback_flag)."""
    import torch
    import torch.distributed as dist
    want = os.environ.get("TVDN_DIST_BACKEND", "nccl")
    dist.init_process_group("gloo")
    dist.barrier()
    if want != "nccl":
        return None, "gloo", False
    ok, g, err = 1, None, ""
    try:
        # RCCL's own stream from the high-priority pool: a hardware-queue class the sweeps (default stream) are not in, so the
        # send/recv kernels neither queue behind the interior sweep nor wait for its workgroups to be dispatched first
        opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
        g = dist.new_group(backend="nccl", device_id=torch.device("cuda", local_rank), pg_options=opts)
        dist.barrier(group=g)   # first collective with every rank taking part (batched P2P must not be the first one)
        torch.cuda.synchronize()
    except Exception as e:
        ok, err = 0, repr(e)
    t = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if int(t[0]) == 1:
        return g, "rccl", False
    print(f"[bench] RCCL unusable on at least one rank ({err or 'another rank failed'}); ALL ranks fall back to gloo "
          "with host-staged halo rows", file=sys.stderr, flush=True)
    return None, "gloo", True


class Watchdog:
    """N > 1 only.  RCCL has not met a second GPU in any round: if a collective never returns, every rank is stuck in C++ with no way
    out and the run leaves nothing behind.  A timer thread (it runs while the main thread waits with the GIL released) then prints
    ONE JSON line from rank 0 -- the metric's name, value null, the stage the run was in -- on the real stdout and ends the process."""

    def __init__(self, seconds, rank, world, a):
        import threading
        self.stage, self.rank, self.world, self.a, self.fd = "start", rank, world, a, None
        # (the other ranks go a little later: a rank that exits first has the launcher end rank 0 before its line is out)
        self.timer = threading.Timer(seconds + (0.0 if rank == 0 else 15.0), self.fire) if seconds > 0 and world > 1 else None
        if self.timer:
            self.timer.daemon = True
            self.timer.start()

    def fire(self):
        if self.rank == 0:
            line = json.dumps({"metric": "Gvoxel-iters/s (4D aniso FISTA)", "value": None, "unit": "Gvoxel-iters/s", "n_gpus": self.world,
                               "steps": self.a.steps, "warmup": self.a.warmup, "higher_is_better": True, "scaling": "weak",
                               "error": f"watchdog: no result after {self.a.watchdog_s:.0f} s; the run was in stage '{self.stage}' "
                                        "(a collective that never returned?); TVDN_DIST_BACKEND=gloo measures with host-staged halo rows"})
            os.write(self.fd if self.fd is not None else 1, (line + "\n").encode())
        os._exit(4)

    def done(self):
        if self.timer:
            self.timer.cancel()


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(a))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # libgomp reads these when it is first loaded (numpy/torch pull it in): set them before any import.  The thread BINDING
    # SURVEY 8d asks of the CPU leg (OMP_PROC_BIND=spread, OMP_PLACES=cores) is set in a child process that runs nothing but
    # that leg: with it in this process libgomp pins the MAIN thread to one core, every host thread the library starts
    # (staging lanes, page-touching, copies) inherits that mask, and the API-level entries measure one core's memcpy.
    os.environ.setdefault("OMP_NUM_THREADS", str(host_cores()))
    if a.cpu_child is not None:
        os.environ.setdefault("OMP_PROC_BIND", "spread")
        os.environ.setdefault("OMP_PLACES", "cores")
        import numpy as np
        x = np.load(a.cpu_child, mmap_mode="r") if a.cpu_child else None
        print(json.dumps(cpu_baseline(a.cpu_seconds, None if x is None else np.ascontiguousarray(x))), flush=True)
        return

    import numpy as np
    import torch
    import torch.distributed as dist
    from cytvdn_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):       # fresh checkout: compile the HIP library in-tree (hipcc, ~30 s)
        if local_rank == 0:
            _lib.build()
        while not os.path.exists(_lib.LIB_PATH):
            time.sleep(1.0)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: cytvdn_amd has no CPU fallback")

    want_backend = os.environ.get("TVDN_DIST_BACKEND", "nccl")
    if want_backend != "nccl":
        local_rank %= torch.cuda.device_count()             # rehearsal: several ranks may then share one GPU
    torch.cuda.set_device(local_rank)
    group, transport, fallback, preflight = None, None, False, None
    overlap = True
    dog = Watchdog(a.watchdog_s, rank, world, a)
    if world > 1:
        # libraries chat on stdout while the groups come up ("[Gloo] Rank 0 is connected to ..."): stdout is for the
        # one JSON line, so file descriptor 1 points at stderr until the measurement starts
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        dog.fd = saved_stdout
        os.dup2(2, 1)
        dog.stage = "process groups (gloo control plane, RCCL data plane)"
        group, transport, fallback = init_groups(local_rank)
        dog.stage = "pre-flight: exchange self-check on plain memory"
        if not a.no_preflight:
            from cytvdn_amd.distributed import selfcheck_exchange
            preflight = selfcheck_exchange(group=group, device=local_rank)
            if not preflight["overlap"]:
                overlap = False                  # keep measuring, visibly, with the blocking exchange
            if not preflight["blocking"] and rank == 0:
                print(f"[bench] exchange self-check FAILED on {transport}: {preflight}", file=sys.stderr, flush=True)
            # The slabs of the measurement live on granules of HIP virtual memory (csrc/tvdn_devmem.hip), which no transport has been
            # handed on hardware before the first node: the same check with its small states on granules.  Wrong bits or an error
            # there, with the plain check green, puts the measurement on plain hipMalloc memory (TVDN_VMM=0) and says so in the line.
            if preflight["blocking"] and os.environ.get("TVDN_VMM", "1") != "0":
                # (on_granules=True: the check's states are forced onto granules, says what they really were -- "state_mem" --
                # and counts anything else as failed: a check that ran on plain memory proves nothing about granules)
                dog.stage = "pre-flight: exchange self-check with the states on granules"
                pg = selfcheck_exchange(group=group, device=local_rank, on_granules=True)
                preflight["on_granules"] = {k: pg[k] for k in ("overlap", "blocking", "error", "state_mem")}
                if not pg["blocking"]:
                    os.environ["TVDN_VMM"] = "0"
                    preflight["state_mem_fallback"] = "plain: the exchange self-check failed with the states on granules"
                    if rank == 0:
                        print(f"[bench] exchange self-check on granules FAILED on {transport}: {pg}; measuring on plain memory", file=sys.stderr, flush=True)
                elif not pg["overlap"]:
                    overlap = False
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
        dog.fd = None

    dtype_name = a.dtype
    if a.shape:
        shape = tuple(int(v) for v in a.shape.lower().split("x"))
    elif world == 1 and not a.slab_of:
        shape = (256, 256, 128, 128)
    else:
        shape = (64 * max(world, a.slab_of), 512, 256, 256)
    fista = not a.plain
    try:
        traffic_table = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except Exception:
        traffic_table = None

    dog.stage = f"measurement ({a.warmup} + {a.steps} steps, transport {transport}, overlap {overlap})"
    main_res = measure(shape, dtype_name, fista, a.state, a.steps, a.warmup, local_rank, rank, world, group,
                       a.slab_of, overlap, traffic_table, a.audition)
    dog.stage = "after the measurement"

    also = None
    headline = world == 1 and not a.slab_of and not a.shape and dtype_name == "f32" and fista
    if headline and not a.no_also:
        also = []
        for shp, dn, fi, st_, wu, slab in (((256, 256, 128, 128), "f64", False, 20, 3, 0),     # BASELINE configs[2]
                                           ((128, 128, 512), "f32", True, 200, 20, 0),         # configs[0] shape on the GPU
                                           ((512, 512, 256, 256), "f32", True, 10, 2, 8),      # one slab of configs[3]
                                           # the unaccelerated f32 forms (round 6; denoise3D's DEFAULT is FISTA=False, cyTVDN.py:253)
                                           ((512, 512, 512), "f32", False, 100, 10, 0),
                                           ((256, 256, 128, 128), "f32", False, 20, 3, 0),
                                           # the fourth corner of dtype x mode on the config-2 cube: FISTA in float64 (120 GiB of state)
                                           ((256, 256, 128, 128), "f64", True, 12, 3, 0)):
            try:
                r = measure(shp, dn, fi, a.state, st_, wu, local_rank, slab_of=slab, traffic_table=traffic_table,
                            audition=a.audition)
                r["steps"], r["warmup"] = st_, wu
                also.append(r)
            except Exception as e:   # e.g. a smaller GPU: say so instead of failing the headline
                also.append({"config": {"workload": workload_name(shp, dn, fi, 1, slab)}, "error": repr(e)})

    best_placement = None
    if headline and a.audition_extra > 1:
        prev_vmm = os.environ.get("TVDN_VMM")
        try:   # the same steps on plain hipMalloc blocks, the fastest of N placements: what rounds 1-4 measured, and this box's lottery
            os.environ["TVDN_VMM"] = "0"
            r = measure(shape, dtype_name, fista, a.state, a.steps, a.warmup, local_rank, traffic_table=traffic_table,
                        audition=a.audition_extra)
            best_placement = {"value": r["value"], "ms_per_step": r["ms_per_step"], "candidates": a.audition_extra,
                              "state_mem": r["config"]["state_mem"],
                              "placement_audition_ms": r["config"]["placement_audition_ms"],
                              "kernel_ms": r["roofline"]["kernel_ms"], "frac": r["roofline"]["frac"],
                              "moved_frac": r["roofline"]["moved_frac"]}
        except Exception as e:
            best_placement = {"error": repr(e)}
        finally:
            if prev_vmm is None:
                os.environ.pop("TVDN_VMM", None)
            else:
                os.environ["TVDN_VMM"] = prev_vmm

    sustained = None
    if headline and not a.no_sustained:
        try:
            from cytvdn_amd.driver import _audition_candidates
            n_sus = max(400, a.steps)      # a run long enough for the product's own audition rule to apply
            sustained = measure(shape, dtype_name, fista, a.state, n_sus, a.warmup, local_rank, traffic_table=traffic_table,
                                audition=_audition_candidates(n_sus))
            sustained["steps"], sustained["warmup"] = n_sus, a.warmup
        except Exception as e:
            sustained = {"error": repr(e)}

    api = None
    if headline and rank == 0 and not a.no_api:
        api = []
        x2 = x_half = None
        try:
            x2 = synth_host(shape, local_rank)
            api.extend(api_denoise4d(x2, (50, 200), local_rank))                               # NumPy -> NumPy, PCIe included
        except Exception as e:
            api.append({"config": {"workload": "cytvdn_amd.denoise4D NumPy -> NumPy"}, "error": repr(e)})
        try:
            api.extend(api_denoise3d_config1(local_rank))                                      # BASELINE configs[0] on the GPU, whole call
        except Exception as e:
            api.append({"config": {"workload": "cytvdn_amd.denoise3D NumPy -> NumPy (configs[0])"}, "error": repr(e)})
        half = (64, 1024, 256, 256)
        # What the resident measurements left behind (torch's cache, the library's kept state block) goes back to the driver
        # BEFORE the 16 GiB input of the streamed runs is synthesised: the driver clears freed HBM in the background, and a
        # 245 GiB hipMalloc issued right behind a 60 GiB hipFree waits for that (2-6 s of "set-up" that no first call of a
        # process would see).
        try:
            _lib.lib().tvdn_release_cache()
            torch.cuda.empty_cache()
            x_half = synth_host(half, local_rank)
        except Exception:
            x_half = None
        for shp, rows, k, iters, what, xin, resident in (
                # (the tallest device block first: the library keeps it, and the later runs carve theirs out of it instead of
                #  releasing and re-allocating a quarter of a TB, which stalls for seconds while the driver clears the memory)
                (half, -1, -1, 80, "BASELINE config 5 planes on one GPU, OUT OF CORE: HALF a rank slab (64x1024x256x256 of the 128 rows a "
                                   "rank of 8 holds; a whole one needs 320 GiB of page-locked host memory, this box's control group allows "
                                   "300), forced through the streamed tvdn_run with the library's own plan (rows, k, rows resident in HBM)",
                 None, -1),
                (half, -1, -1, -3, "the same half rank slab with EVERY row streamed (stream_resident = 0, the library's deepest k, three "
                                   "passes of it, chained): the PCIe-bound regime a whole rank slab of config 5 is in", None, 0),
                (shape, 16, 128, 256, "BASELINE config 2 cube advanced from HOST-resident state (streamed tvdn_run, 16-row chunks, 128 "
                                      "iterations per PCIe round trip, no rows resident)", "config2", 0)):
            try:
                if xin == "config2":
                    x_half = None
                    xin = x2
                else:
                    if x_half is None:
                        x_half = synth_host(shp, local_rank)
                    xin = x_half
                api.append(api_streamed(shp, rows, k, iters, what, xin, local_rank, force_stream=True, resident=resident))
            except Exception as e:
                api.append({"config": {"workload": what}, "error": repr(e)})
        x_half = None
        x2 = None
        # A cube that does NOT fit resident (15 arrays of 22 GiB) but whose 10 kept arrays and a few rings do: the library keeps every
        # row in HBM and sweeps it in place (the lean layout of the streamed engine) -- nothing page-locked, PCIe for the cube itself
        try:
            big = (88, 1024, 256, 256)
            if mem_available_gib() >= 60.0:
                # (the device block the last entry kept is re-dealt at this run's size: the granules it has stay, the few
                #  missing ones are created -- csrc/tvdn_devmem.hip dev_resize)
                torch.cuda.empty_cache()
                api.append(api_streamed(big, -1, -1, 80, "88x1024x256x256 (22 GiB per array: 330 GiB of resident state, more than the HBM) with "
                                                         "the library's own plan: every row kept in HBM (10 arrays + rings), swept in place",
                                        None, local_rank, force_stream=True, resident=-1))
        except Exception as e:
            api.append({"config": {"workload": "88x1024x256x256 beyond the resident engine"}, "error": repr(e)})

    cpu = None
    if headline and rank == 0 and not a.no_cpu_baseline:
        x_host = None
        if mem_available_gib() >= 56.0:         # 10 arrays x 4 GiB + the input + slack
            buf = torch.empty(shape, dtype=torch.float32, device="cuda")
            from cytvdn_amd import synth
            _lib.check(_lib.lib().tvdn_synth_fill(0, 4, _lib.shape_arr(shape), synth.SEED_4D, 0, shape[0],
                                                  buf.data_ptr(), _lib.current_stream(local_rank)))
            x_host = buf.cpu().numpy()
            del buf
            torch.cuda.empty_cache()
        cpu = cpu_baseline_in_child(a.cpu_seconds, x_host)
        del x_host

    if rank == 0:
        out = {"metric": "Gvoxel-iters/s (4D aniso FISTA)", "value": main_res["value"], "unit": "Gvoxel-iters/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": main_res["ms_per_step"],
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype_name,
               "data": "synthetic", "config": main_res["config"], "roofline": main_res["roofline"],
               "cpu_baseline": cpu, "check": main_res["check"]}
        if world > 1:
            out["config"]["transport"] = transport
            out["config"]["overlap"] = bool(overlap)
            out["transport_fallback"] = bool(fallback)
            out["preflight"] = preflight
        out["config"]["audition_rule"] = ("none: the state is composed from 1 GiB physical granules (state_mem), on which the sweep's time "
                                          "does not depend on the draw; hipmalloc_placements shows plain blocks on the same box "
                                          "(the product auditions those: 3 placements from 400 iterations, 4 from 800)")
        if best_placement is not None:
            out["hipmalloc_placements"] = best_placement
        if sustained is not None:
            out["sustained"] = sustained
        if also is not None or api:
            out["also"] = (also or []) + (api or [])
        print(json.dumps(out), flush=True)
    if world > 1:
        dog.stage = "final barrier"
        dist.barrier()
        dog.done()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
