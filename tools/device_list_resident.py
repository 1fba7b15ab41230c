#!/usr/bin/env python3
"""tvdn_run with a DEVICE LIST, resident slabs (the in-process form of cyTVDN/mpi.py:314-434: one slab of axis 0 per listed GPU,
halo rows by peer copies): wall time, Gvoxel-iters/s, what the slabs' blocks are made of, the verdict of the peer-copy check
(tvdn_run_stats.peer_check; ABI 9: slabs on granules grant every device of the list access), bit-identity against the one-device
run.  A step of tools/first_node_run.sh, run twice: as the library decides, and with TVDN_VMM_PEER=0 (plain hipMalloc blocks).

    python3 tools/device_list_resident.py --devices 0,1,2,3 --shape 256x512x256x256 --iters 20 [--check]
On one GPU (rehearsal): --devices 0,0,0 with TVDN_PEER_CHECK=1 runs the check between slabs of the same device."""
import argparse, ctypes as C, hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cytvdn_amd import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--devices", default="0,0")
ap.add_argument("--shape", default="64x256x128x128")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--check", action="store_true", help="also run on the first device alone and compare the bits")
a = ap.parse_args()
devices = [int(v) for v in a.devices.split(",")]
shape = tuple(int(v) for v in a.shape.split("x"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
x = bench.synth_host(shape, devices[0])
mu = np.array([1.0, 1.0, 0.5, 0.5], np.float32)
lam = mu / np.float32(32.0)


def run(devs):
    r = _lib.RunArgs(dtype=0, ndim=4, bc_mode=2, device=devs[0], n_fista=a.iters, n_plain=0, n_devices=len(devs) if len(devs) > 1 else 0)
    for i, v in enumerate(shape):
        r.shape[i] = v
    for i, d in enumerate(devs):
        r.devices[i] = d
    for q in range(4):
        r.clip[q], r.lambda_mu[q] = float((1.0 / lam)[q]), float((lam / mu).astype(np.float32)[q])
    recon, sums, st = np.empty_like(x), np.zeros((a.iters, 3)), _lib.RunStats()
    r.data, r.recon_out, r.sums_out, r.stats = x.ctypes.data, recon.ctypes.data, sums.ctypes.data, C.addressof(st)
    t0 = time.perf_counter()
    _lib.check(_lib.lib().tvdn_run(C.byref(r)))
    return recon, time.perf_counter() - t0, st


run(devices)                                    # first call: allocations, peer access, staging lanes
recon, t, st = run(devices)
out = {"devices": devices, "distinct_devices": len(set(devices)), "shape": list(shape), "iters": a.iters, "seconds": round(t, 3),
       "value": round(float(np.prod(shape)) * a.iters / t / 1e9, 2), "unit": "Gvoxel-iters/s (whole call, PCIe included)",
       "peer_check": st.peer_check, "loop_s": round(st.loop_s, 3), "setup_s": round(st.setup_s, 3),
       "env": {k: v for k, v in os.environ.items() if k.startswith("TVDN_")},
       "mem": {d: {k: v for k, v in _lib.mem_status(d).items() if k in ("vmm_state", "canary", "faults", "blocks", "first_fault")} for d in sorted(set(devices))}}
if a.check:
    one, _, _ = run(devices[:1])
    out["bit_identical_to_one_device"] = hashlib.sha1(one.tobytes()).hexdigest() == hashlib.sha1(recon.tobytes()).hexdigest()
print(json.dumps(out), flush=True)
