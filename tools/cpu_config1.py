#!/usr/bin/env python3
"""BASELINE config 1 on the host cores: denoise3D FISTA f32 128x128x512, 200 iterations, with the reference's own
compiled kernels (oracle/_ref) when present, else this repo's C port.  Run BEFORE anything touches the GPU."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
n = bench.host_cores()
os.environ.setdefault("OMP_NUM_THREADS", str(n))
os.environ.setdefault("OMP_PROC_BIND", "spread")
os.environ.setdefault("OMP_PLACES", "cores")
import numpy as np
from oracle import oracle
from cytvdn_amd import synth
x = synth.eels3d((128, 128, 512))
mu = np.array([1, 1, .5], np.float32)
lam = mu / np.float32(16)
kind = "reference" if oracle.have_reference_kernels() else "port"
k = oracle.load_reference_kernels() if kind == "reference" else oracle
if kind == "port":
    oracle.set_threads(n)
acc = [np.zeros_like(x) for _ in range(3)]; dd = [np.zeros_like(x) for _ in range(3)]; recon = x.copy()
ratios = oracle.fista_schedule(200)
lam_inv, lam_mu = 1.0 / lam, (lam / mu).astype(np.float32)
t0 = time.perf_counter()
for i in range(200):
    for ax in range(3):
        k.accumulator_update_3D_FISTA(recon, acc[ax], dd[ax], ratios[i], ax, lam_inv[ax], BC_mode=2)
    k.datacube_update_3D(x, recon, acc[0], acc[1], acc[2], lam_mu, BC_mode=2)
dt = time.perf_counter() - t0
import hashlib
print(json.dumps({"config": "1: denoise3D FISTA f32 128x128x512 x200", "kind": kind, "threads": n, "seconds": round(dt, 2),
                  "Gvoxel_iters_per_s": round(x.size * 200 / dt / 1e9, 3), "recon_sha1": hashlib.sha1(recon.tobytes()).hexdigest()}))
