#!/usr/bin/env python3
"""BUILD CONTAINER ONLY: how fast is the tracked CPU port (oracle/libtvdn_oracle_timed.so, what bench.py's
`cpu_baseline` times on the GPU box) next to the reference's own compiled kernels (oracle/_ref, built here from
/root/reference/cyTVDN/{anisotropic,utils}.c by `make -C oracle ref`)?  Same arrays, same loop body
(cyTVDN/cyTVDN.py:153-184), alternating rounds, best of N per side.  Nothing built from the reference's sources travels to
the GPU box, so this ratio is measured here and committed:

    python tools/port_vs_reference.py > profiles/r06_port_vs_reference.json      (bench.py quotes the ratio from there)
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    cores = len(os.sched_getaffinity(0))
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    os.environ.setdefault("OMP_PROC_BIND", "spread")
    os.environ.setdefault("OMP_PLACES", "cores")
    import numpy as np
    from oracle import oracle
    from cytvdn_amd import synth
    if not os.path.isdir("/root/reference/cyTVDN"):
        raise SystemExit("needs /root/reference (the build container)")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle", "ref"])
    oracle.build()
    oracle.set_threads(cores)
    shape = (64, 64, 128, 128)                      # 2^26 voxels: 1/16 of config 2, 2.75 GiB of state
    x = synth.stem4d(shape, dtype=np.float32)
    mu = np.array([1.0, 1.0, 0.5, 0.5], np.float32)
    lam = mu / np.float32(32.0)
    lam_inv, lam_mu = 1.0 / lam, (lam / mu).astype(np.float32)
    ratios = oracle.fista_schedule(64)
    vox = float(np.prod(shape))
    sides = {"port": oracle.timed_kernels(), "reference": oracle.load_reference_kernels()}
    best = {k: 0.0 for k in sides}
    rounds = {k: [] for k in sides}
    recon_out = {}
    n_it = 6
    for rnd in range(4):
        for name, k in sides.items():
            acc = [np.zeros_like(x) for _ in range(4)]
            dd = [np.zeros_like(x) for _ in range(4)]
            recon = x.copy()

            def one(i):
                for ax in range(4):
                    k.accumulator_update_4D_FISTA(recon, acc[ax], dd[ax], ratios[i], ax, lam_inv[ax], BC_mode=2)
                k.datacube_update_4D(x, recon, acc[0], acc[1], acc[2], acc[3], lam_mu, BC_mode=2)

            one(0)
            t0 = time.perf_counter()
            for i in range(1, 1 + n_it):
                one(i)
            dt = time.perf_counter() - t0
            v = vox * n_it / dt / 1e9
            rounds[name].append(round(v, 4))
            best[name] = max(best[name], v)
            recon_out[name] = recon
    same = recon_out["port"].tobytes() == recon_out["reference"].tobytes()
    print(json.dumps({
        "what": "port (oracle/libtvdn_oracle_timed.so) vs the reference's own compiled kernels (oracle/_ref), build container",
        "workload": f"denoise4D FISTA f32 {'x'.join(map(str, shape))} synthetic 4D-STEM, {n_it} iterations per round, 4 alternating rounds",
        "cores": cores, "unit": "Gvoxel-iters/s", "best": {k: round(v, 4) for k, v in best.items()}, "rounds": rounds,
        "port_over_reference": round(best["port"] / best["reference"], 3), "recon_bit_identical": bool(same),
    }))


if __name__ == "__main__":
    main()
