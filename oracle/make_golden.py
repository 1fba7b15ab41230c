#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (build container only).

    OMP_NUM_THREADS=1 python oracle/make_golden.py

What runs: the reference's own compiled kernels (oracle/_ref, compiled by oracle/Makefile from
the C files the reference ships) and the reference's own Python driver, imported from
/root/reference/cyTVDN/cyTVDN.py (see oracle.load_reference_driver).  This script holds no
reference code: it only prepares seeded inputs, calls the reference, and stores what came back.

What is stored: inputs and outputs (data), plus a JSON manifest of the call parameters.
One OpenMP thread is forced so that the reference's dtype-width running sums (b_norm,
delta_recon, MSE) are reproducible; recon/acc/d do not depend on the thread count.

Fixture files:
  kernels.npz   single-call cases for every kernel-level function
  loops.npz     denoise3D / denoise4D runs over the flag matrix (small shapes, full outputs)
  large.npz     config-1 shape (128,128,512) FISTA x200 and a 4-D 20^2x32^2 run:
                SHA-1 of recon, a strided subsample, and the scalar traces
"""
import hashlib
import json
import os
import sys

os.environ["OMP_NUM_THREADS"] = "1"

import numpy as np  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import oracle  # noqa: E402
from cytvdn_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def rnd(rng, shape, dtype, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(dtype)


def kernel_cases(ref):
    rng = np.random.default_rng(20261003)
    man, data = [], {}

    def put(i, **arrs):
        for k, v in arrs.items():
            data[f"k{i:03d}_{k}"] = v

    i = 0
    shapes4 = [(6, 5, 7, 9), (1, 4, 1, 5), (2, 2, 2, 2), (3, 4, 5, 1)]
    shapes3 = [(5, 6, 11), (3, 1, 4), (1, 1, 7), (4, 3, 2)]
    for dtype in (np.float32, np.float64):
        for shape in shapes4 + shapes3:
            nd = len(shape)
            for ax in range(nd):
                for bc in (0, 1, 2):
                    if bc == 1 and shape[ax] < 2:
                        continue  # a[1] does not exist: out-of-bounds read upstream
                    for fista in (False, True):
                        a = rnd(rng, shape, dtype, 3.0)
                        b = rnd(rng, shape, dtype, 0.7)
                        clip = dtype(0.9)
                        b_in = b.copy()
                        if fista:
                            d = rnd(rng, shape, dtype, 0.7)
                            d_in = d.copy()
                            tk = dtype(0.37)
                            fn = getattr(ref, f"accumulator_update_{nd}D_FISTA")
                            ret = fn(a, b, d, tk, ax, clip, BC_mode=bc)
                            put(i, a=a, b_in=b_in, d_in=d_in, b_out=b, d_out=d, ret=np.float64(ret))
                            man.append(dict(i=i, fn=f"accumulator_update_{nd}D_FISTA", ax=ax, bc=bc,
                                            clip=float(clip), tk=float(tk), dtype=np.dtype(dtype).name))
                        else:
                            fn = getattr(ref, f"accumulator_update_{nd}D")
                            ret = fn(a, b, ax, clip, BC_mode=bc)
                            put(i, a=a, b_in=b_in, b_out=b, ret=np.float64(ret))
                            man.append(dict(i=i, fn=f"accumulator_update_{nd}D", ax=ax, bc=bc,
                                            clip=float(clip), dtype=np.dtype(dtype).name))
                        i += 1
            for bc in (0, 2):
                orig = rnd(rng, shape, dtype, 3.0)
                recon = rnd(rng, shape, dtype, 3.0)
                bs = [rnd(rng, shape, dtype, 0.7) for _ in range(nd)]
                lm = (np.array([1 / 32, 1 / 40, 1 / 64, 1 / 33][:nd])).astype(dtype)
                r_in = recon.copy()
                fn = getattr(ref, f"datacube_update_{nd}D")
                ret = fn(orig, recon, *bs, lm, BC_mode=bc)
                put(i, orig=orig, recon_in=r_in, recon_out=recon, lm=lm, ret=np.float64(ret),
                    **{f"b{q}": bs[q] for q in range(nd)})
                man.append(dict(i=i, fn=f"datacube_update_{nd}D", bc=bc, dtype=np.dtype(dtype).name))
                i += 1
            a = rnd(rng, shape, dtype, 3.0)
            b = rnd(rng, shape, dtype, 3.0)
            ret = getattr(ref, f"sum_square_error_{nd}D")(a, b)
            put(i, a=a, b=b, ret=np.float64(ret))
            man.append(dict(i=i, fn=f"sum_square_error_{nd}D", dtype=np.dtype(dtype).name))
            i += 1
    # NaN propagation through clipval (anisotropic.c:2423-2437)
    a = rnd(rng, (3, 4, 5), np.float32, 3.0)
    a[1, 2, 3] = np.nan
    b = rnd(rng, (3, 4, 5), np.float32, 0.7)
    b_in = b.copy()
    ret = ref.accumulator_update_3D(a, b, 1, np.float32(0.9), BC_mode=2)
    put(i, a=a, b_in=b_in, b_out=b, ret=np.float64(ret))
    man.append(dict(i=i, fn="accumulator_update_3D", ax=1, bc=2, clip=float(np.float32(0.9)), dtype="float32",
                    note="nan"))
    data["manifest"] = np.array(json.dumps(man))
    return data


def loop_cases(drv):
    man, data = [], {}
    i = 0

    def run(nd, dtype, iterations, fista, bc, lam_mode, with_ref, stop, shape, seed):
        nonlocal i
        x = synth.cube(shape, seed=seed, dtype=dtype)
        mu = np.array([1.0, 1.0, 0.5, 0.5][:nd] if nd == 4 else [1.0, 1.0, 0.5], dtype=dtype)
        lam = None
        if lam_mode == "explicit":
            lam = (mu / np.array([40.0, 33.0, 64.0, 100.0][:nd])).astype(dtype)
        refd = synth.cube(shape, seed=seed, dtype=dtype, kind="mean") if with_ref else None
        kw = dict(BC_mode=bc, lam=lam, quiet=True, reference_data=refd, stopping_relative_change=stop)
        x_in = x.copy()
        if nd == 4:
            out = drv.denoise4D(x, mu, iterations, FISTA=fista, **kw)
        else:
            out = drv.denoise3D(x, mu, iterations, FISTA=fista, **kw)
        assert np.array_equal(x, x_in), "reference mutated its input"
        data[f"l{i:03d}_recon"] = out[0]
        data[f"l{i:03d}_b_norm"] = out[1]
        data[f"l{i:03d}_delta_recon"] = out[2]
        if with_ref:
            data[f"l{i:03d}_MSE"] = out[3]
        man.append(dict(i=i, nd=nd, dtype=np.dtype(dtype).name, iterations=iterations, FISTA=fista, bc=bc,
                        lam=None if lam is None else [float(v) for v in lam], mu=[float(v) for v in mu],
                        with_ref=with_ref, stop=stop, shape=list(shape), seed=seed))
        i += 1

    s4, s3 = (6, 5, 8, 12), (7, 6, 16)
    for dtype in (np.float32, np.float64):
        for nd, shape in ((4, s4), (3, s3)):
            for bc in (2, 0):
                run(nd, dtype, 9, True, bc, "none", False, None, shape, 11)
                run(nd, dtype, 9, False, bc, "none", False, None, shape, 12)
                run(nd, dtype, [5, 4], True, bc, "explicit", True, None, shape, 13)
            # stopping_relative_change hit mid-run (plain; FISTA; hybrid whose FISTA phase breaks early)
            run(nd, dtype, 40, False, 2, "none", False, 0.02, shape, 14)
            run(nd, dtype, 40, True, 2, "none", True, 0.02, shape, 15)
            run(nd, dtype, [30, 6], True, 2, "none", False, 0.03, shape, 16)
            # odd shapes / unit axes
            run(nd, dtype, 6, True, 2, "explicit", False, None, (3, 1, 5, 7)[:nd] if nd == 4 else (1, 5, 9), 17)
    data["manifest"] = np.array(json.dumps(man))
    return data


def large_cases(drv):
    man, data = [], {}
    specs = [
        dict(nd=3, shape=(128, 128, 512), iterations=200, FISTA=True, seed=synth.SEED_3D, dtype="float32",
             mu=[1.0, 1.0, 0.5]),
        dict(nd=4, shape=(20, 20, 32, 32), iterations=20, FISTA=True, seed=synth.SEED_4D, dtype="float32",
             mu=[1.0, 1.0, 0.5, 0.5]),
        dict(nd=4, shape=(20, 20, 32, 32), iterations=12, FISTA=False, seed=synth.SEED_4D, dtype="float64",
             mu=[1.0, 1.0, 0.5, 0.5]),
    ]
    for i, sp in enumerate(specs):
        dtype = np.dtype(sp["dtype"])
        x = synth.cube(sp["shape"], seed=sp["seed"], dtype=dtype)
        mu = np.array(sp["mu"], dtype=dtype)
        fn = drv.denoise4D if sp["nd"] == 4 else drv.denoise3D
        recon, bn, dl = fn(x, mu, sp["iterations"], FISTA=sp["FISTA"], quiet=True)
        sub = tuple(slice(None, None, 7) for _ in range(sp["nd"]))
        data[f"g{i}_sub"] = recon[sub].copy()
        data[f"g{i}_b_norm"] = bn
        data[f"g{i}_delta_recon"] = dl
        sp = dict(sp, i=i, shape=list(sp["shape"]), sha1=hashlib.sha1(recon.tobytes()).hexdigest(),
                  input_sha1=hashlib.sha1(x.tobytes()).hexdigest(), stride=7)
        man.append(sp)
        print("large", sp)
    data["manifest"] = np.array(json.dumps(man))
    return data


def main():
    ref = oracle.load_reference_kernels()
    drv = oracle.load_reference_driver()
    os.makedirs(OUT, exist_ok=True)
    for name, fn, arg in (("kernels", kernel_cases, ref), ("loops", loop_cases, drv), ("large", large_cases, drv)):
        d = fn(arg)
        p = os.path.join(OUT, name + ".npz")
        np.savez_compressed(p, **d)
        print(name, len(d) - 1, "arrays", os.path.getsize(p) // 1024, "KiB")


if __name__ == "__main__":
    main()
