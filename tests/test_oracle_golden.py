"""The CPU oracle against every golden vector taken from the real reference (CPU only)."""
import hashlib

import numpy as np
import pytest

from golden_util import bits_equal, load, scalar_tol
from cytvdn_amd import synth


def _cases(name):
    d, man = load(name)
    return [(d, m) for m in man]


K, KMAN = load("kernels")
L, LMAN = load("loops")
G, GMAN = load("large")


@pytest.mark.parametrize("m", KMAN, ids=lambda m: f"{m['i']}-{m['fn']}-{m['dtype']}")
def test_kernel_case(oracle, m):
    p = f"k{m['i']:03d}_"
    fn = m["fn"]
    if fn.startswith("accumulator_update"):
        a = K[p + "a"]
        b = K[p + "b_in"].copy()
        clip = a.dtype.type(m["clip"])
        if fn.endswith("FISTA"):
            d = K[p + "d_in"].copy()
            ret = getattr(oracle, fn)(a, b, d, a.dtype.type(m["tk"]), m["ax"], clip, BC_mode=m["bc"])
            assert bits_equal(d, K[p + "d_out"])
        else:
            ret = getattr(oracle, fn)(a, b, m["ax"], clip, BC_mode=m["bc"])
        assert bits_equal(b, K[p + "b_out"])
    elif fn.startswith("datacube_update"):
        nd = int(fn[-2])
        recon = K[p + "recon_in"].copy()
        bs = [K[p + f"b{q}"] for q in range(nd)]
        ret = getattr(oracle, fn)(K[p + "orig"], recon, *bs, K[p + "lm"], BC_mode=m["bc"])
        assert bits_equal(recon, K[p + "recon_out"])
    else:
        ret = getattr(oracle, fn)(K[p + "a"], K[p + "b"])
    want = float(K[p + "ret"])
    # one thread, same visiting order, same dtype-width running sum => identical scalar
    assert (np.isnan(ret) and np.isnan(want)) or ret == want


@pytest.mark.parametrize("m", LMAN, ids=lambda m: f"{m['i']}-{m['nd']}D-{m['dtype']}-it{m['iterations']}-F{int(m['FISTA'])}-bc{m['bc']}")
def test_loop_case(oracle, m):
    p = f"l{m['i']:03d}_"
    dtype = np.dtype(m["dtype"])
    x = synth.cube(m["shape"], seed=m["seed"], dtype=dtype)
    mu = np.array(m["mu"], dtype)
    lam = None if m["lam"] is None else np.array(m["lam"], dtype)
    refd = synth.cube(m["shape"], seed=m["seed"], dtype=dtype, kind="mean") if m["with_ref"] else None
    its = m["iterations"]
    fn = oracle.denoise4D if m["nd"] == 4 else oracle.denoise3D
    out = fn(x, mu, its, FISTA=m["FISTA"], BC_mode=m["bc"], lam=lam, reference_data=refd,
             stopping_relative_change=m["stop"], quiet=True)
    assert bits_equal(out[0], L[p + "recon"])
    assert bits_equal(out[1], L[p + "b_norm"])
    assert bits_equal(out[2], L[p + "delta_recon"])
    if m["with_ref"]:
        assert bits_equal(out[3], L[p + "MSE"])


@pytest.mark.parametrize("m", GMAN, ids=lambda m: f"{m['nd']}D-{m['dtype']}-{'x'.join(map(str, m['shape']))}")
def test_large_case(oracle, m):
    dtype = np.dtype(m["dtype"])
    x = synth.cube(m["shape"], seed=m["seed"], dtype=dtype)
    assert hashlib.sha1(x.tobytes()).hexdigest() == m["input_sha1"]
    oracle.set_threads(8)  # recon does not depend on the thread count; the scalars do (a little)
    try:
        r = oracle.denoise(x, np.array(m["mu"], dtype), m["iterations"], m["FISTA"])
    finally:
        oracle.set_threads(1)
    assert hashlib.sha1(r["recon"].tobytes()).hexdigest() == m["sha1"]
    sub = tuple(slice(None, None, m["stride"]) for _ in range(m["nd"]))
    assert bits_equal(r["recon"][sub], G[f"g{m['i']}_sub"])
    n = int(np.prod(m["shape"]))
    tol = scalar_tol(dtype, n)
    np.testing.assert_allclose(r["b_norm"], G[f"g{m['i']}_b_norm"], rtol=tol)
    np.testing.assert_allclose(r["delta_recon"], G[f"g{m['i']}_delta_recon"], rtol=tol)
    # the f64 yardstick agrees with the reference's dtype-width sums to the same tolerance
    np.testing.assert_allclose(r["b_norm64"], G[f"g{m['i']}_b_norm"].astype(np.float64), rtol=tol)
