"""Parity of the HIP path (through the C ABI) with the reference's golden vectors and the oracle.

Bars: recon / accumulators bit-exact (same IEEE operations in the same order, no FMA); the scalar
traces are f64 tree sums here and dtype-width running sums upstream, compared with the size-aware
tolerance of golden_util.scalar_tol and, tightly (1e-12), with the oracle's f64 yardsticks.
"""
import hashlib

import numpy as np
import pytest

from golden_util import bits_equal, load, scalar_tol

pytestmark = pytest.mark.gpu

K, KMAN = load("kernels")
L, LMAN = load("loops")
G, GMAN = load("large")


@pytest.fixture(scope="module")
def tv():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import cytvdn_amd
    return cytvdn_amd


def _close(got, want, dtype, n):
    tol = scalar_tol(dtype, n)
    if np.isnan(want):
        assert np.isnan(got)
    else:
        assert got == pytest.approx(want, rel=tol, abs=1e-30)


@pytest.mark.parametrize("m", KMAN, ids=lambda m: f"{m['i']}-{m['fn']}-{m['dtype']}")
def test_kernel_golden(tv, m):
    p = f"k{m['i']:03d}_"
    fn = m["fn"]
    if fn.startswith("accumulator_update"):
        a = K[p + "a"].copy()
        b = K[p + "b_in"].copy()
        clip = a.dtype.type(m["clip"])
        if fn.endswith("FISTA"):
            d = K[p + "d_in"].copy()
            ret = getattr(tv, fn)(a, b, d, a.dtype.type(m["tk"]), m["ax"], clip, BC_mode=m["bc"])
            assert bits_equal(d, K[p + "d_out"])
        else:
            ret = getattr(tv, fn)(a, b, m["ax"], clip, BC_mode=m["bc"])
        assert bits_equal(b, K[p + "b_out"])
        assert bits_equal(a, K[p + "a"])  # read-only role untouched
        n = a.size
    elif fn.startswith("datacube_update"):
        nd = int(fn[-2])
        recon = K[p + "recon_in"].copy()
        bs = [K[p + f"b{q}"].copy() for q in range(nd)]
        ret = getattr(tv, fn)(K[p + "orig"].copy(), recon, *bs, K[p + "lm"], BC_mode=m["bc"])
        assert bits_equal(recon, K[p + "recon_out"])
        n = recon.size
    else:
        a = K[p + "a"].copy()
        ret = getattr(tv, fn)(a, K[p + "b"].copy())
        n = a.size
    _close(ret, float(K[p + "ret"]), m["dtype"], n)


def _run_loop(tv, m, x, refd):
    dtype = np.dtype(m["dtype"])
    mu = np.array(m["mu"], dtype)
    lam = None if m["lam"] is None else np.array(m["lam"], dtype)
    fn = tv.denoise4D if m["nd"] == 4 else tv.denoise3D
    return fn(x, mu, m["iterations"], FISTA=m["FISTA"], BC_mode=m["bc"], lam=lam, reference_data=refd,
              stopping_relative_change=m["stop"], quiet=True)


@pytest.mark.parametrize("m", LMAN, ids=lambda m: f"{m['i']}-{m['nd']}D-{m['dtype']}-it{m['iterations']}-F{int(m['FISTA'])}-bc{m['bc']}")
def test_loop_golden(tv, m):
    from cytvdn_amd import synth
    p = f"l{m['i']:03d}_"
    dtype = np.dtype(m["dtype"])
    x = synth.cube(m["shape"], seed=m["seed"], dtype=dtype)
    x0 = x.copy()
    refd = synth.cube(m["shape"], seed=m["seed"], dtype=dtype, kind="mean") if m["with_ref"] else None
    out = _run_loop(tv, m, x, refd)
    assert bits_equal(x, x0), "input mutated"
    assert bits_equal(out[0], L[p + "recon"])
    n = x.size
    tol = scalar_tol(dtype, n)
    for k, name in ((1, "b_norm"), (2, "delta_recon")):
        want = L[p + name]
        assert out[k].dtype == dtype and out[k].shape == want.shape
        # early stopping: identical zero tails (same iteration count)
        assert np.array_equal(out[k] == 0, want == 0), name
        np.testing.assert_allclose(out[k], want, rtol=tol)
    if m["with_ref"]:
        assert len(out) == 4
        np.testing.assert_allclose(out[3], L[p + "MSE"], rtol=tol)
    else:
        assert len(out) == 3


@pytest.mark.parametrize("m", GMAN, ids=lambda m: f"{m['nd']}D-{m['dtype']}-{'x'.join(map(str, m['shape']))}")
def test_large_golden(tv, m):
    """config-1 shape (128,128,512) x 200 FISTA iterations, and two 4-D runs: SHA-1 of recon."""
    from cytvdn_amd import synth
    dtype = np.dtype(m["dtype"])
    x = synth.cube(m["shape"], seed=m["seed"], dtype=dtype)
    assert hashlib.sha1(x.tobytes()).hexdigest() == m["input_sha1"]
    fn = tv.denoise4D if m["nd"] == 4 else tv.denoise3D
    recon, bn, dl = fn(x, np.array(m["mu"], dtype), m["iterations"], FISTA=m["FISTA"], quiet=True)
    assert hashlib.sha1(recon.tobytes()).hexdigest() == m["sha1"]
    tol = scalar_tol(dtype, x.size)
    np.testing.assert_allclose(bn, G[f"g{m['i']}_b_norm"], rtol=tol)
    np.testing.assert_allclose(dl, G[f"g{m['i']}_delta_recon"], rtol=tol)


SHAPES = [
    ((9, 6, 10, 16), np.float32), ((5, 4, 6, 8), np.float64), ((12, 10, 32), np.float32), ((6, 7, 10), np.float64),
    ((40, 3, 5, 8), np.float32),   # several marching chunks
    ((70, 4, 4), np.float32), ((3, 5, 7, 9), np.float32), ((2, 1, 1, 4), np.float64), ((1, 1, 1, 1), np.float32),
    ((1, 6, 8), np.float64),
    # C-rows of 8 / 16 / 32 lanes with A divisible by the rows a wavefront holds (wave-level sharing of the A neighbours)
    ((5, 8, 3, 32), np.float32), ((4, 4, 5, 64), np.float32), ((3, 6, 4, 128), np.float32), ((3, 4, 3, 64), np.float64),
    ((11, 16, 2, 32), np.float32),
]


@pytest.mark.parametrize("shape,dtype", SHAPES, ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else np.dtype(v).name)
@pytest.mark.parametrize("bc", [2, 0])
@pytest.mark.parametrize("mode", ["fista", "plain", "hybrid"])
@pytest.mark.parametrize("state", ["compact", "reference"])
def test_fused_vs_oracle(tv, oracle, monkeypatch, shape, dtype, bc, mode, state):
    """Seeded inputs at shapes the oracle finishes in milliseconds: vector and scalar paths,
    chunk seams, unit axes, both boundary conditions, both accumulator-state representations
    (compact d-rotation and the reference's (b, d) pairs); f64 scalars to 1e-12."""
    from cytvdn_amd import engine, synth
    monkeypatch.setattr(engine, "DEFAULT_STATE", state)
    dtype = np.dtype(dtype)
    nd = len(shape)
    x = synth.cube(shape, seed=77, dtype=dtype) + dtype.type(0.25)
    mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dtype)
    its = {"fista": 7, "plain": 7, "hybrid": [4, 3]}[mode]
    if mode == "hybrid" and shape[0] % 2:
        its = [0, 5] if shape[0] % 3 == 0 else [1, 4]      # hybrid runs that start (almost) unaccelerated
    fista = mode != "plain"
    fn = tv.denoise4D if nd == 4 else tv.denoise3D
    recon, bn, dl = fn(x, mu, its, FISTA=fista, BC_mode=bc, quiet=True)
    ref = oracle.denoise(x, mu, its, fista, BC_mode=bc)
    assert bits_equal(recon, ref["recon"])
    np.testing.assert_allclose(bn.astype(np.float64), ref["b_norm64"].astype(dtype).astype(np.float64), rtol=1e-6 if dtype == np.float32 else 1e-12)
    want = (ref["delta64"].astype(dtype) / ref["rnorm64"].astype(dtype)).astype(np.float64)
    np.testing.assert_allclose(dl.astype(np.float64), want, rtol=1e-6 if dtype == np.float32 else 1e-12)


@pytest.mark.parametrize("patch", ["2,4", "4,2", "8,16", "0"])
def test_fused_patch_order_of_tiles(tv, oracle, monkeypatch, patch):
    """The fused sweep's workgroups may walk the cross-section in 2-D patches (A-rows x tiles) instead of plain order
    (long A-rows: config-4 planes).  A pure permutation of who computes what: the bits must not move."""
    from cytvdn_amd import synth
    monkeypatch.setenv("TVDN_PATCH", patch)
    for shape, dtype, bc in (((5, 4, 64, 128), np.float32, 2), ((3, 8, 128, 256), np.float32, 0), ((4, 4, 64, 128), np.float64, 2),
                             ((6, 8, 32, 512), np.float32, 2)):
        dt = np.dtype(dtype)
        x = synth.cube(shape, seed=61, dtype=dt) + dt.type(0.25)
        mu = np.array([1.0, 0.8, 0.5, 0.6], dt)
        recon, bn, dl = tv.denoise4D(x, mu, [3, 2], BC_mode=bc, quiet=True)
        ref = oracle.denoise(x, mu, [3, 2], True, BC_mode=bc)
        assert bits_equal(recon, ref["recon"]), (shape, patch)
        np.testing.assert_allclose(bn.astype(np.float64), ref["b_norm64"].astype(dt).astype(np.float64),
                                   rtol=1e-6 if dt == np.float32 else 1e-12)


@pytest.mark.parametrize("block,xcd,patch", [("256", "1", None), ("256", "0", None), ("128", "1", None), ("128", "0", None),
                                             ("256", "0", "32,8"), ("128", "0", "16,8"), ("256", "1", "0")])
def test_fused_launch_shapes_do_not_move_the_bits(tv, oracle, monkeypatch, block, xcd, patch):
    """Workgroup size (256 / 128 threads), XCD remap on / off and the patch walk are launch decisions the library takes by
    shape (csrc/tvdn_fused.hip); forced here in every combination on shapes with long A-rows (patches of 32 x 8 tiles
    apply), short A-rows and a 3-D cube: who computes which voxel changes, the voxel's arithmetic does not."""
    from cytvdn_amd import synth
    monkeypatch.setenv("TVDN_BLOCK", block)
    monkeypatch.setenv("TVDN_XCD", xcd)
    if patch is not None:
        monkeypatch.setenv("TVDN_PATCH", patch)
    for shape, dtype in (((3, 32, 128, 256), np.float32), ((2, 64, 64, 128), np.float64), ((5, 4, 16, 64), np.float32),
                         ((9, 24, 512), np.float32)):
        dt = np.dtype(dtype)
        nd = len(shape)
        x = synth.cube(shape, seed=67, dtype=dt) + dt.type(0.25)
        mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
        fn = tv.denoise4D if nd == 4 else tv.denoise3D
        recon, bn, dl = fn(x, mu, [2, 2], FISTA=True, quiet=True)
        ref = oracle.denoise(x, mu, [2, 2], True)
        assert bits_equal(recon, ref["recon"]), (shape, block, xcd, patch)
        # f64 data, a million terms: the oracle's own serial running sum carries ~1e-11
        np.testing.assert_allclose(bn.astype(np.float64), ref["b_norm64"].astype(dt).astype(np.float64),
                                   rtol=1e-6 if dt == np.float32 else 1e-9)


def test_device_tensors_in_place(tv, oracle):
    """Kernel-level calls on torch CUDA tensors update HBM in place."""
    import torch
    rng = np.random.default_rng(5)
    a = rng.standard_normal((4, 5, 6, 8)).astype(np.float32)
    b = rng.standard_normal((4, 5, 6, 8)).astype(np.float32)
    d = rng.standard_normal((4, 5, 6, 8)).astype(np.float32)
    ta, tb, td = (torch.from_numpy(v.copy()).cuda() for v in (a, b, d))
    ret = tv.accumulator_update_4D_FISTA(ta, tb, td, 0.3, 2, np.float32(0.8))
    b2, d2 = b.copy(), d.copy()
    nT, n64 = oracle.acc_update(a, b2, d2, np.float32(0.3), 2, np.float32(0.8), 2)
    assert bits_equal(tb.cpu().numpy(), b2) and bits_equal(td.cpu().numpy(), d2)
    assert ret == pytest.approx(n64, rel=1e-12)


def test_nonfinite_and_strided_views(tv, oracle):
    """NaN propagates through clip as upstream; non-contiguous writable views are accepted
    by the kernel-level functions."""
    rng = np.random.default_rng(6)
    a = rng.standard_normal((6, 5, 4)).astype(np.float64)
    a[2, 3, 1] = np.nan
    big = rng.standard_normal((6, 5, 8)).astype(np.float64)
    b = big[:, :, ::2]
    assert not b.flags["C_CONTIGUOUS"]
    b_ref = np.ascontiguousarray(b).copy()
    tv.accumulator_update_3D(a, b, 1, 0.7, BC_mode=0)
    oracle.acc_update(a, b_ref, None, 0.0, 1, 0.7, 0)
    assert bits_equal(np.ascontiguousarray(b), b_ref)
    assert np.isnan(b[2, 3, 1]) and np.isnan(b[2, 4, 1])


@pytest.mark.parametrize("shape,dtype,its,fista,stop,with_ref", [
    ((6, 5, 8, 12), np.float32, 7, True, None, False),
    ((7, 6, 16), np.float64, [4, 3], True, None, True),
    ((6, 5, 8, 12), np.float32, 40, False, 0.02, False),
    ((5, 3, 7, 9), np.float64, [30, 6], True, 0.03, True),
])
def test_tvdn_run_host_entry(tv, shape, dtype, its, fista, stop, with_ref):
    """The C whole-loop entry point (tvdn_run, host pointers) gives what denoise3D/4D give, bit for bit,
    including its own float64 tk recurrence and the early-stop bookkeeping."""
    import ctypes as C
    from cytvdn_amd import _lib, synth
    dt = np.dtype(dtype)
    nd = len(shape)
    x = synth.cube(shape, seed=31, dtype=dt)
    refd = synth.cube(shape, seed=31, dtype=dt, kind="mean") if with_ref else None
    mu = np.array([1.0, 1.0, 0.5, 0.5][:nd], dt)
    lam = mu / dt.type(32.0 if nd == 4 else 16.0)
    fn = tv.denoise4D if nd == 4 else tv.denoise3D
    want = fn(x, mu, its, FISTA=fista, stopping_relative_change=stop, reference_data=refd, quiet=True)
    n_f, n_p = (its if isinstance(its, list) else ((its, 0) if fista else (0, its)))
    n = n_f + n_p
    a = _lib.RunArgs(dtype=_lib.dtype_code(dt), ndim=nd, bc_mode=2, device=0, n_fista=n_f, n_plain=n_p,
                     use_stop=int(stop is not None), stop=float(stop or 0.0))
    for i, s in enumerate(shape):
        a.shape[i] = s
    for q in range(nd):
        a.clip[q] = float((1.0 / lam)[q])
        a.lambda_mu[q] = float((lam / mu).astype(dt)[q])
    recon = np.empty_like(x)
    sums = np.zeros((n, 3))
    mse = np.zeros(n + 1)
    ran = C.c_int32(0)
    a.data, a.recon_out, a.sums_out = x.ctypes.data, recon.ctypes.data, sums.ctypes.data
    if with_ref:
        a.reference, a.mse_out = refd.ctypes.data, mse.ctypes.data
    a.iters_run = C.addressof(ran)
    _lib.check(_lib.lib().tvdn_run(C.byref(a)))
    assert bits_equal(recon, want[0])
    assert ran.value == int(np.count_nonzero(want[2]))
    assert bits_equal(np.where(sums[:, 2] != 0, sums[:, 0], 0).astype(dt), want[1])
    with np.errstate(divide="ignore", invalid="ignore"):
        dl = np.where(sums[:, 2] != 0, sums[:, 1].astype(dt) / sums[:, 2].astype(dt), 0).astype(dt)
    assert bits_equal(dl, want[2])
    if with_ref:
        assert bits_equal(mse.astype(dt), want[3])


@pytest.mark.parametrize("shape,dtype,its,fista,stop,with_ref,bc,devices", [
    ((23, 5, 8, 12), np.float32, 7, True, None, False, 2, [0, 0]),          # two slabs on one GPU: peer copies degenerate to D2D
    ((23, 5, 8, 12), np.float32, [4, 3], True, None, True, 2, [0, 0, 0]),   # hybrid + reference_data over three slabs
    ((20, 6, 16), np.float64, 6, True, None, False, 0, [0, 0]),             # periodic ring of two slabs
    ((31, 3, 7, 9), np.float32, 40, False, 0.02, False, 2, [0, 0, 0, 0]),   # global stopping rule over four slabs (scalar packs)
    ((7, 4, 8, 8), np.float32, 5, True, None, False, 2, [0] * 7),           # one row per slab
    ((64, 6, 8, 16), np.float32, 6, True, None, False, 2, [0, 0]),          # 32-row slabs: 8-row edge blocks + interior
])
def test_tvdn_run_device_list(tv, shape, dtype, its, fista, stop, with_ref, bc, devices):
    """tvdn_run with a device list: one slab of axis 0 per entry, edge blocks first, halo rows moved by peer copies on
    a copy stream under the interior sweep -- bit-identical to the single-device run (and so to the reference),
    including the global sums, the stopping iteration and the MSE trace."""
    import ctypes as C
    from cytvdn_amd import _lib, synth
    dt = np.dtype(dtype)
    nd = len(shape)
    x = synth.cube(shape, seed=37, dtype=dt) + dt.type(0.25)
    refd = synth.cube(shape, seed=37, dtype=dt, kind="mean") if with_ref else None
    mu = np.array([1.0, 1.0, 0.5, 0.5][:nd], dt)
    lam = mu / dt.type(32.0 if nd == 4 else 16.0)
    fn = tv.denoise4D if nd == 4 else tv.denoise3D
    want = fn(x, mu, its, FISTA=fista, stopping_relative_change=stop, reference_data=refd, BC_mode=bc, quiet=True)
    n_f, n_p = (its if isinstance(its, list) else ((its, 0) if fista else (0, its)))
    n = n_f + n_p
    a = _lib.RunArgs(dtype=_lib.dtype_code(dt), ndim=nd, bc_mode=bc, device=0, n_fista=n_f, n_plain=n_p,
                     use_stop=int(stop is not None), stop=float(stop or 0.0), n_devices=len(devices))
    for i, d in enumerate(devices):
        a.devices[i] = d
    for i, s in enumerate(shape):
        a.shape[i] = s
    for q in range(nd):
        a.clip[q] = float((1.0 / lam)[q])
        a.lambda_mu[q] = float((lam / mu).astype(dt)[q])
    recon = np.empty_like(x)
    sums = np.zeros((n, 3))
    mse = np.zeros(n + 1)
    ran = C.c_int32(0)
    a.data, a.recon_out, a.sums_out = x.ctypes.data, recon.ctypes.data, sums.ctypes.data
    if with_ref:
        a.reference, a.mse_out = refd.ctypes.data, mse.ctypes.data
    a.iters_run = C.addressof(ran)
    _lib.check(_lib.lib().tvdn_run(C.byref(a)))
    assert bits_equal(recon, want[0])
    assert ran.value == int(np.count_nonzero(want[2]))
    tol = 1e-6 if dt == np.float32 else 1e-12            # sums are added slab by slab: same value, other rounding
    np.testing.assert_allclose(np.where(sums[:, 2] != 0, sums[:, 0], 0).astype(dt), want[1], rtol=tol)
    with np.errstate(divide="ignore", invalid="ignore"):
        dl = np.where(sums[:, 2] != 0, sums[:, 1].astype(dt) / sums[:, 2].astype(dt), 0).astype(dt)
    np.testing.assert_allclose(dl, want[2], rtol=tol)
    assert np.array_equal(dl == 0, want[2] == 0)
    if with_ref:
        np.testing.assert_allclose(mse.astype(dt), want[3], rtol=tol)
    # argument errors of the device list
    a.n_devices = 40
    assert _lib.lib().tvdn_run(C.byref(a)) == -1
    a.n_devices, a.devices[0] = 1, 99
    assert _lib.lib().tvdn_run(C.byref(a)) == -1 and b"devices[0]" in _lib.lib().tvdn_last_error()


def test_subnormals_signed_zeros_and_infinities(tv, oracle):
    """f32 subnormals are kept (no flush-to-zero), -0.0 survives, +-inf/NaN propagate as on the CPU:
    the arithmetic contract of SURVEY Appendix A, on the one-pass kernels and on the fused sweep."""
    rng = np.random.default_rng(11)
    shape = (6, 5, 4, 8)
    tiny = np.float32(1e-41)                       # subnormal
    a = (rng.standard_normal(shape) * tiny * 50).astype(np.float32)
    b = (rng.standard_normal(shape) * tiny * 20).astype(np.float32)
    d = (rng.standard_normal(shape) * tiny * 20).astype(np.float32)
    a[0, 0, 0, 0], b[0, 0, 0, 0] = np.float32(-0.0), np.float32(-0.0)
    assert np.any((a != 0) & (np.abs(a) < np.finfo(np.float32).tiny))
    b2, d2 = b.copy(), d.copy()
    tv.accumulator_update_4D_FISTA(a, b, d, np.float32(0.3), 3, np.float32(tiny * 7))
    oracle.acc_update(a, b2, d2, np.float32(0.3), 3, np.float32(tiny * 7), 2)
    assert bits_equal(b, b2) and bits_equal(d, d2)
    assert np.any((b != 0) & (np.abs(b) < np.finfo(np.float32).tiny)), "subnormal results must not be flushed"
    # fused loop on subnormal-scale data with infinities sprinkled in
    x = (rng.standard_normal(shape) * tiny * 1000).astype(np.float32)
    x[2, 3, 1, 5] = np.inf
    x[4, 1, 2, 0] = -np.inf
    mu = np.array([1, 1, .5, .5], np.float32)
    got = tv.denoise4D(x, mu, 4, quiet=True)
    ref = oracle.denoise(x, mu, 4, True)
    assert bits_equal(got[0], ref["recon"])
    assert np.isnan(got[0]).any() and np.isfinite(got[0]).any()


def test_plain_c_caller_of_the_abi(tv, tmp_path):
    """examples/tvdn_run_demo.c: a C program linked against libtvdn_hip.so (no Python in the data path) must
    produce the recon the Python driver produces for the same input."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "tvdn_run_demo")
    subprocess.check_call(["gcc", "-O2", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "tvdn_run_demo.c"),
                           "-L" + os.path.join(root, "cytvdn_amd"), "-ltvdn_hip",
                           "-Wl,-rpath," + os.path.join(root, "cytvdn_amd"), "-lm", "-o", exe])
    shape, iters = (7, 6, 8, 16), 6
    out = subprocess.check_output([exe] + [str(s) for s in shape] + [str(iters)], text=True)
    got = dict(l.split(" ", 1) for l in out.strip().splitlines() if l.startswith(("recon_fnv1a", "iters_run")))
    # the same input, generated the same way
    n = int(np.prod(shape))
    z = np.uint64(88172645463325252)
    x = np.empty(n, np.float32)
    with np.errstate(over="ignore"):
        for i in range(n):
            z ^= z << np.uint64(13); z ^= z >> np.uint64(7); z ^= z << np.uint64(17)
            x[i] = np.float32(int(z >> np.uint64(60))) + np.float32(0.001) * np.float32(i % 977)
    recon = tv.denoise4D(x.reshape(shape), np.array([1, 1, .5, .5], np.float32), iters, quiet=True)[0]
    h = 1469598103934665603
    for byte in recon.tobytes():
        h = ((h ^ byte) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert int(got["iters_run"]) == iters
    assert got["recon_fnv1a"] == f"{h:016x}"


def test_c_abi_error_paths_and_timing(tv):
    """Status codes and messages of the C ABI for bad arguments (nothing is launched), and the HIP-event timing aid."""
    import ctypes as C
    import torch
    from cytvdn_amd import _lib
    L, ctx = _lib.lib(), _lib.ctx(0)
    t = [torch.zeros((4, 3, 2, 8), dtype=torch.float32, device="cuda") for _ in range(12)]
    out = torch.zeros(4, dtype=torch.float64, device="cuda")
    sh = _lib.shape_arr((4, 3, 2, 8))

    def err():
        return L.tvdn_last_error().decode()

    assert L.tvdn_accumulator_update(ctx, 0, 4, sh, t[0].data_ptr(), t[1].data_ptr(), None, 0.0, 7, 1.0, 2, out.data_ptr(), None) == -1
    assert "ax = 7" in err()
    assert L.tvdn_accumulator_update(ctx, 0, 4, sh, t[0].data_ptr(), t[1].data_ptr(), None, 0.0, 0, 1.0, 5, out.data_ptr(), None) == -1
    assert L.tvdn_accumulator_update(ctx, 3, 4, sh, t[0].data_ptr(), t[1].data_ptr(), None, 0.0, 0, 1.0, 2, out.data_ptr(), None) == -1
    bp = (C.c_void_p * 4)(*[x.data_ptr() for x in t[2:6]])
    lm = (C.c_double * 4)(0.03, 0.03, 0.03, 0.03)
    assert L.tvdn_datacube_update(ctx, 0, 4, sh, t[0].data_ptr(), t[1].data_ptr(), bp, lm, 1, out.data_ptr(), None) == -2
    assert "mirror" in err()
    a = _lib.IterArgs(dtype=0, ndim=4, row_lo=0, row_hi=4, lo_mode=0, hi_mode=0, bc_mode=2, mode=_lib.ITER_FISTA_D)
    for i, s in enumerate((4, 3, 2, 8)):
        a.shape[i] = s
    a.orig, a.recon_in, a.recon_out = t[0].data_ptr(), t[1].data_ptr(), t[1].data_ptr()
    assert L.tvdn_iterate_fused(ctx, C.byref(a), out.data_ptr(), None) == -1 and "not in-place" in err()
    a.recon_out = t[2].data_ptr()
    assert L.tvdn_iterate_fused(ctx, C.byref(a), out.data_ptr(), None) == -1 and "d_in" in err()       # missing state
    a.row_hi = 9
    assert L.tvdn_iterate_fused(ctx, C.byref(a), out.data_ptr(), None) == -1 and "own rows" in err()
    a.row_hi, a.hi_mode, a.bc_mode = 4, _lib.EDGE_ZERO, 0
    assert L.tvdn_iterate_fused(ctx, C.byref(a), out.data_ptr(), None) == -1 and "Jia-Zhao" in err()
    a.hi_mode, a.bc_mode, a.mode = 0, 2, 9
    assert L.tvdn_iterate_fused(ctx, C.byref(a), out.data_ptr(), None) == -1 and "bad mode" in err()
    # timing aid: one valid call bracketed by events
    a.mode = _lib.ITER_PLAIN
    for q in range(4):
        a.b_in[q], a.b_out[q] = t[3 + q].data_ptr(), t[7 + q].data_ptr()
        a.clip[q], a.lambda_mu[q] = 32.0, 1 / 32
    assert L.tvdn_ctx_timing_enable(ctx, 1) == 0
    assert L.tvdn_iterate_fused(ctx, C.byref(a), out.data_ptr(), _lib.current_stream(0)) == 0
    ms, n = C.c_double(), C.c_int64()
    assert L.tvdn_ctx_timing_read(ctx, C.byref(ms), C.byref(n)) == 0 and n.value == 1 and ms.value > 0
    assert L.tvdn_ctx_timing_enable(ctx, 0) == 0


@pytest.mark.parametrize("shape,dtype,its,fista", [
    ((6, 5, 8, 12), np.float32, 9, True), ((7, 6, 16), np.float64, [4, 3], True), ((5, 3, 7, 9), np.float32, 6, False),
    ((9, 4, 4, 8), np.float64, [0, 5], True), ((9, 4, 4, 8), np.float32, [5, 0], True),
])
def test_native_loop_equals_python_loop(tv, monkeypatch, shape, dtype, its, fista):
    """tvdn_iterate_many (the whole schedule behind one library call, used whenever nothing watches the iterations)
    against the per-iteration Python loop: same launches, so the same bits, traces included."""
    from cytvdn_amd import synth
    dt = np.dtype(dtype)
    nd = len(shape)
    x = synth.cube(shape, seed=43, dtype=dt) + dt.type(0.25)
    mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
    fn = tv.denoise4D if nd == 4 else tv.denoise3D
    monkeypatch.setenv("TVDN_LOOP", "native")
    a = fn(x, mu, its, FISTA=fista, quiet=True)
    monkeypatch.setenv("TVDN_LOOP", "python")
    b = fn(x, mu, its, FISTA=fista, quiet=True)
    for u, v in zip(a, b):
        assert bits_equal(u, v)


@pytest.mark.parametrize("shape,dtype,its,fista", [
    ((6, 5, 8, 12), np.float32, 70, True), ((9, 6, 16), np.float64, [33, 3], True), ((5, 3, 7, 9), np.float32, 32, False),
    ((12, 4, 8, 16), np.float32, 5, True),
])
def test_batched_fold_of_the_sums_changes_no_bit(tv, oracle, monkeypatch, shape, dtype, its, fista):
    """Inside the library's own loops small launches park their partial sums and one launch folds up to 32 iterations'
    worth (csrc/tvdn_capi.hip sums_defer_*): the same tree per iteration, so b_norm / delta_recon agree to the last bit with
    the fold after every sweep -- more iterations than ring slots, the d -> b transition, a pipelined run (whose partial
    sweeps accumulate and are folded at once, in order with what is parked)."""
    from cytvdn_amd import synth
    dt = np.dtype(dtype)
    nd = len(shape)
    x = synth.cube(shape, seed=53, dtype=dt) + dt.type(0.25)
    mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
    fn = tv.denoise4D if nd == 4 else tv.denoise3D
    got = {}
    for defer in ("0", "1"):
        monkeypatch.setenv("TVDN_DEFER_SUMS", defer)
        for loop, pipe in (("run", "0"), ("run", "4,2,2"), ("native", "0")):
            monkeypatch.setenv("TVDN_LOOP", loop)
            monkeypatch.setenv("TVDN_PIPELINE", pipe)
            got[defer, loop, pipe] = fn(x, mu, its, FISTA=fista, quiet=True)
    for key in (("run", "0"), ("run", "4,2,2"), ("native", "0")):
        for u, v in zip(got[("0",) + key], got[("1",) + key]):
            assert bits_equal(u, v), key
    ref = oracle.denoise(x, mu, its, fista)
    assert bits_equal(got["1", "run", "0"][0], ref["recon"])
    np.testing.assert_allclose(got["1", "run", "0"][1].astype(np.float64), ref["b_norm64"].astype(dt).astype(np.float64), rtol=1e-5)


@pytest.mark.parametrize("shape,dtype", [((6, 5, 8, 12), np.float32), ((7, 6, 16), np.float64)])
def test_zero_iterations_return_the_input(tv, shape, dtype):
    """iterations=0: upstream's loops do not run (cyTVDN.py:148, :196) -- recon is a copy of the input, the traces are empty;
    the input array is left alone."""
    from cytvdn_amd import synth
    dt = np.dtype(dtype)
    nd = len(shape)
    x = synth.cube(shape, seed=59, dtype=dt) + dt.type(0.25)
    x0 = x.copy()
    mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
    fn = tv.denoise4D if nd == 4 else tv.denoise3D
    for its, fista in ((0, True), (0, False), ([0, 0], True)):
        recon, bn, dl = fn(x, mu, its, FISTA=fista, quiet=True)
        assert bits_equal(recon, x0) and recon is not x and bits_equal(x, x0)
        assert bn.shape == (0,) and dl.shape == (0,) and bn.dtype == dt and dl.dtype == dt
