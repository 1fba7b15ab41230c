"""Randomised parity (hypothesis): arbitrary small shapes, dtypes, schedules, boundary conditions and
parameter scales through the fused HIP sweep and the one-pass kernels, against the CPU oracle, bit for bit."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from golden_util import bits_equal

pytestmark = pytest.mark.gpu

dims4 = st.tuples(st.integers(1, 9), st.integers(1, 6), st.integers(1, 7), st.sampled_from([1, 2, 3, 4, 5, 8, 12, 16, 20]))
dims3 = st.tuples(st.integers(1, 12), st.integers(1, 9), st.sampled_from([1, 2, 4, 6, 7, 8, 16, 24, 36]))


@settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
@given(shape=st.one_of(dims4, dims3), f64=st.booleans(), bc=st.sampled_from([0, 2]),
       sched=st.sampled_from(["fista", "plain", "hybrid"]), seed=st.integers(0, 2 ** 31 - 1),
       scale=st.sampled_from([1e-3, 1.0, 37.0]), lam_div=st.sampled_from([None, 40.0, 333.0]))
def test_fused_loop_random(oracle, shape, f64, bc, sched, seed, scale, lam_div):
    import cytvdn_amd as tv
    dt = np.dtype(np.float64 if f64 else np.float32)
    nd = len(shape)
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(shape) * scale + rng.poisson(3.0, shape)).astype(dt)
    mu = (np.array([1.0, 0.7, 0.5, 1.3][:nd]) * scale).astype(dt)
    lam = None if lam_div is None else (mu / dt.type(lam_div)).astype(dt)
    its = {"fista": 5, "plain": 4, "hybrid": [3, 2]}[sched]
    fista = sched != "plain"
    fn = tv.denoise4D if nd == 4 else tv.denoise3D
    got = fn(x, mu, its, FISTA=fista, BC_mode=bc, lam=lam, quiet=True)
    ref = oracle.denoise(x, mu, its, fista, BC_mode=bc, lam=lam)
    assert bits_equal(got[0], ref["recon"])
    tol = 1e-6 if dt == np.float32 else 1e-12
    np.testing.assert_allclose(got[1].astype(np.float64), ref["b_norm64"].astype(dt).astype(np.float64), rtol=tol)


@settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
@given(shape=st.one_of(dims4, dims3), f64=st.booleans(), bc=st.sampled_from([0, 1, 2]), fista=st.booleans(),
       seed=st.integers(0, 2 ** 31 - 1), data=st.data())
def test_one_pass_kernels_random(oracle, shape, f64, bc, fista, seed, data):
    import cytvdn_amd as tv
    dt = np.dtype(np.float64 if f64 else np.float32)
    nd = len(shape)
    ax = data.draw(st.integers(0, nd - 1))
    if bc == 1 and shape[ax] < 2:
        bc = 2
    rng = np.random.default_rng(seed)
    a = (rng.standard_normal(shape) * 3).astype(dt)
    b = (rng.standard_normal(shape) * 0.7).astype(dt)
    d = (rng.standard_normal(shape) * 0.7).astype(dt)
    if data.draw(st.booleans()):
        a.flat[int(rng.integers(a.size))] = np.inf       # non-finite input propagates as upstream
    b2, d2 = b.copy(), d.copy()
    clip, tk = dt.type(0.8), dt.type(0.41)
    suffix = f"{nd}D_FISTA" if fista else f"{nd}D"
    if fista:
        ret = getattr(tv, "accumulator_update_" + suffix)(a, b, d, tk, ax, clip, BC_mode=bc)
        _, n64 = oracle.acc_update(a, b2, d2, tk, ax, clip, bc)
        assert bits_equal(d, d2)
    else:
        ret = getattr(tv, "accumulator_update_" + suffix)(a, b, ax, clip, BC_mode=bc)
        _, n64 = oracle.acc_update(a, b2, None, 0.0, ax, clip, bc)
    assert bits_equal(b, b2)
    assert (np.isnan(ret) and np.isnan(n64)) or ret == pytest.approx(n64, rel=1e-12)
    if bc != 1:
        orig = (rng.standard_normal(shape) * 3).astype(dt)
        recon = (rng.standard_normal(shape) * 3).astype(dt)
        bs = [(rng.standard_normal(shape) * 0.7).astype(dt) for _ in range(nd)]
        lm = (np.array([1 / 32, 1 / 40, 1 / 64, 1 / 33][:nd])).astype(dt)
        r2 = recon.copy()
        getattr(tv, f"datacube_update_{nd}D")(orig, recon, *bs, lm, BC_mode=bc)
        oracle.recon_update(orig, r2, bs, lm, bc)
        assert bits_equal(recon, r2)


@settings(max_examples=40, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
@given(rows=st.integers(2, 14), plane=st.one_of(st.tuples(st.integers(1, 4), st.integers(1, 5), st.sampled_from([1, 3, 4, 8, 12])),
                                                st.tuples(st.integers(1, 6), st.sampled_from([2, 4, 7, 8, 16]))),
       f64=st.booleans(), bc=st.sampled_from([0, 2]), sched=st.sampled_from(["fista", "plain", "hybrid"]),
       seed=st.integers(0, 2 ** 31 - 1), world=st.integers(2, 5), split=st.booleans(), bad_row0=st.booleans(),
       uneven=st.booleans())
def test_slab_layouts_random(oracle, rows, plane, f64, bc, sched, seed, world, split, bad_row0, uneven):
    """Random cubes cut into random slab layouts on one GPU (balanced or explicit uneven bounds, whole sweeps or
    edge-blocks-first, chain or ring, with the exact Jia-Zhao wrap row when the first row is not finite): the gathered
    result must equal the oracle's single-cube result bit for bit, and the summed scalars its f64 yardsticks."""
    from cytvdn_amd.engine import HipBackend, LocalSlabs, SlabLayout
    world = min(world, rows)
    shape = (rows,) + tuple(plane)
    dt = np.dtype(np.float64 if f64 else np.float32)
    nd = len(shape)
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(shape) * 2 + rng.poisson(3.0, shape)).astype(dt)
    wrap_row = False
    if bad_row0 and bc == 2:
        x[(0,) + tuple(int(rng.integers(s)) for s in shape[1:])] = np.inf
        wrap_row = True
    mu = np.array([1.0, 0.7, 0.5, 1.3][:nd], dt)
    lam = mu / dt.type(32.0 if nd == 4 else 16.0)
    n_f, n_p = {"fista": (5, 0), "plain": (0, 4), "hybrid": (3, 2)}[sched]
    bounds = None
    if uneven and rows > world:
        cuts = sorted(rng.choice(np.arange(1, rows), size=world - 1, replace=False).tolist())
        bounds = tuple([0] + cuts + [rows])
    bes = []
    for r in range(world):
        lay = SlabLayout(shape, r, world, bc, bounds=bounds, wrap_row=wrap_row)
        be = HipBackend(lay, dt, n_f > 0, device=0, max_iters=n_f + n_p)
        be.set_params(1.0 / lam, (lam / mu).astype(dt))
        be.set_input(x[lay.local_rows_global()])
        bes.append(be)
    grp = LocalSlabs(bes, split_sweeps=split)
    grp.run(n_f, n_p)
    its = [n_f, n_p] if (n_f and n_p) else (n_f or n_p)
    ref = oracle.denoise(x, mu, its, n_f > 0, BC_mode=bc)
    assert bits_equal(grp.gather_recon().cpu().numpy(), ref["recon"])
    if not wrap_row:
        s = grp.global_sums().cpu().numpy()
        np.testing.assert_allclose(s[:, 0], ref["b_norm64"], rtol=1e-12)
        np.testing.assert_allclose(s[:, 1], ref["delta64"], rtol=1e-12)


@settings(max_examples=25, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
@given(rows=st.integers(2, 12), plane=st.tuples(st.integers(1, 4), st.integers(1, 5), st.sampled_from([1, 4, 8, 12])),
       f64=st.booleans(), bc=st.sampled_from([0, 2]), n_f=st.integers(0, 5), n_p=st.integers(0, 4),
       seed=st.integers(0, 2 ** 31 - 1), slabs=st.integers(1, 5), bad_row0=st.booleans())
def test_tvdn_run_device_list_random(oracle, rows, plane, f64, bc, n_f, n_p, seed, slabs, bad_row0):
    """The C whole-loop entry with a random number of slabs on device 0 against the oracle."""
    import ctypes as C
    from cytvdn_amd import _lib
    if n_f + n_p == 0:
        n_f = 1
    slabs = min(slabs, rows)
    shape = (rows,) + tuple(plane)
    dt = np.dtype(np.float64 if f64 else np.float32)
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(shape) * 2 + rng.poisson(3.0, shape)).astype(dt)
    if bad_row0:
        x[(0,) + tuple(int(rng.integers(s)) for s in shape[1:])] = np.nan
    mu = np.array([1.0, 0.7, 0.5, 1.3], dt)
    lam = mu / dt.type(32.0)
    a = _lib.RunArgs(dtype=_lib.dtype_code(dt), ndim=4, bc_mode=bc, device=0, n_fista=n_f, n_plain=n_p, n_devices=slabs)
    for i, s in enumerate(shape):
        a.shape[i] = s
    for q in range(4):
        a.clip[q] = float((1.0 / lam)[q])
        a.lambda_mu[q] = float((lam / mu).astype(dt)[q])
    recon, sums = np.empty_like(x), np.zeros((n_f + n_p, 3))
    a.data, a.recon_out, a.sums_out = x.ctypes.data, recon.ctypes.data, sums.ctypes.data
    _lib.check(_lib.lib().tvdn_run(C.byref(a)))
    its = [n_f, n_p] if (n_f and n_p) else (n_f or n_p)
    ref = oracle.denoise(x, mu, its, n_f > 0, BC_mode=bc)
    assert bits_equal(recon, ref["recon"])


@settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
@given(shape=st.one_of(st.tuples(st.integers(2, 40), st.integers(1, 4), st.integers(1, 5), st.sampled_from([1, 3, 4, 8, 12])),
                       st.tuples(st.integers(2, 40), st.integers(1, 6), st.sampled_from([2, 4, 7, 8, 16]))),
       f64=st.booleans(), sched=st.sampled_from(["fista", "plain", "hybrid"]), seed=st.integers(0, 2 ** 31 - 1),
       rows=st.integers(1, 48), k0=st.integers(0, 9), k1=st.integers(0, 9), n1=st.integers(1, 9), n2=st.integers(1, 6))
def test_pipelined_transfers_random(oracle, shape, f64, sched, seed, rows, k0, k1, n1, n2):
    """tvdn_run's pipelined transfers with arbitrary chunk heights and wavefront depths (more levels than rows, chunks
    taller than the cube, depths beyond the iteration count, a d -> b transition anywhere) against the oracle."""
    import os
    import cytvdn_amd as tv
    dt = np.dtype(np.float64 if f64 else np.float32)
    nd = len(shape)
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(shape) + rng.poisson(3.0, shape)).astype(dt)
    mu = np.array([1.0, 0.7, 0.5, 1.3][:nd], dt)
    its = {"fista": n1, "plain": n1, "hybrid": [n1, n2]}[sched]
    fista = sched != "plain"
    fn = tv.denoise4D if nd == 4 else tv.denoise3D
    old = os.environ.get("TVDN_PIPELINE")
    os.environ["TVDN_PIPELINE"] = f"{rows},{k0},{k1}"
    try:
        got = fn(x, mu, its, FISTA=fista, quiet=True)
    finally:
        if old is None:
            del os.environ["TVDN_PIPELINE"]
        else:
            os.environ["TVDN_PIPELINE"] = old
    ref = oracle.denoise(x, mu, its, fista)
    assert bits_equal(got[0], ref["recon"])
    tol = 1e-6 if dt == np.float32 else 1e-12
    np.testing.assert_allclose(got[1].astype(np.float64), ref["b_norm64"].astype(dt).astype(np.float64), rtol=tol)


@settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
@given(rows=st.integers(2, 30), plane=st.one_of(st.tuples(st.integers(1, 4), st.integers(1, 5), st.sampled_from([1, 3, 4, 8, 12])),
                                                st.tuples(st.integers(1, 6), st.sampled_from([2, 4, 7, 8, 16]))),
       f64=st.booleans(), bc=st.sampled_from([0, 2]), n_f=st.integers(0, 7), n_p=st.integers(0, 5), seed=st.integers(0, 2 ** 31 - 1),
       chunk=st.integers(1, 12), k=st.integers(1, 9), bad_row0=st.booleans())
def test_streamed_run_random(oracle, rows, plane, f64, bc, n_f, n_p, seed, chunk, k, bad_row0):
    """The C++ streamed loop (csrc/tvdn_stream.hip) with arbitrary chunk heights and depths, both boundary conditions, any
    schedule, a non-finite first row now and then: recon against the oracle bit for bit, the sums against its f64 yardsticks."""
    from test_gpu_run_streamed import _check_traces, _oracle_bc, _run
    if n_f + n_p == 0:
        n_f = 1
    shape = (rows,) + tuple(plane)
    dt = np.dtype(np.float64 if f64 else np.float32)
    nd = len(shape)
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(shape) * 2 + rng.poisson(3.0, shape)).astype(dt)
    if bad_row0 and bc == 2:
        x[(0,) + tuple(int(rng.integers(s)) for s in shape[1:])] = np.inf
    mu = np.array([1.0, 0.7, 0.5, 1.3][:nd], dt)
    ref = _oracle_bc(oracle, x, mu, n_f, n_p, bc)
    recon, sums, _, ran = _run(x, mu, n_f, n_p, stream=(chunk, k), bc=bc)
    assert ran == n_f + n_p
    assert bits_equal(recon, ref["recon"])
    if np.isfinite(ref["b_norm64"]).all() and np.isfinite(ref["delta64"]).all():
        _check_traces(sums, ref, n_f + n_p)


@settings(max_examples=50, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
@given(rows=st.integers(3, 30), plane=st.one_of(st.tuples(st.integers(1, 4), st.integers(2, 5), st.sampled_from([4, 8, 12, 32])),
                                                st.tuples(st.integers(2, 6), st.sampled_from([4, 7, 8, 16, 64]))),
       f64=st.booleans(), n_f=st.integers(0, 7), n_p=st.integers(0, 5), seed=st.integers(0, 2 ** 31 - 1),
       chunk=st.integers(1, 8), k=st.integers(1, 9), res_frac=st.floats(0.0, 1.0), in_place=st.booleans(), bad_row0=st.booleans(),
       stop=st.booleans(), chain=st.booleans(), bc=st.sampled_from([2, 2, 0]), with_mse=st.booleans())
def test_streamed_hybrid_random(oracle, rows, plane, f64, n_f, n_p, seed, chunk, k, res_frac, in_place, bad_row0, stop, chain, bc, with_mse):
    """The resident + streamed hybrid with any share of the rows resident, the caller's arrays page-locked where they are or
    staged through packed copies, chained or drained passes, with and without a stopping rule, a non-finite first row now and
    then: the oracle's bits and traces."""
    import os
    from test_gpu_run_streamed import _check_traces, _oracle_bc, _run
    if n_f + n_p == 0:
        n_f = 2
    shape = (rows,) + tuple(plane)
    dt = np.dtype(np.float64 if f64 else np.float32)
    nd = len(shape)
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(shape) * 2 + rng.poisson(3.0, shape)).astype(dt)
    clean = rng.poisson(3.0, shape).astype(dt) if with_mse else None
    if bad_row0 and bc == 2:
        x[(0,) + tuple(int(rng.integers(s)) for s in shape[1:])] = np.nan
    mu = np.array([1.0, 0.7, 0.5, 1.3][:nd], dt)
    resident = int(round(res_frac * rows))
    stop_v = 0.02 if stop else None
    env = {"TVDN_PIN_IN_PLACE_MIN": "4K" if in_place else "1G", "TVDN_STREAM_CHAIN": "1" if chain else "0"}
    old = {k_: os.environ.get(k_) for k_ in env}
    os.environ.update(env)
    try:
        recon, sums, mse, ran = _run(x, mu, n_f, n_p, stream=(chunk, k), resident=resident, stop=stop_v, bc=bc, ref=clean)
    finally:
        for k_, v in old.items():
            if v is None:
                del os.environ[k_]
            else:
                os.environ[k_] = v
    ref = _oracle_bc(oracle, x, mu, n_f, n_p, bc, stopping_relative_change=stop_v, reference_data=clean)
    assert bits_equal(recon, ref["recon"])
    if stop_v is None:
        assert ran == n_f + n_p
        if np.isfinite(ref["b_norm64"]).all() and np.isfinite(ref["delta64"]).all():
            _check_traces(sums, ref, n_f + n_p)
            if with_mse:
                np.testing.assert_allclose(mse[:ran + 1], ref["MSE64"][:ran + 1], rtol=1e-6 if dt == np.float32 else 1e-11)
    else:
        assert ran == ref["iters_done"]


@settings(max_examples=40, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
@given(rows=st.integers(4, 28), plane=st.tuples(st.integers(1, 4), st.integers(2, 5), st.sampled_from([4, 8, 12])),
       f64=st.booleans(), bc=st.sampled_from([0, 2]), n_f=st.integers(0, 6), n_p=st.integers(0, 4), seed=st.integers(0, 2 ** 31 - 1),
       chunk=st.integers(1, 6), k=st.integers(1, 7), slabs=st.integers(2, 4), bad_row0=st.booleans(), stop=st.booleans())
def test_streamed_device_list_random(oracle, rows, plane, f64, bc, n_f, n_p, seed, chunk, k, slabs, bad_row0, stop):
    """tvdn_run with a device list AND stream_rows / stream_k (BASELINE configs[4] in structure, one process): slabs streamed
    from host arrays they share, any cut, both boundary conditions, stopping rule, non-finite first row: the oracle's bits."""
    from test_gpu_run_streamed import _oracle_bc, _run
    if n_f + n_p == 0:
        n_p = 2
    slabs = min(slabs, rows)
    shape = (rows,) + tuple(plane)
    dt = np.dtype(np.float64 if f64 else np.float32)
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(shape) * 2 + rng.poisson(3.0, shape)).astype(dt)
    if bad_row0 and bc == 2:
        x[(0,) + tuple(int(rng.integers(s)) for s in shape[1:])] = np.inf
    mu = np.array([1.0, 0.7, 0.5, 1.3], dt)
    stop_v = 0.02 if stop else None
    ref = _oracle_bc(oracle, x, mu, n_f, n_p, bc, stopping_relative_change=stop_v)
    recon, sums, _, ran = _run(x, mu, n_f, n_p, stream=(chunk, k), bc=bc, devices=[0] * slabs, stop=stop_v)
    assert bits_equal(recon, ref["recon"])
    assert ran == (ref["iters_done"] if stop_v is not None else n_f + n_p)
