"""tvdn_run's out-of-core branch (csrc/tvdn_stream.hip): a cube streamed through the GPU from pinned host memory with the
wavefront schedule gives the bits of the resident run -- recon and the b_norm / delta_recon traces -- for every schedule
(FISTA, hybrid, unaccelerated, stopping rule), dtype, rank, chunk height and depth, with non-finite first rows, and the
MSE trace within summation-order tolerance.  Every case is ALSO compared with the CPU oracle directly (recon bit for bit,
traces against its f64 yardsticks), not only with the resident HIP run: the ring instantiation of the sweep and the
streamed loop are the least-travelled code in the tree (reference loop: cyTVDN/cyTVDN.py:148-242)."""
import ctypes as C

import numpy as np
import pytest

from golden_util import bits_equal

pytestmark = pytest.mark.gpu


def _oracle_bc(oracle, x, mu, n_f, n_p, bc, **kw):
    if n_f and n_p:
        return oracle.denoise(x, mu, [n_f, n_p], True, BC_mode=bc, **kw)
    return oracle.denoise(x, mu, n_f or n_p, bool(n_f), BC_mode=bc, **kw)


def _oracle(oracle, x, mu, n_f, n_p, **kw):
    """The CPU oracle on the same schedule (lam = mu/32 for 4-D, mu/16 for 3-D, as `_run` sets it)."""
    if n_f and n_p:
        return oracle.denoise(x, mu, [n_f, n_p], True, **kw)
    return oracle.denoise(x, mu, n_f or n_p, bool(n_f), **kw)


def _check_traces(sums, ref, n):
    np.testing.assert_allclose(sums[:n, 0], ref["b_norm64"][:n], rtol=1e-9)
    np.testing.assert_allclose(sums[:n, 1], ref["delta64"][:n], rtol=1e-9)
    np.testing.assert_allclose(sums[:n, 2], ref["rnorm64"][:n], rtol=1e-9)


def _run(x, mu, n_f, n_p, stop=None, ref=None, stream=None, device=0, bc=2, resident=0, stats=None, devices=None):
    from cytvdn_amd import _lib
    dt = x.dtype
    nd = x.ndim
    lam = mu / dt.type(32.0 if nd == 4 else 16.0)
    n = n_f + n_p
    a = _lib.RunArgs(dtype=_lib.dtype_code(dt), ndim=nd, bc_mode=bc, device=device, n_fista=n_f, n_plain=n_p,
                     use_stop=int(stop is not None), stop=float(stop or 0.0))
    if stream:
        a.stream_rows, a.stream_k = stream
        a.stream_resident = resident
    if stats is not None:
        a.stats = C.addressof(stats)
    if devices is not None:
        a.n_devices = len(devices)
        for i, d in enumerate(devices):
            a.devices[i] = d
    for i, s in enumerate(x.shape):
        a.shape[i] = s
    for q in range(nd):
        a.clip[q] = float((1.0 / lam)[q])
        a.lambda_mu[q] = float((lam / mu).astype(dt)[q])
    recon = np.empty_like(x)
    sums = np.zeros((max(n, 1), 3))
    mse = np.zeros(n + 1)
    ran = C.c_int32(0)
    a.data, a.recon_out, a.sums_out = x.ctypes.data, recon.ctypes.data, sums.ctypes.data
    if ref is not None:
        a.reference, a.mse_out = ref.ctypes.data, mse.ctypes.data
    a.iters_run = C.addressof(ran)
    _lib.check(_lib.lib().tvdn_run(C.byref(a)))
    return recon, sums[:n], mse, ran.value


@pytest.mark.parametrize("shape,dtype,n_f,n_p,rows,k", [
    ((23, 3, 4, 8), np.float32, 9, 0, 5, 3),
    ((23, 3, 4, 8), np.float32, 9, 0, 2, 9),          # deeper than a chunk
    ((23, 3, 4, 8), np.float32, 5, 4, 4, 4),          # hybrid: the d -> b transition between passes and inside one
    ((23, 3, 4, 8), np.float32, 5, 4, 3, 9),
    ((17, 6, 16), np.float64, 0, 7, 3, 5),            # unaccelerated, 3-D, f64
    ((17, 6, 16), np.float64, 7, 0, 17, 2),
    ((9, 2, 5, 7), np.float32, 6, 0, 2, 8),           # scalar packs; more levels than rows
    ((5, 3, 4, 8), np.float32, 12, 0, 16, 5),         # chunk taller than the cube
    ((19, 3, 4, 8), np.float32, 6, 3, 1, 7),          # one-row chunks
    ((40, 4, 8, 16), np.float32, 11, 0, 8, 128),      # k beyond the iteration count: one pass
    ((40, 4, 8, 16), np.float32, 11, 0, 4, 3),        # four passes
    ((40, 4, 8, 16), np.float32, 7, 6, 3, 4),         # the d -> b transition inside the second pass
    ((37, 6, 16), np.float64, 0, 9, 2, 2),            # odd row count: the last chunk of a pass is short
    ((33, 3, 4, 8), np.float32, 10, 0, 1, 5),         # one-row chunks, two passes
])
def test_streamed_run_equals_resident_run(oracle, shape, dtype, n_f, n_p, rows, k):
    from cytvdn_amd import synth
    dt = np.dtype(dtype)
    nd = len(shape)
    x = synth.cube(shape, seed=61, dtype=dt) + dt.type(0.25)
    mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
    want = _run(x, mu, n_f, n_p)
    x_before = x.copy()
    got = _run(x, mu, n_f, n_p, stream=(rows, k))
    assert bits_equal(x, x_before)                       # the input is page-locked in place, never written
    assert bits_equal(got[0], want[0])
    assert got[3] == want[3] == n_f + n_p
    np.testing.assert_allclose(got[1], want[1], rtol=1e-12)   # f64 sums: per-launch partials added in another order
    ref = _oracle(oracle, x, mu, n_f, n_p)                    # ... and the oracle itself, not only the resident HIP run
    assert bits_equal(got[0], ref["recon"])
    _check_traces(got[1], ref, n_f + n_p)


@pytest.mark.parametrize("shape,dtype,n_f,n_p,rows,k,resident", [
    ((23, 3, 4, 8), np.float32, 9, 0, 5, 3, 7),           # the boundary inside the second chunk; three passes
    ((23, 3, 4, 8), np.float32, 9, 0, 5, 3, 5),           # ... exactly at a chunk boundary
    ((23, 3, 4, 8), np.float32, 9, 0, 5, 3, 1),           # one resident row
    ((23, 3, 4, 8), np.float32, 9, 0, 5, 3, 22),          # one streamed row
    ((23, 3, 4, 8), np.float32, 9, 0, 5, 3, 23),          # every row resident: nothing crosses PCIe between the passes
    ((23, 3, 4, 8), np.float32, 9, 0, 5, 3, -1),          # as many as fit (all of this cube)
    ((23, 3, 4, 8), np.float32, 5, 4, 3, 9, 11),          # hybrid, d -> b inside the one pass
    ((40, 4, 8, 16), np.float32, 7, 6, 3, 4, 17),         # hybrid over four passes: the kept state changes form in the store
    ((17, 6, 16), np.float64, 0, 7, 3, 5, 8),             # unaccelerated, 3-D, f64
    ((17, 6, 16), np.float64, 7, 0, 17, 2, 9),            # one chunk per pass
    ((9, 2, 5, 7), np.float32, 6, 0, 2, 8, 4),            # scalar packs; more levels than rows
    ((19, 3, 4, 8), np.float32, 6, 3, 1, 7, 10),          # one-row chunks
    ((33, 3, 4, 8), np.float32, 10, 0, 1, 5, 32),
])
def test_streamed_run_with_resident_rows(oracle, shape, dtype, n_f, n_p, rows, k, resident):
    """The resident + streamed hybrid (tvdn_run_args.stream_resident): the low rows keep their state in HBM between the
    passes and enter / leave the rings by device copies, the others cross PCIe; wherever the boundary falls the result is
    the resident run's and the oracle's, bit for bit, traces included."""
    from cytvdn_amd import _lib, synth
    dt = np.dtype(dtype)
    nd = len(shape)
    x = synth.cube(shape, seed=23, dtype=dt) + dt.type(0.25)
    mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
    want = _run(x, mu, n_f, n_p)
    st = _lib.RunStats()
    got = _run(x, mu, n_f, n_p, stream=(rows, k), resident=resident, stats=st)
    n = n_f + n_p
    assert got[3] == want[3] == n
    assert bits_equal(got[0], want[0])
    np.testing.assert_allclose(got[1], want[1], rtol=1e-12)
    ref = _oracle(oracle, x, mu, n_f, n_p)
    assert bits_equal(got[0], ref["recon"])
    _check_traces(got[1], ref, n)
    want_res = shape[0] if resident < 0 else min(resident, shape[0])
    assert st.engine == 1 and st.resident_rows == want_res
    # what crossed PCIe: the cube once each way plus, per pass, the streamed rows only (none of them in the first pass's state upload)
    hr = shape[0] - want_res
    per_row = x.nbytes // shape[0]
    n_state = 2 if n_f else 1
    assert st.h2d_bytes <= per_row * (want_res + st.n_passes * hr * (2 + nd * n_state))
    assert st.d2h_bytes <= per_row * (want_res + st.n_passes * hr * (1 + nd * n_state))
    if hr == 0:
        assert st.h2d_bytes == x.nbytes and st.d2h_bytes == x.nbytes


@pytest.mark.parametrize("shape,dtype,n_f,n_p,rows,k,resident,kind", [
    ((40, 4, 8, 16), np.float32, 12, 0, 4, 4, 40, 3),      # every row kept, three passes of four levels: the lean layout, no copy at all
    ((40, 4, 8, 16), np.float32, 12, 0, 2, 3, 40, 3),      # ... the shallowest depth at which the shared state is written in place
    ((40, 4, 8, 16), np.float32, 12, 0, 16, 3, 40, 3),     # ... tall chunks (the lean layout's rings leave room for them)
    ((40, 4, 8, 16), np.float32, 11, 0, 4, 4, 40, 3),      # ... passes of 4 + 4 + 3 levels
    ((40, 4, 8, 16), np.float32, 10, 0, 4, 4, 40, 3),      # ... of 4 + 3 + 3
    ((40, 4, 8, 16), np.float32, 8, 0, 4, 3, 40, 3),       # asked as 3 + 3 + 2 levels: a pass of two needs the level-0 rings -- the lean run makes it 4 + 4
    ((40, 4, 8, 16), np.float32, 12, 0, 4, 2, 40, 1),      # two levels: level 0's inputs and level 1's outputs only
    ((40, 4, 8, 16), np.float32, 7, 6, 3, 4, 40, 3),       # hybrid schedule: the kept state changes form inside a pass and between passes
    ((40, 4, 8, 16), np.float32, 0, 9, 4, 3, 40, 3),       # unaccelerated: one state array per axis
    ((40, 4, 8, 16), np.float32, 12, 0, 4, 4, 33, 1),      # seven streamed rows among the kept ones: in place between them
    ((40, 4, 8, 16), np.float32, 12, 0, 1, 4, 30, 1),      # one-row chunks
    ((40, 4, 8, 16), np.float32, 7, 6, 2, 5, 36, 1),
    ((41, 6, 16), np.float64, 9, 0, 3, 3, 41, 3),          # 3-D, f64, a ragged last chunk
    ((41, 6, 16), np.float64, 5, 5, 5, 5, 37, 1),
    ((23, 2, 5, 7), np.float32, 8, 0, 2, 4, 23, 3),        # scalar packs
    ((40, 4, 8, 16), np.float32, 12, 0, 4, 1, 40, 0),      # one level per pass: nothing to gain, copies as before
])
def test_kept_rows_are_swept_in_place(oracle, monkeypatch, shape, dtype, n_f, n_p, rows, k, resident, kind):
    """Rows kept in HBM between the passes are read by level 0 where they are kept and written there by the last level (ring sizes of
    their own per array, tvdn_iter_args ABI 8; csrc/tvdn_stream_chain.hip) wherever the rows next to them are kept too -- every
    row when all are (`kept_in_place` 2; 3 on the lean layout a run takes when every pass is at least three levels deep: rings for
    the levels in between only, the first pass in place too, recon from the data term and the state from a plane of zeros).  The bits are those of the copying schedule (TVDN_STREAM_INPLACE=0), of the resident run
    and of the oracle, traces included; a non-finite first row (exact Jia-Zhao wrap: row 0 of every level kept aside) as well."""
    from cytvdn_amd import _lib, synth
    dt = np.dtype(dtype)
    nd = len(shape)
    mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
    n = n_f + n_p
    for bad in (False, True):
        x = synth.cube(shape, seed=29, dtype=dt) + dt.type(0.25)
        if bad:
            x[0, ..., 2] = np.inf
        st = _lib.RunStats()
        monkeypatch.delenv("TVDN_STREAM_INPLACE", raising=False)
        got = _run(x, mu, n_f, n_p, stream=(rows, k), resident=resident, stats=st)
        assert st.engine == 1 and st.resident_rows == resident and st.kept_in_place == kind
        if kind == 3:       # the same run on the general layout: every pass but the first in place
            st2 = _lib.RunStats()
            monkeypatch.setenv("TVDN_STREAM_LEAN", "0")
            fat = _run(x, mu, n_f, n_p, stream=(rows, k), resident=resident, stats=st2)
            monkeypatch.delenv("TVDN_STREAM_LEAN")
            assert st2.kept_in_place == 2 and bits_equal(got[0], fat[0])
            np.testing.assert_allclose(got[1], fat[1], rtol=1e-12)     # (the lean run may cut the iterations into other passes: f64 sums to rounding)
        st0 = _lib.RunStats()
        monkeypatch.setenv("TVDN_STREAM_INPLACE", "0")
        copied = _run(x, mu, n_f, n_p, stream=(rows, k), resident=resident, stats=st0)
        assert st0.kept_in_place == 0
        assert bits_equal(got[0], copied[0])
        np.testing.assert_allclose(got[1], copied[1], rtol=1e-12)
        assert (st.h2d_bytes, st.d2h_bytes) == (st0.h2d_bytes, st0.d2h_bytes)
        ref = _oracle(oracle, x, mu, n_f, n_p)
        assert bits_equal(got[0], ref["recon"])
        if not bad:
            _check_traces(got[1], ref, n)


@pytest.mark.parametrize("shape,dtype,n_f,n_p,rows,k,resident", [
    ((36, 8, 16, 32), np.float32, 9, 0, 4, 3, 20),        # three passes: the last one sends the resident rows' results home itself
    ((36, 8, 16, 32), np.float32, 5, 4, 3, 9, 30),        # ONE pass (first = last), hybrid schedule
    ((36, 8, 16, 32), np.float32, 8, 0, 2, 4, 36),        # every row resident: nothing of the caller's is page-locked, results go home at the end
    ((40, 32, 64), np.float64, 0, 7, 5, 3, 13),           # unaccelerated, 3-D, f64: 7 = 3 + 2 + 2
])
@pytest.mark.parametrize("home_after", [False, True])
def test_resident_rows_with_the_callers_arrays_page_locked_in_place(oracle, monkeypatch, shape, dtype, n_f, n_p, rows, k, resident, home_after):
    """Arrays of 256 MiB and more are page-locked where they are (no packed copies): the form every large run takes, reached
    here on small cubes by lowering the threshold.  In that form the run's LAST pass sends the results of the resident rows
    across PCIe as they reach the last level (tvdn_run_stats.results_under_last_pass) instead of in one piece after it;
    TVDN_STREAM_HOME_AFTER=1 keeps the old order.  Same bits either way: the oracle's."""
    from cytvdn_amd import _lib, synth
    monkeypatch.setenv("TVDN_PIN_IN_PLACE_MIN", "64K")
    if home_after:
        monkeypatch.setenv("TVDN_STREAM_HOME_AFTER", "1")
    dt = np.dtype(dtype)
    nd = len(shape)
    x = synth.cube(shape, seed=29, dtype=dt) + dt.type(0.25)
    assert x.nbytes >= 128 * 1024                              # its own pages (the allocator maps blocks of this size)
    mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
    st = _lib.RunStats()
    x_before = x.copy()
    got = _run(x, mu, n_f, n_p, stream=(rows, k), resident=resident, stats=st)
    n = n_f + n_p
    ref = _oracle(oracle, x, mu, n_f, n_p)
    assert bits_equal(x, x_before) and bits_equal(got[0], ref["recon"])
    _check_traces(got[1], ref, n)
    assert st.resident_rows == resident and st.results_under_last_pass == (0 if home_after or resident == shape[0] else 1)
    per_row = x.nbytes // shape[0]
    hr = shape[0] - resident
    n_state = 2 if n_f else 1
    assert st.d2h_bytes <= per_row * (resident + st.n_passes * hr * (1 + nd * n_state))


@pytest.mark.parametrize("chain", ["1", "0"])
@pytest.mark.parametrize("shape,dtype,n_f,n_p,rows,k,resident", [
    ((23, 3, 4, 8), np.float32, 12, 0, 5, 3, 0),          # four passes, the seam between two passes inside a chunk (23 = 4 x 5 + 3)
    ((23, 3, 4, 8), np.float32, 12, 0, 5, 3, 9),          # ... with rows resident in HBM
    ((24, 3, 4, 8), np.float32, 7, 6, 4, 4, 0),           # hybrid: the d -> b transition inside the second of four passes
    ((24, 3, 4, 8), np.float32, 7, 6, 4, 4, 11),
    ((30, 6, 16), np.float64, 0, 10, 2, 3, 0),            # unaccelerated 3-D f64, passes of 3 + 3 + 2 + 2 levels
    ((30, 6, 16), np.float64, 9, 0, 3, 2, 14),            # five passes (2 + 2 + 2 + 2 + 1)
    ((16, 2, 5, 7), np.float32, 8, 0, 1, 4, 5),           # one-row chunks, scalar packs
])
def test_chained_passes(oracle, monkeypatch, shape, dtype, n_f, n_p, rows, k, resident, chain):
    """Several passes of a Jia-Zhao run are CHAINED (pass p + 1 starts uploading while the upper levels of pass p are still
    climbing, one running row index over all passes, launches cut at the seams) when k <= N0 - 3 rows; TVDN_STREAM_CHAIN=0
    keeps them apart.  Either way: the resident run's and the oracle's bits, traces included; an MSE trace and a non-finite
    first row (two sets of top-face planes, one per pass in flight) too."""
    from cytvdn_amd import _lib, synth
    monkeypatch.setenv("TVDN_STREAM_CHAIN", chain)
    dt = np.dtype(dtype)
    nd = len(shape)
    x = synth.cube(shape, seed=29, dtype=dt) + dt.type(0.25)
    mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
    n = n_f + n_p
    assert k <= shape[0] - 3 * rows                       # these cases do chain
    st = _lib.RunStats()
    got = _run(x, mu, n_f, n_p, stream=(rows, k), resident=resident, stats=st)
    ref = _oracle(oracle, x, mu, n_f, n_p)
    assert bits_equal(got[0], ref["recon"]) and got[3] == n and st.n_passes == -(-n // k)
    _check_traces(got[1], ref, n)
    if resident == 0:
        clean = synth.cube(shape, seed=29, dtype=dt, kind="mean")
        got = _run(x, mu, n_f, n_p, ref=clean, stream=(rows, k))
        oref = _oracle(oracle, x, mu, n_f, n_p, reference_data=clean)
        assert bits_equal(got[0], oref["recon"])
        np.testing.assert_allclose(got[2], oref["MSE64"], rtol=2e-7 if dt == np.float32 else 1e-12)
    x[0].flat[3] = np.inf                                 # exact wrap at the top face, passes in flight on both sides of a seam
    got = _run(x, mu, n_f, n_p, stream=(rows, k), resident=resident)
    ref = _oracle(oracle, x, mu, n_f, n_p)
    assert np.isnan(ref["recon"][-1]).any() and bits_equal(got[0], ref["recon"])


def test_streamed_hybrid_with_stopping_rule_in_place_and_nonfinite_row(oracle):
    from cytvdn_amd import _lib, synth
    dt = np.dtype(np.float32)
    # stopping rule: one level per pass, kept rows or not
    shape = (12, 5, 8, 12)
    x = synth.cube(shape, seed=31, dtype=dt)
    mu = np.array([1.0, 1.0, 0.5, 0.5], dt)
    want = _run(x, mu, 0, 40, stop=0.02)
    for resident in (5, 12):
        got = _run(x, mu, 0, 40, stop=0.02, stream=(4, 6), resident=resident)
        assert 0 < got[3] == want[3] < 40 and bits_equal(got[0], want[0])
        np.testing.assert_allclose(got[1], want[1], rtol=1e-12)
    # exact Jia-Zhao wrap (non-finite first row) with the first rows resident
    shape = (14, 3, 6, 8)
    x = synth.cube(shape, seed=7, dtype=dt) + dt.type(0.5)
    mu = np.array([1.0, 0.8, 0.5, 0.6], dt)
    x[0, 1, 2, 3] = np.inf
    ref = _oracle(oracle, x, mu, 7, 0)
    for resident in (1, 6, 14):
        got = _run(x, mu, 7, 0, stream=(2, 2), resident=resident)
        assert np.isnan(ref["recon"][-1]).any() and bits_equal(got[0], ref["recon"]), resident
    # in place (data is recon_out), and recon_out overlapping data partly
    shape = (21, 3, 4, 8)
    x = synth.cube(shape, seed=13, dtype=dt) + dt.type(0.25)
    ref = _oracle(oracle, x, mu, 7, 0)
    lam = mu / dt.type(32.0)
    for resident, shift in ((8, 0), (0, 0), (8, 3), (0, 3), (21, 3), (5, -2)):
        plane = x[0].size
        buf = np.zeros(x.size + 8 * plane, dt)
        src = buf[4 * plane:4 * plane + x.size].reshape(shape)
        src[...] = x
        dst = buf[(4 + shift) * plane:(4 + shift) * plane + x.size].reshape(shape)
        a = _lib.RunArgs(dtype=0, ndim=4, bc_mode=2, device=0, n_fista=7, n_plain=0, stream_rows=3, stream_k=2, stream_resident=resident)
        for i, s_ in enumerate(shape):
            a.shape[i] = s_
            a.clip[i] = float((1.0 / lam)[i])
            a.lambda_mu[i] = float((lam / mu).astype(dt)[i])
        sums = np.zeros((7, 3))
        a.data, a.recon_out, a.sums_out = src.ctypes.data, dst.ctypes.data, sums.ctypes.data
        _lib.check(_lib.lib().tvdn_run(C.byref(a)))
        assert bits_equal(dst, ref["recon"]), (resident, shift)
        _check_traces(sums, ref, 7)


@pytest.mark.parametrize("shape,dtype,n_f,n_p,stop,rows,k", [
    ((12, 5, 8, 12), np.float32, 0, 40, 0.02, 4, 6),
    ((11, 3, 7, 9), np.float64, 30, 6, 0.03, 3, 4),
])
def test_streamed_run_with_stopping_rule(oracle, shape, dtype, n_f, n_p, stop, rows, k):
    """With a stopping rule every pass is one iteration deep; both phases and the zero tails as the resident run."""
    from cytvdn_amd import synth
    dt = np.dtype(dtype)
    nd = len(shape)
    x = synth.cube(shape, seed=31, dtype=dt)
    mu = np.array([1.0, 1.0, 0.5, 0.5][:nd], dt)
    want = _run(x, mu, n_f, n_p, stop=stop)
    got = _run(x, mu, n_f, n_p, stop=stop, stream=(rows, k))
    assert 0 < want[3] < n_f + n_p                     # the rule did cut the run short
    assert got[3] == want[3]
    assert bits_equal(got[0], want[0])
    assert np.array_equal(got[1][:, 2] != 0, want[1][:, 2] != 0)
    np.testing.assert_allclose(got[1], want[1], rtol=1e-12)
    ref = _oracle(oracle, x, mu, n_f, n_p, stopping_relative_change=stop)
    assert ref["iters_done"] == got[3]                   # stops where the oracle stops
    assert bits_equal(got[0], ref["recon"])
    ran = got[1][:, 2] != 0
    np.testing.assert_allclose(got[1][ran, 0], ref["b_norm64"][ran], rtol=1e-9)
    np.testing.assert_allclose(got[1][ran, 1], ref["delta64"][ran], rtol=1e-9)


def test_streamed_run_mse_trace_and_nonfinite_first_row(oracle):
    from cytvdn_amd import synth
    dt = np.dtype(np.float32)
    shape = (14, 3, 6, 8)
    x = synth.cube(shape, seed=7, dtype=dt) + dt.type(0.5)
    ref = synth.cube(shape, seed=7, dtype=dt, kind="mean")
    mu = np.array([1.0, 0.8, 0.5, 0.6], dt)
    want = _run(x, mu, 6, 2, ref=ref)
    got = _run(x, mu, 6, 2, ref=ref, stream=(3, 4))
    assert bits_equal(got[0], want[0])
    np.testing.assert_allclose(got[2], want[2], rtol=1e-6)
    oref = _oracle(oracle, x, mu, 6, 2, reference_data=ref)
    assert bits_equal(got[0], oref["recon"])
    # the device squares the f32 difference in f32 before the f64 sum, the yardstick squares in f64: one f32 rounding per term
    np.testing.assert_allclose(got[2], oref["MSE64"], rtol=2e-7)
    # an Inf in the cube's first row turns the wrapped axis-0 accumulator into NaN upstream (anisotropic.pyx:65-73):
    # the streamed run keeps row 0 of every level aside and must reproduce the resident run's NaN pattern
    x[0, 1, 2, 3] = np.inf
    want = _run(x, mu, 5, 0)
    got = _run(x, mu, 5, 0, stream=(4, 3))
    assert np.isnan(want[0]).any()
    assert bits_equal(got[0], want[0])
    oref = _oracle(oracle, x, mu, 5, 0)
    assert np.isnan(oref["recon"][-1]).any() and bits_equal(got[0], oref["recon"])   # NaNs where the oracle has them
    # ... and over four passes of two levels
    want = _run(x, mu, 7, 0)
    got = _run(x, mu, 7, 0, stream=(2, 2))
    assert bits_equal(got[0], want[0])
    assert bits_equal(got[0], _oracle(oracle, x, mu, 7, 0)["recon"])
    # a hybrid schedule whose d -> b transition falls INSIDE a pass, on rings, with the non-finite first row
    got = _run(x, mu, 3, 3, stream=(3, 4))
    assert bits_equal(got[0], _oracle(oracle, x, mu, 3, 3)["recon"])


def test_streamed_run_argument_checks():
    from cytvdn_amd import _lib, synth
    dt = np.dtype(np.float32)
    x = synth.cube((6, 3, 4, 8), seed=1, dtype=dt)
    mu = np.array([1.0, 0.8, 0.5, 0.6], dt)
    with pytest.raises(ValueError, match="must both be 0"):
        _run(x, mu, 2, 0, stream=(4, 0))
    with pytest.raises(ValueError, match="must both be 0"):
        _run(x, mu, 2, 0, stream=(-1, 3))
    # -1 / -1 = "stream when needed": this cube fits, so the resident engine runs
    want = _run(x, mu, 3, 0)
    got = _run(x, mu, 3, 0, stream=(-1, -1))
    assert bits_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


def test_streamed_run_in_place(oracle):
    """data and recon_out the SAME array (the resident run allows it): a streamed run of several passes must
    not feed its own output back as the data term."""
    from cytvdn_amd import _lib, synth
    dt = np.dtype(np.float32)
    shape = (21, 3, 4, 8)
    x = synth.cube(shape, seed=13, dtype=dt) + dt.type(0.25)
    mu = np.array([1.0, 0.8, 0.5, 0.6], dt)
    ref = _oracle(oracle, x, mu, 7, 0)
    for stream in (None, (3, 2)):                        # resident, then streamed in four passes
        buf = x.copy()
        lam = mu / dt.type(32.0)
        a = _lib.RunArgs(dtype=0, ndim=4, bc_mode=2, device=0, n_fista=7, n_plain=0)
        if stream:
            a.stream_rows, a.stream_k = stream
        for i, s_ in enumerate(shape):
            a.shape[i] = s_
            a.clip[i] = float((1.0 / lam)[i])
            a.lambda_mu[i] = float((lam / mu).astype(dt)[i])
        sums = np.zeros((7, 3))
        a.data = a.recon_out = buf.ctypes.data
        a.sums_out = sums.ctypes.data
        _lib.check(_lib.lib().tvdn_run(C.byref(a)))
        assert bits_equal(buf, ref["recon"]), stream
        _check_traces(sums, ref, 7)


@pytest.mark.parametrize("shape,dtype,n_f,n_p,rows,k,stop", [
    ((23, 3, 4, 8), np.float32, 9, 0, 5, 3, None),         # three passes; the two ends are each other's halo
    ((17, 6, 16), np.float64, 4, 3, 4, 6, None),           # hybrid, the d -> b transition inside the second pass
    ((6, 3, 4, 8), np.float32, 7, 0, 2, 9, None),          # k capped at the cube height
    ((15, 3, 4, 8), np.float32, 0, 8, 1, 5, None),         # one-row chunks, unaccelerated
    ((11, 2, 5, 7), np.float32, 6, 0, 3, 2, None),         # scalar packs
    ((12, 5, 8, 12), np.float32, 0, 40, 4, 6, 0.02),       # stopping rule: one level per pass
])
def test_streamed_run_with_periodic_boundaries(oracle, shape, dtype, n_f, n_p, rows, k, stop):
    """bc_mode 0 through the streamed tvdn_run: the cube between k wrapped rows at either end (which give up a row per
    level, like the face between two slabs), old and new host state in separate arrays.  Recon and traces are the
    oracle's (reference loop with BC_mode=0: cyTVDN.py:148-242, anisotropic.pyx:65-73, utils.pyx:98-101), the sums count
    every row of the cube exactly once, and an MSE trace counts own rows only."""
    from cytvdn_amd import synth
    dt = np.dtype(dtype)
    nd = len(shape)
    x = synth.cube(shape, seed=71, dtype=dt) + dt.type(0.25)
    mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
    refc = synth.cube(shape, seed=71, dtype=dt, kind="mean")
    x_before = x.copy()
    got = _run(x, mu, n_f, n_p, stop=stop, ref=refc, stream=(rows, k), bc=0)
    want = _run(x, mu, n_f, n_p, stop=stop, ref=refc, bc=0)              # the resident run wraps for real
    assert bits_equal(x, x_before)
    assert got[3] == want[3] and bits_equal(got[0], want[0])
    ref = _oracle_bc(oracle, x, mu, n_f, n_p, 0, stopping_relative_change=stop, reference_data=refc)
    assert bits_equal(got[0], ref["recon"]) and ref["iters_done"] == got[3]
    ran = got[1][:, 2] != 0
    assert ran.sum() == got[3]
    np.testing.assert_allclose(got[1][ran, 0], ref["b_norm64"][ran], rtol=1e-9)
    np.testing.assert_allclose(got[1][ran, 1], ref["delta64"][ran], rtol=1e-9)
    np.testing.assert_allclose(got[1][ran, 2], ref["rnorm64"][ran], rtol=1e-9)
    n_run = np.nonzero(ran)[0]
    np.testing.assert_allclose(got[2][0], ref["MSE64"][0], rtol=2e-7)
    np.testing.assert_allclose(got[2][n_run + 1], ref["MSE64"][n_run + 1], rtol=2e-7)


@pytest.mark.parametrize("shape,dtype,n_f,n_p,rows,k,bc,slabs,stop", [
    ((23, 3, 4, 8), np.float32, 9, 0, 5, 3, 2, 2, None),       # two slabs (11 + 12 rows), three passes
    ((23, 3, 4, 8), np.float32, 9, 0, 2, 4, 2, 3, None),       # three slabs, passes of 3 + 3 + 3 levels
    ((23, 3, 4, 8), np.float32, 5, 4, 3, 9, 2, 3, None),       # hybrid in ONE pass deeper than a slab is tall: halos reach past the neighbour
    ((40, 4, 8, 16), np.float32, 7, 6, 3, 4, 2, 4, None),      # hybrid over four passes, four slabs
    ((17, 6, 16), np.float64, 0, 7, 3, 5, 2, 2, None),         # unaccelerated, 3-D, f64
    ((17, 6, 16), np.float64, 7, 0, 4, 2, 0, 3, None),         # periodic: every face artificial, the first and the last slab wrap
    ((23, 3, 4, 8), np.float32, 9, 0, 5, 3, 0, 2, None),
    ((9, 2, 5, 7), np.float32, 6, 0, 2, 8, 2, 3, None),        # scalar packs; more levels than rows
    ((12, 5, 8, 12), np.float32, 0, 40, 4, 6, 2, 3, 0.02),     # the GLOBAL stopping rule: one level per pass, sums over the slabs
    ((11, 3, 7, 9), np.float64, 30, 6, 3, 4, 0, 2, 0.03),      # ... periodic, both phases
])
def test_streamed_device_list(oracle, shape, dtype, n_f, n_p, rows, k, bc, slabs, stop):
    """A device list whose slabs are STREAMED (BASELINE configs[4] in structure, inside one process): the state of the whole
    cube in page-locked host arrays shared by the slabs (two sets), every slab streamed through its device (here: several
    slabs on device 0) reading k rows of its neighbours' state beyond each interior face, the slabs meeting after every pass;
    global sums, global stopping rule, MSE trace.  The oracle's bits (the reference has no counterpart that works:
    cyTVDN/mpi.py:131-239 + :314-434, SURVEY B-7)."""
    from cytvdn_amd import _lib, synth
    dt = np.dtype(dtype)
    nd = len(shape)
    x = synth.cube(shape, seed=37, dtype=dt) + dt.type(0.25)
    mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
    n = n_f + n_p
    st = _lib.RunStats()
    got = _run(x, mu, n_f, n_p, stop=stop, stream=(rows, k), bc=bc, devices=[0] * slabs, stats=st)
    ref = _oracle_bc(oracle, x, mu, n_f, n_p, bc, **({"stopping_relative_change": stop} if stop is not None else {}))
    assert bits_equal(got[0], ref["recon"])
    assert st.engine == 1
    if stop is None:
        assert got[3] == n
        _check_traces(got[1], ref, n)
    else:
        assert 0 < got[3] == ref["iters_done"] < n
        ran = got[1][:, 2] != 0
        np.testing.assert_allclose(got[1][ran, 0], ref["b_norm64"][ran], rtol=1e-9)
        np.testing.assert_allclose(got[1][ran, 1], ref["delta64"][ran], rtol=1e-9)


def test_streamed_device_list_mse_in_place_and_nonfinite_first_row(oracle):
    from cytvdn_amd import _lib, synth
    dt = np.dtype(np.float32)
    shape = (21, 3, 6, 8)
    x = synth.cube(shape, seed=41, dtype=dt) + dt.type(0.5)
    clean = synth.cube(shape, seed=41, dtype=dt, kind="mean")
    mu = np.array([1.0, 0.8, 0.5, 0.6], dt)
    got = _run(x, mu, 6, 2, ref=clean, stream=(3, 3), devices=[0, 0, 0])
    oref = _oracle(oracle, x, mu, 6, 2, reference_data=clean)
    assert bits_equal(got[0], oref["recon"])
    np.testing.assert_allclose(got[2], oref["MSE64"], rtol=2e-7)
    # in place: data is recon_out
    lam = mu / dt.type(32.0)
    buf = x.copy()
    a = _lib.RunArgs(dtype=0, ndim=4, bc_mode=2, n_fista=7, n_plain=0, stream_rows=3, stream_k=2, n_devices=2)
    for i, s_ in enumerate(shape):
        a.shape[i] = s_
        a.clip[i] = float((1.0 / lam)[i])
        a.lambda_mu[i] = float((lam / mu).astype(dt)[i])
    sums = np.zeros((7, 3))
    a.data = a.recon_out = buf.ctypes.data
    a.sums_out = sums.ctypes.data
    _lib.check(_lib.lib().tvdn_run(C.byref(a)))
    ref = _oracle(oracle, x, mu, 7, 0)
    assert bits_equal(buf, ref["recon"])
    _check_traces(sums, ref, 7)
    # a non-finite first row: the exact Jia-Zhao wrap, row 0 of every level handed from the first slab's thread to the last's
    x[0, 1, 2, 3] = np.inf
    x[0, 2, 1, 5] = np.nan
    for devs, n_f, n_p, stream, stop in (([0, 0], 5, 0, (3, 2), None), ([0, 0, 0], 4, 3, (2, 3), None), ([0, 0, 0, 0], 6, 0, (2, 4), 1e-9)):
        got = _run(x, mu, n_f, n_p, stream=stream, devices=devs, stop=stop)
        ref = _oracle(oracle, x, mu, n_f, n_p, stopping_relative_change=stop)
        assert np.isnan(ref["recon"][-1]).any() and bits_equal(got[0], ref["recon"]), devs


@pytest.mark.parametrize("shape,dtype,n_f,n_p,rows,k,chain", [
    ((24, 3, 4, 8), np.float32, 12, 0, 2, 4, "1"), ((24, 3, 4, 8), np.float32, 12, 0, 2, 4, "0"),
    ((24, 3, 4, 8), np.float32, 5, 7, 3, 4, "1"),            # hybrid: the last FISTA pass leaves b, the rebuild follows the form
    ((30, 6, 16), np.float64, 0, 9, 1, 3, "1"), ((30, 6, 16), np.float64, 8, 0, 4, 2, "0"),
])
def test_recon_does_not_cross_pcie_between_passes(oracle, monkeypatch, shape, dtype, n_f, n_p, rows, k, chain):
    """Round 5: a pass that continues a run uploads the data term and the accumulator state only -- recon is rebuilt from them on
    the device (csrc/tvdn_rebuild.hip: the reconstruction update of cyTVDN/utils.pyx:90-104 applied to state that is already
    there) -- and only the last pass brings recon down.  Same bits as the oracle and as the run that ships recon
    (TVDN_STREAM_SHIP_RECON=1, round 4's way); the byte counts say which of the two ran."""
    from cytvdn_amd import _lib, synth
    monkeypatch.setenv("TVDN_STREAM_CHAIN", chain)
    dt = np.dtype(dtype)
    nd = len(shape)
    x = synth.cube(shape, seed=43, dtype=dt) + dt.type(0.25)
    mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
    n = n_f + n_p
    ref = _oracle(oracle, x, mu, n_f, n_p)
    cube = x.nbytes
    seen = {}
    for ship in (False, True):
        if ship:
            monkeypatch.setenv("TVDN_STREAM_SHIP_RECON", "1")
        else:
            monkeypatch.delenv("TVDN_STREAM_SHIP_RECON", raising=False)
        st = _lib.RunStats()
        got = _run(x, mu, n_f, n_p, stream=(rows, k), resident=0, stats=st)
        assert bits_equal(got[0], ref["recon"]) and got[3] == n, ship
        _check_traces(got[1], ref, n)
        seen[ship] = (st.h2d_bytes, st.d2h_bytes, st.n_passes)
    P = seen[False][2]
    assert P == seen[True][2] >= 3
    # arrays of state a pass carries: d pairs while the iterations before / after it are FISTA ones, b afterwards
    depth = [n // P + (1 if q < n % P else 0) for q in range(P)]
    start = [sum(depth[:q]) for q in range(P)]
    n_in = [0 if q == 0 else nd * (2 if start[q] <= n_f and n_f > 0 and start[q] - 1 < n_f else 1) for q in range(P)]
    n_out = [nd * (2 if start[q] + depth[q] <= n_f and n_f > 0 else 1) for q in range(P)]
    up_free = cube * sum(1 + n_in[q] for q in range(P))
    down_free = cube * (sum(n_out) + 1)                       # ... + recon once, with the last pass
    up_ship = cube * sum(1 + n_in[q] + (1 if q else 0) for q in range(P))
    down_ship = cube * sum(n_out[q] + 1 for q in range(P))
    assert seen[False][:2] == (up_free, down_free), (seen, up_free, down_free)
    assert seen[True][:2] == (up_ship, down_ship), (seen, up_ship, down_ship)


@pytest.mark.parametrize("shape,dtype,n_f,n_p,rows,k,bc,resident,devices", [
    ((40, 4, 8, 16), np.float32, 12, 0, 2, 4, 2, 0, None),       # three chained passes, every row streamed
    ((40, 4, 8, 16), np.float32, 8, 0, 4, 4, 2, 0, None),        # two drained passes
    ((40, 4, 8, 16), np.float32, 9, 3, 3, 4, 2, 17, None),       # hybrid schedule, rows kept among streamed ones
    ((24, 6, 16), np.float64, 9, 0, 2, 3, 0, 0, None),           # periodic: the drained pass with two sets of host state
    ((40, 4, 8, 16), np.float32, 8, 0, 4, 4, 2, 0, [0, 0, 0]),   # a device list: every slab its own helper thread
])
def test_rows_come_down_the_same_whichever_way(oracle, monkeypatch, shape, dtype, n_f, n_p, rows, k, bc, resident, devices):
    """Downloads by the runtime's copies one at a time from a helper thread (the default, csrc/tvdn_stream_parts.hpp DownPump), by
    round 4's copy kernel (TVDN_STREAM_DOWN_PUMP=0) and by copies queued behind events (TVDN_STREAM_DOWN_BLOCKS=0): the same bits
    and the same bytes across PCIe, and the oracle's."""
    from cytvdn_amd import _lib, synth
    dt = np.dtype(dtype)
    nd = len(shape)
    x = synth.cube(shape, seed=31, dtype=dt) + dt.type(0.25)
    mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
    results = []
    for env in ({}, {"TVDN_STREAM_DOWN_PUMP": "0"}, {"TVDN_STREAM_DOWN_BLOCKS": "0"}):
        for key in ("TVDN_STREAM_DOWN_PUMP", "TVDN_STREAM_DOWN_BLOCKS"):
            monkeypatch.delenv(key, raising=False)
        for key, v in env.items():
            monkeypatch.setenv(key, v)
        st = _lib.RunStats()
        got = _run(x, mu, n_f, n_p, stream=(rows, k), resident=resident, bc=bc, stats=st, devices=devices)
        results.append((got[0], got[1], st.h2d_bytes, st.d2h_bytes))
    for r in results[1:]:
        assert bits_equal(r[0], results[0][0])
        np.testing.assert_allclose(r[1], results[0][1], rtol=1e-12)
        assert (r[2], r[3]) == (results[0][2], results[0][3])
    ref = _oracle_bc(oracle, x, mu, n_f, n_p, bc)
    assert bits_equal(results[0][0], ref["recon"])


@pytest.mark.parametrize("shape,dtype,n_f,n_p,rows,k,slabs,stop,bad", [
    ((40, 4, 8, 16), np.float32, 9, 0, 3, 3, 2, None, False),     # two slabs of 20 rows: 17 + 17 interior rows may be kept
    ((40, 4, 8, 16), np.float32, 7, 6, 2, 4, 3, None, False),     # hybrid schedule over four passes, three slabs
    ((41, 6, 16), np.float64, 0, 9, 4, 3, 2, None, False),        # unaccelerated, 3-D, f64
    ((40, 4, 8, 16), np.float32, 8, 0, 3, 2, 4, None, True),      # a non-finite first row: the exact wrap's planes handed between the slabs
    ((30, 3, 5, 7), np.float32, 0, 40, 4, 6, 3, 0.02, False),     # the global stopping rule: one level per pass
])
def test_streamed_device_list_keeps_interior_rows(oracle, shape, dtype, n_f, n_p, rows, k, slabs, stop, bad):
    """Slabs of a streamed device list keep their interior rows -- none of the k a neighbour reads at a shared face -- in HBM between
    the passes, as far as they fit their share of their device (stream_resident != 0); the shared host arrays stay indexed by cube
    row, the kept rows' slots simply go unused.  The oracle's bits, fewer bytes across PCIe than with every row streamed."""
    from cytvdn_amd import _lib, synth
    dt = np.dtype(dtype)
    nd = len(shape)
    x = synth.cube(shape, seed=43, dtype=dt) + dt.type(0.25)
    if bad:
        x[0, 1, 2, 3] = np.inf
    mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
    n = n_f + n_p
    st, st0 = _lib.RunStats(), _lib.RunStats()
    got = _run(x, mu, n_f, n_p, stop=stop, stream=(rows, k), devices=[0] * slabs, resident=-1, stats=st)
    none = _run(x, mu, n_f, n_p, stop=stop, stream=(rows, k), devices=[0] * slabs, resident=0, stats=st0)
    ref = _oracle(oracle, x, mu, n_f, n_p, **({"stopping_relative_change": stop} if stop is not None else {}))
    assert bits_equal(got[0], ref["recon"]) and bits_equal(none[0], ref["recon"])
    depth = 1 if stop is not None else k
    interior = sum(max(0, (r + 1) * shape[0] // slabs - r * shape[0] // slabs - (depth if r > 0 else 0) - (depth if r < slabs - 1 else 0))
                   for r in range(slabs))
    assert st.engine == 1 and st.resident_rows == interior > 0 and st0.resident_rows == 0
    assert st.h2d_bytes < st0.h2d_bytes and st.d2h_bytes < st0.d2h_bytes
    if stop is None and not bad:
        assert got[3] == n
        _check_traces(got[1], ref, n)
