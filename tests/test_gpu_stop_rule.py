"""A stopping rule looked at one iteration behind (csrc/tvdn_run.hip, "speculation by one"): the resident tvdn_run queues iteration
i+1 before it reads the sums of iteration i, and takes iteration i+1 back when iteration i satisfied the rule.  Whatever the
slot the rule fires at -- the first, the last, the last accelerated one of a hybrid run, the middle of either phase -- the result,
the traces, their zero tails and the iteration counts are the oracle's (cyTVDN.py:189-201: a FISTA-phase break falls through to
the unaccelerated phase), and the same as the blocking form's (TVDN_STOP_LAG=0)."""
import os

import numpy as np
import pytest

from golden_util import bits_equal

pytestmark = pytest.mark.gpu


def _thresholds(delta, slots):
    """for each wanted slot k a threshold that delta first drops below AT k (None where the trace does not allow one)"""
    out = {}
    for k in slots:
        if k >= len(delta) or not np.isfinite(delta[k]) or delta[k] <= 0:
            continue
        before = delta[:k]
        if k == 0:
            out[k] = float(delta[0]) * 2.0
        elif before.min() > delta[k] * 1.0001:
            out[k] = float(np.sqrt(float(before.min()) * float(delta[k])))
    return out


def _call(fn, x, mu, its, fista, thr, refd, bc, devs, lag):
    kw = {} if devs is None else {"device": devs}
    prev = os.environ.get("TVDN_STOP_LAG")
    try:
        if lag is None:
            os.environ.pop("TVDN_STOP_LAG", None)
        else:
            os.environ["TVDN_STOP_LAG"] = str(lag)
        return fn(x, mu, its, FISTA=fista, stopping_relative_change=thr, reference_data=refd, BC_mode=bc, quiet=True, **kw)
    finally:
        if prev is None:
            os.environ.pop("TVDN_STOP_LAG", None)
        else:
            os.environ["TVDN_STOP_LAG"] = prev


@pytest.mark.parametrize("shape,dtype,its,fista,with_ref,bc,devs", [
    ((9, 6, 8, 12), np.float32, 14, True, False, 2, None),          # FISTA only
    ((9, 6, 8, 12), np.float32, 14, False, True, 2, None),          # unaccelerated only, MSE trace
    ((8, 7, 16), np.float64, [8, 7], True, True, 2, None),          # hybrid, 3-D, f64, MSE trace
    ((10, 5, 8, 8), np.float32, [7, 8], True, False, 0, None),      # hybrid, periodic boundaries
    ((23, 4, 8, 12), np.float32, [7, 8], True, True, 2, [0, 0, 0]),  # hybrid over three slabs: the global criterion
    ((16, 6, 16), np.float64, 12, True, False, 0, [0, 0]),          # periodic ring of two slabs
])
def test_the_rule_fires_at_every_kind_of_slot(oracle, shape, dtype, its, fista, with_ref, bc, devs):
    import cytvdn_amd as tv
    from cytvdn_amd import synth
    dt = np.dtype(dtype)
    nd = len(shape)
    x = synth.cube(shape, seed=53, dtype=dt) + dt.type(0.25)
    refd = synth.cube(shape, seed=53, dtype=dt, kind="mean") if with_ref else None
    mu = np.array([1.0, 1.0, 0.5, 0.5][:nd] if nd == 4 else [1.0, 1.0, 0.5], dt)
    fn = tv.denoise4D if nd == 4 else tv.denoise3D
    n_f, n_p = (its if isinstance(its, list) else ((its, 0) if fista else (0, its)))
    n = n_f + n_p
    free = oracle.denoise(x, mu, its, fista, reference_data=refd, BC_mode=bc)
    d64 = free["delta64"] / free["rnorm64"]
    # the first slot, the second, the middle of the first phase, its last slot (in a hybrid run the speculated iteration IS the
    # one to resume with), the last slot of the run (nothing to speculate)
    first_phase = n_f if n_f else n_p
    slots = sorted({0, 1, first_phase // 2, first_phase - 1, n - 1})
    thr = _thresholds(d64, slots)
    assert len(thr) >= 3, (slots, thr)
    fired_at = set()
    for k, t in thr.items():
        ref = oracle.denoise(x, mu, its, fista, stopping_relative_change=t, reference_data=refd, BC_mode=bc)
        got = _call(fn, x, mu, its, fista, t, refd, bc, devs, None)
        blocking = _call(fn, x, mu, its, fista, t, refd, bc, devs, 0)
        # the two forms: same bits everywhere
        assert len(got) == len(blocking)
        for u, v in zip(got, blocking):
            assert u.dtype == v.dtype and bits_equal(u, v)
        # the oracle: the result bit for bit, the same slots run, the same zero tails
        assert bits_equal(got[0], ref["recon"])
        assert np.array_equal(got[2] != 0, ref["delta_recon"] != 0), (k, got[2], ref["delta_recon"])
        assert np.array_equal(got[1] != 0, ref["b_norm"] != 0)
        # (the traces against the oracle's float64 yardsticks: the device sums are float64 trees, the reference's own sums are of
        #  the data's width and depend on its thread count)
        tol = 3e-7 if dt == np.float32 else 1e-11
        ran_mask = ref["delta_recon"] != 0
        np.testing.assert_allclose(got[1][ran_mask], ref["b_norm64"][ran_mask], rtol=tol)
        np.testing.assert_allclose(got[2][ran_mask], ref["delta64"][ran_mask] / ref["rnorm64"][ran_mask], rtol=tol)
        if with_ref:
            assert np.array_equal(got[3] != 0, ref["MSE"] != 0)
            np.testing.assert_allclose(got[3], ref["MSE64"], rtol=tol)
        ran = np.nonzero(ref["delta_recon"])[0]
        assert ran[0] == 0 and (k in ran)
        fired_at.add(k)
        if n_f and n_p and k < n_f - 1:
            # the accelerated phase stopped early: slots k+1 .. n_f-1 never ran (one of them DID, ahead of the rule, and was taken
            # back), and the unaccelerated phase started from slot n_f on the state iteration k left
            assert not got[2][k + 1:n_f].any() and got[2][n_f] != 0
    assert len(fired_at) >= 3


def test_a_rule_that_never_fires_changes_nothing(oracle):
    """... and the run with a rule nobody meets is the run without one, traces and all."""
    import cytvdn_amd as tv
    from cytvdn_amd import synth
    dt = np.dtype(np.float32)
    shape = (12, 16, 64)
    x = synth.cube(shape, seed=59, dtype=dt)
    mu = np.array([1.0, 1.0, 0.5], dt)
    plain = tv.denoise3D(x, mu, [9, 6], FISTA=True, quiet=True)
    ruled = tv.denoise3D(x, mu, [9, 6], FISTA=True, stopping_relative_change=1e-30, quiet=True)
    for u, v in zip(plain, ruled):
        assert bits_equal(u, v)
    ref = oracle.denoise(x, mu, [9, 6], True)
    assert bits_equal(ruled[0], ref["recon"])


def test_calls_from_several_threads_share_what_runs_keep(oracle):
    """ctypes releases the GIL inside tvdn_run, so denoise3D/4D may run on several threads at once.  What a resident run keeps for the
    next one of its device -- reduction context, streams, sums buffer (csrc/tvdn_run_state.hip kit_acquire / kit_release) -- goes to
    ONE run at a time: four threads, calls with and without a stopping rule interleaved, every result the oracle's."""
    import threading
    import cytvdn_amd as tv
    from cytvdn_amd import synth, _lib
    dt = np.dtype(np.float32)
    cases = []
    for i, (shape, its, stop) in enumerate([((9, 10, 32), 12, None), ((7, 6, 8, 12), [5, 4], 0.05), ((12, 8, 16), 15, 0.02),
                                            ((6, 5, 8, 8), 9, None)]):
        nd = len(shape)
        x = synth.cube(shape, seed=71 + i, dtype=dt) + dt.type(0.25)
        mu = np.array([1.0, 1.0, 0.5, 0.5][:nd] if nd == 4 else [1.0, 1.0, 0.5], dt)
        ref = oracle.denoise(x, mu, its, True, stopping_relative_change=stop)
        cases.append((tv.denoise4D if nd == 4 else tv.denoise3D, x, mu, its, stop, ref))
    errors = []

    def work(k):
        try:
            for rep in range(6):
                fn, x, mu, its, stop, ref = cases[(k + rep) % len(cases)]
                got = fn(x, mu, its, FISTA=True, stopping_relative_change=stop, quiet=True)
                if not bits_equal(got[0], ref["recon"]) or not np.array_equal(got[2] != 0, ref["delta_recon"] != 0):
                    errors.append((k, rep, "differs from the oracle"))
        except Exception as e:      # noqa: BLE001 -- reported below
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    _lib.check(_lib.lib().tvdn_release_cache())      # the kept kit goes back with the kept state
    got = cases[0][0](cases[0][1], cases[0][2], cases[0][3], FISTA=True, quiet=True)   # ... and the next run makes its own
    assert bits_equal(got[0], cases[0][5]["recon"])


@pytest.mark.parametrize("shape,dtype,n_f,n_p", [((26, 4, 8, 12), np.float32, 14, 0), ((24, 5, 16), np.float64, 7, 8), ((28, 3, 8, 8), np.float32, 0, 14)])
def test_a_rule_with_a_pipelined_start(oracle, monkeypatch, shape, dtype, n_f, n_p):
    """With a stopping rule the START of a resident run is still pipelined (csrc/tvdn_run.hip): three iterations follow the upload
    chunk by chunk, their sums are looked at when the last chunk has been swept, the rest of the run looks at every iteration one
    behind.  A rule met INSIDE those first iterations sends the whole run round again in plain order (stats.pipelined says which
    happened); whichever slot it fires at, the bits are the oracle's."""
    import ctypes as C
    from cytvdn_amd import _lib, synth
    dt = np.dtype(dtype)
    nd = len(shape)
    x = synth.cube(shape, seed=67, dtype=dt) + dt.type(0.25)
    mu = np.array([1.0, 1.0, 0.5, 0.5][:nd] if nd == 4 else [1.0, 1.0, 0.5], dt)
    lam = mu / dt.type(32.0 if nd == 4 else 16.0)
    its = [n_f, n_p] if (n_f and n_p) else (n_f or n_p)
    fista = n_f > 0
    n = n_f + n_p
    free = oracle.denoise(x, mu, its, fista)
    d64 = free["delta64"] / free["rnorm64"]
    k0 = 3
    thr = _thresholds(d64, sorted({0, 1, k0 - 1, k0, k0 + 1, (n_f or n_p) - 1, n - 1}))
    thr[None] = 1e-30          # never met
    assert len(thr) >= 5
    monkeypatch.setenv("TVDN_PIPELINE", f"4,{k0},2")
    seen_pipelined, seen_retry = 0, 0
    for k, t in thr.items():
        ref = oracle.denoise(x, mu, its, fista, stopping_relative_change=t)
        a = _lib.RunArgs(dtype=_lib.dtype_code(dt), ndim=nd, bc_mode=2, device=0, n_fista=n_f, n_plain=n_p, use_stop=1, stop=float(t))
        for i, s in enumerate(shape):
            a.shape[i] = s
        for q in range(nd):
            a.clip[q] = float((1.0 / lam)[q])
            a.lambda_mu[q] = float((lam / mu).astype(dt)[q])
        recon = np.empty_like(x)
        sums = np.zeros((n, 3))
        phases = (C.c_int32 * 2)(0, 0)
        st = _lib.RunStats()
        a.data, a.recon_out, a.sums_out = x.ctypes.data, recon.ctypes.data, sums.ctypes.data
        a.phase_iters = C.addressof(phases)
        a.stats = C.addressof(st)
        _lib.check(_lib.lib().tvdn_run(C.byref(a)))
        assert bits_equal(recon, ref["recon"]), k
        ran = ref["delta_recon"] != 0
        assert np.array_equal(sums[:, 2] != 0, ran), (k, sums[:, 2], ref["delta_recon"])
        assert phases[0] == int(ran[:n_f].sum()) and phases[1] == int(ran[n_f:].sum())
        tol = 3e-7 if dt == np.float32 else 1e-11
        np.testing.assert_allclose(sums[ran, 0], ref["b_norm64"][ran], rtol=tol)
        np.testing.assert_allclose(sums[ran, 1] / sums[ran, 2], ref["delta64"][ran] / ref["rnorm64"][ran], rtol=tol)
        first_stop = int(np.nonzero(ran)[0][-1]) if k is not None else n
        if k is not None and k < k0:
            assert st.pipelined == 0      # met inside the iterations that followed the upload: done again in plain order
            seen_retry += 1
        else:
            assert st.pipelined == 1
            seen_pipelined += 1
        del first_stop
    assert seen_pipelined >= 2 and seen_retry >= 1


@pytest.mark.parametrize("shape,its,fista,bc", [((9, 6, 8, 10), [5, 4], True, 2), ((11, 7, 14), 9, True, 0), ((8, 5, 6, 126), 6, False, 2),
                                               ((10, 12, 510), 7, True, 2), ((7, 3, 4, 2), 5, True, 0)])
def test_an_even_last_axis_that_is_no_multiple_of_four(oracle, monkeypatch, shape, its, fista, bc):
    """float32 cubes whose last axis is even but no multiple of four elements (126, 510 channels: detectors and spectrometers do
    come like that) run on one element per thread (csrc/tvdn_fused.hip; packs of 8 bytes were measured no faster): the oracle's
    bits -- resident, over slabs, and streamed."""
    import cytvdn_amd as tv
    from cytvdn_amd import synth
    dt = np.dtype(np.float32)
    nd = len(shape)
    x = synth.cube(shape, seed=83, dtype=dt) + dt.type(0.25)
    mu = np.array([1.0, 1.0, 0.5, 0.5][:nd] if nd == 4 else [1.0, 1.0, 0.5], dt)
    fn = tv.denoise4D if nd == 4 else tv.denoise3D
    ref = oracle.denoise(x, mu, its, fista, BC_mode=bc)
    got = fn(x, mu, its, FISTA=fista, BC_mode=bc, quiet=True)
    assert bits_equal(got[0], ref["recon"])
    np.testing.assert_allclose(got[1], ref["b_norm64"], rtol=3e-7)
    if shape[0] >= 8:
        slabs = fn(x, mu, its, FISTA=fista, BC_mode=bc, quiet=True, device=[0, 0, 0])
        assert bits_equal(slabs[0], ref["recon"])
        if bc == 2:
            monkeypatch.setenv("TVDN_WAVEFRONT", "4,3")
            streamed = fn(x, mu, its, FISTA=fista, BC_mode=bc, quiet=True)
            assert bits_equal(streamed[0], ref["recon"])
