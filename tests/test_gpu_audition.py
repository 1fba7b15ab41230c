"""Placement audition (engine.HipBackend.best_of): trying several allocations of the state and keeping the fastest must
change nothing but the speed -- the kept backend is as freshly constructed (all-zero state, roles reset), so a run after
an audition gives the bits of a run without one, which are the oracle's (loop: cyTVDN/cyTVDN.py:148-242)."""
import numpy as np
import pytest

from golden_util import bits_equal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,dtype,its,fista", [
    ((9, 5, 8, 16), np.float32, 7, True), ((12, 6, 16), np.float64, [3, 2], True), ((7, 3, 4, 8), np.float32, 5, False),
])
def test_run_after_an_audition_matches_the_oracle(oracle, monkeypatch, shape, dtype, its, fista):
    import cytvdn_amd as tv
    from cytvdn_amd import synth
    dt = np.dtype(dtype)
    nd = len(shape)
    x = synth.cube(shape, seed=19, dtype=dt) + dt.type(0.25)
    mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
    fn = tv.denoise4D if nd == 4 else tv.denoise3D
    ref = oracle.denoise(x, mu, its, fista)
    for cand in ("1", "3"):
        monkeypatch.setenv("TVDN_AUDITION", cand)
        recon, bn, dl = fn(x, mu, its, FISTA=fista, quiet=True)
        assert bits_equal(recon, ref["recon"]), cand
        np.testing.assert_allclose(bn, ref["b_norm64"], rtol=1e-5)


def test_best_of_reports_its_candidates_and_frees_the_losers():
    import torch
    from cytvdn_amd.engine import HipBackend, SlabLayout
    shape = (64, 32, 64, 64)                                       # 32 MiB per array: 15 arrays = 480 MiB per candidate
    lay = SlabLayout(shape, 0, 1, 2)
    torch.cuda.empty_cache()
    free0, _ = torch.cuda.mem_get_info(0)
    be = HipBackend.best_of(3, lay, np.float32, True, device=0, max_iters=4)
    assert len(be.audition) == 3 and be.audition[0] == min(be.audition) and all(t > 0 for t in be.audition)
    assert be.cur == 0 and be.d_form and be.tk_prev == 0.0
    assert float(be.sums.abs().sum()) == 0.0 and float(be.S[0][0].abs().sum()) == 0.0 and float(be.recon[1].abs().sum()) == 0.0
    free1, _ = torch.cuda.mem_get_info(0)
    assert free0 - free1 < 2 * be.state_bytes()                    # the two losers went back to the driver
    one = HipBackend.best_of(1, lay, np.float32, True, device=0, max_iters=4)
    assert one.audition == []


@pytest.mark.parametrize("shape,dtype,n_f,n_p", [((9, 5, 8, 16), np.float32, 6, 0), ((11, 6, 16), np.float64, 3, 3), ((7, 3, 4, 8), np.float32, 0, 5)])
def test_tvdn_run_auditions_too(oracle, monkeypatch, shape, dtype, n_f, n_p):
    """The C entry point tries several placements of its state before long runs as well (csrc/tvdn_run.hip); forced here
    on small cubes: same bits, same traces as the oracle, with and without."""
    import ctypes as C
    from cytvdn_amd import _lib, synth
    dt = np.dtype(dtype)
    nd = len(shape)
    x = synth.cube(shape, seed=23, dtype=dt) + dt.type(0.25)
    mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
    lam = mu / dt.type(32.0 if nd == 4 else 16.0)
    ref = oracle.denoise(x, mu, [n_f, n_p] if (n_f and n_p) else (n_f or n_p), bool(n_f))
    for cand in ("1", "3"):
        monkeypatch.setenv("TVDN_AUDITION", cand)
        a = _lib.RunArgs(dtype=_lib.dtype_code(dt), ndim=nd, bc_mode=2, device=0, n_fista=n_f, n_plain=n_p)
        for i, s_ in enumerate(shape):
            a.shape[i] = s_
        for q in range(nd):
            a.clip[q] = float((1.0 / lam)[q])
            a.lambda_mu[q] = float((lam / mu).astype(dt)[q])
        recon, sums = np.empty_like(x), np.zeros((n_f + n_p, 3))
        a.data, a.recon_out, a.sums_out = x.ctypes.data, recon.ctypes.data, sums.ctypes.data
        _lib.check(_lib.lib().tvdn_run(C.byref(a)))
        assert bits_equal(recon, ref["recon"]), cand
        np.testing.assert_allclose(sums[:, 0], ref["b_norm64"], rtol=1e-9)
        np.testing.assert_allclose(sums[:, 1], ref["delta64"], rtol=1e-9)
