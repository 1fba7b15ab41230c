"""cytvdn_amd/pipelined.py: an in-core denoise3D/4D from NumPy whose first iterations run under the upload and whose last
ones run over the download (a wavefront of partial sweeps on the resident arrays) gives the bits of the plain loop, i.e.
the oracle's (reference loop: cyTVDN/cyTVDN.py:148-242) -- recon and the b_norm / delta_recon traces -- for every chunk
height and depth, FISTA / unaccelerated / hybrid schedules (a transition inside the start or the end wavefront), f32 /
f64, 3-D / 4-D, scalar packs."""
import numpy as np
import pytest

from golden_util import bits_equal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,dtype,its,fista,pipe", [
    ((40, 4, 8, 16), np.float32, 12, True, "8,4,4"),
    ((40, 4, 8, 16), np.float32, 12, True, "5,6,6"),          # chunks that do not divide the cube, no middle part
    ((23, 6, 16), np.float64, [5, 4], True, "4,3,3"),         # hybrid: the d -> b transition in the middle
    ((23, 6, 16), np.float64, [2, 7], True, "4,4,3"),         # ... inside the start wavefront
    ((23, 6, 16), np.float32, [8, 2], True, "4,3,4"),         # ... inside the end wavefront
    ((19, 3, 5, 7), np.float32, 9, False, "3,2,5"),           # unaccelerated, scalar packs
    ((17, 3, 4, 8), np.float32, 6, True, "32,3,3"),           # one chunk taller than the cube
    ((33, 3, 4, 8), np.float32, 7, True, "1,3,2"),            # one-row chunks
    ((16, 3, 4, 8), np.float32, 5, True, "4,5,0"),            # no end wavefront: whole-cube download
    ((16, 3, 4, 8), np.float32, 4, True, "4,1,3"),
])
def test_pipelined_run_matches_the_oracle(oracle, monkeypatch, shape, dtype, its, fista, pipe):
    import cytvdn_amd as tv
    from cytvdn_amd import synth
    dt = np.dtype(dtype)
    nd = len(shape)
    x = synth.cube(shape, seed=29, dtype=dt) + dt.type(0.25)
    x0 = x.copy()
    mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
    fn = tv.denoise4D if nd == 4 else tv.denoise3D
    ref = oracle.denoise(x, mu, its, fista)
    monkeypatch.setenv("TVDN_PIPELINE", pipe)
    recon, bn, dl = fn(x, mu, its, FISTA=fista, quiet=True)
    assert bits_equal(x, x0)
    assert bits_equal(recon, ref["recon"])
    np.testing.assert_allclose(bn.astype(np.float64), ref["b_norm64"].astype(dt).astype(np.float64),
                               rtol=1e-6 if dt == np.float32 else 1e-9)
    want = (ref["delta64"].astype(dt) / ref["rnorm64"].astype(dt)).astype(np.float64)
    np.testing.assert_allclose(dl.astype(np.float64), want, rtol=1e-6 if dt == np.float32 else 1e-9)
    monkeypatch.setenv("TVDN_PIPELINE", "0")
    plain = fn(x, mu, its, FISTA=fista, quiet=True)
    assert bits_equal(recon, plain[0])
    # the same f64 sums, folded per partial launch instead of per sweep: the last bits may differ
    np.testing.assert_allclose(bn.astype(np.float64), plain[1].astype(np.float64), rtol=1e-6 if dt == np.float32 else 1e-12)


def test_the_plan_switches_itself_on_for_large_cubes_only():
    from cytvdn_amd import pipelined
    assert pipelined.plan(256, 50, 4 << 30) == (32, 8, 8)
    assert pipelined.plan(256, 6, 4 << 30) == (32, 3, 3)
    assert pipelined.plan(256, 3, 4 << 30) is None            # too few iterations to hide anything under
    assert pipelined.plan(16, 50, 4 << 30) is None
    assert pipelined.plan(256, 50, 64 << 20) is None          # small cubes: two transfers of milliseconds


def test_a_nonfinite_first_row_takes_the_plain_order(oracle, monkeypatch):
    """The wavefront's top face uses the Jia-Zhao zero, which upstream computes only while row 0 is finite
    (anisotropic.pyx:65-73): such a cube is not pipelined, and still equals the oracle."""
    import cytvdn_amd as tv
    from cytvdn_amd import synth
    dt = np.dtype(np.float32)
    x = synth.cube((24, 3, 4, 8), seed=5, dtype=dt) + dt.type(0.25)
    x[0, 1, 2, 3] = np.inf
    mu = np.array([1.0, 0.8, 0.5, 0.6], dt)
    monkeypatch.setenv("TVDN_PIPELINE", "4,3,3")
    got = tv.denoise4D(x, mu, 8, quiet=True)
    ref = oracle.denoise(x, mu, 8, True)
    assert np.isnan(ref["recon"][-1]).any() and bits_equal(got[0], ref["recon"])
