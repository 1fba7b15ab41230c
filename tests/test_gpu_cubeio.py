"""denoise_file: a cube streamed from a memory-mapped file through the in-core and the out-of-core
engines into an output file must equal denoise3D/4D on the same cube held in memory, bit for bit, and the oracle."""
import numpy as np
import pytest

from golden_util import bits_equal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,stored,limit,stop", [
    ((21, 4, 8, 16), np.float32, None, None),      # in core
    ((21, 4, 8, 16), np.uint16, None, None),       # counts as stored by a detector: converted per block
    ((40, 8, 32, 64), np.float32, "24M", None),    # state exceeds the HBM the planner may use: wavefront engine
    ((40, 8, 32, 64), np.float32, "24M", 0.05),    # ... with a stopping rule: trapezoid blocks, k = 1
    ((30, 12, 40), np.float64, None, None),        # 3-D
], ids=["in-core-f32", "in-core-from-u16", "wavefront", "trapezoid-stop", "3d-f64"])
def test_file_to_file_equals_in_memory(oracle, tmp_path, monkeypatch, shape, stored, limit, stop):
    import cytvdn_amd as tv
    from cytvdn_amd import synth
    monkeypatch.delenv("TVDN_WAVEFRONT", raising=False)
    monkeypatch.delenv("TVDN_STAGED", raising=False)
    nd = len(shape)
    dt = np.dtype(np.float64 if stored == np.float64 else np.float32)
    x = synth.cube(shape, seed=23, dtype=dt)
    if stored == np.uint16:
        x = np.round(x * 7).astype(np.uint16)
    np.save(tmp_path / "in.npy", x)
    mu = np.array([1.0, 1.0, 0.5, 0.5][:nd] if nd == 4 else [1.0, 1.0, 0.5], dt)
    its = [4, 3] if stop is None else 30
    if limit:
        monkeypatch.setenv("TVDN_HBM_LIMIT", limit)
    bn, dl = tv.denoise_file(str(tmp_path / "in.npy"), str(tmp_path / "out.npy"), mu, its, FISTA=True,
                             stopping_relative_change=stop, dtype=dt)
    got = np.load(tmp_path / "out.npy")
    monkeypatch.delenv("TVDN_HBM_LIMIT", raising=False)
    xm = x.astype(dt)
    fn = tv.denoise4D if nd == 4 else tv.denoise3D
    want = fn(xm, mu, its, FISTA=True, stopping_relative_change=stop, quiet=True)
    assert got.dtype == dt and bits_equal(got, want[0])
    assert bits_equal(bn, want[1]) and bits_equal(dl, want[2])
    ref = oracle.denoise(xm, mu, its, True, stopping_relative_change=stop)
    assert bits_equal(got, ref["recon"])
