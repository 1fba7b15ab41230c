"""The one-pass (kernel-level) entry points at sizes where their marches, tiles and vector packs are all in play,
and on device-resident arrays: torch CUDA tensors and any other DLPack producer, f32 + f64,
3-D + 4-D, every entry point -- against the CPU oracle, bit for bit."""
import numpy as np
import pytest

from golden_util import bits_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tv():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import cytvdn_amd
    return cytvdn_amd


def _rand(rng, shape, dt, scale):
    return (rng.standard_normal(shape) * scale).astype(dt)


MID = [
    ((40, 16, 128, 128), np.float32),   # 256 tiles x 10 marches of 4 rows
    ((24, 16, 64, 128), np.float64),    # two doubles per thread
    ((48, 256, 512), np.float32),       # 3-D: axis A absent
    ((33, 5, 7, 36), np.float32),       # ragged: last march short, last tile partly empty
    ((19, 9, 10), np.float64),          # C not a multiple of the pack: scalar path
]


@pytest.mark.parametrize("shape,dtype", MID, ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else np.dtype(v).name)
def test_one_pass_kernels_on_device_tensors(tv, oracle, shape, dtype):
    """Every axis x BC 0/1/2 x plain/FISTA of accumulator_update, datacube_update for BC 0/2 and sum_square_error,
    called on torch CUDA tensors (updated in place in HBM, no host round trip)."""
    import torch
    dt = np.dtype(dtype)
    nd = len(shape)
    rng = np.random.default_rng(hash(shape) % (2 ** 31))
    a = _rand(rng, shape, dt, 3.0)
    ta = torch.from_numpy(a).cuda()
    clip, tk = dt.type(0.8), dt.type(0.41)
    sfx = f"{nd}D"
    # f32 data summed in f64 is exact to the last few bits whatever the order; for f64 data the oracle's serial running
    # sum over millions of terms and the tree sum here each carry ~1e-11 of rounding
    stol = 1e-12 if dt == np.float32 else 1e-9
    for ax in range(nd):
        for bc in (0, 1, 2):
            for fista in (False, True):
                b, d = _rand(rng, shape, dt, 0.7), _rand(rng, shape, dt, 0.7)
                tb, td = torch.from_numpy(b).cuda(), torch.from_numpy(d).cuda()
                if fista:
                    ret = getattr(tv, f"accumulator_update_{sfx}_FISTA")(ta, tb, td, tk, ax, clip, BC_mode=bc)
                    _, n64 = oracle.acc_update(a, b, d, tk, ax, clip, bc)
                    assert bits_equal(td.cpu().numpy(), d), (ax, bc)
                else:
                    ret = getattr(tv, f"accumulator_update_{sfx}")(ta, tb, ax, clip, BC_mode=bc)
                    _, n64 = oracle.acc_update(a, b, None, 0.0, ax, clip, bc)
                assert bits_equal(tb.cpu().numpy(), b), (ax, bc, fista)
                assert ret == pytest.approx(n64, rel=stol)
    assert bits_equal(ta.cpu().numpy(), a)     # the read-only role is untouched
    orig, bs = _rand(rng, shape, dt, 3.0), [_rand(rng, shape, dt, 0.7) for _ in range(nd)]
    lm = np.array([1 / 32, 1 / 40, 1 / 64, 1 / 33][:nd]).astype(dt)
    to, tbs = torch.from_numpy(orig).cuda(), [torch.from_numpy(x).cuda() for x in bs]
    for bc in (0, 2):
        recon = _rand(rng, shape, dt, 3.0)
        tr = torch.from_numpy(recon).cuda()
        ret = getattr(tv, f"datacube_update_{sfx}")(to, tr, *tbs, lm, BC_mode=bc)
        _, dl, rn = oracle.recon_update(orig, recon, bs, lm, bc)
        assert bits_equal(tr.cpu().numpy(), recon), bc
        assert ret == pytest.approx(float(dt.type(dl) / dt.type(rn)), rel=1e-6 if dt == np.float32 else 1e-9)
    r2 = _rand(rng, shape, dt, 3.0)
    got = getattr(tv, f"sum_square_error_{sfx}")(to, torch.from_numpy(r2).cuda())
    # squares are formed in the array dtype (utils.pyx:26), the oracle's yardstick squares in f64
    assert got == pytest.approx(oracle.sse(orig, r2)[1], rel=1e-6 if dt == np.float32 else 1e-9)


@pytest.mark.parametrize("ta", ["1", "2", "4"])
def test_recon_update_a_rows_per_thread(tv, oracle, monkeypatch, ta):
    """datacube_update_4D with 1, 2 and 4 A-rows per thread (the A-neighbour of all but the last is a register):
    extents of A that are and are not multiples of the tile, both wrap-sharing boundary conditions."""
    monkeypatch.setenv("TVDN_RECON_TA", ta)
    rng = np.random.default_rng(int(ta))
    for shape, dt in (((5, 8, 3, 16), np.dtype(np.float32)), ((4, 12, 5, 8), np.dtype(np.float64)),
                      ((3, 6, 4, 32), np.dtype(np.float32)), ((6, 2, 3, 8), np.dtype(np.float32)), ((3, 7, 2, 8), np.dtype(np.float32))):
        orig, recon = _rand(rng, shape, dt, 3.0), _rand(rng, shape, dt, 3.0)
        bs = [_rand(rng, shape, dt, 0.7) for _ in range(4)]
        lm = np.array([1 / 32, 1 / 40, 1 / 64, 1 / 33]).astype(dt)
        for bc in (0, 2):
            r1, r2 = recon.copy(), recon.copy()
            tv.datacube_update_4D(orig, r1, *bs, lm, BC_mode=bc)
            oracle.recon_update(orig, r2, bs, lm, bc)
            assert bits_equal(r1, r2), (shape, bc)


@pytest.mark.parametrize("chunk", ["3", "8"])
def test_one_pass_march_seams(tv, oracle, monkeypatch, chunk):
    """Marches of a forced length (odd, so the last one is short; 8, the production length) on a small array: the
    register-carried axis-0 neighbour across rows and across march seams."""
    monkeypatch.setenv("TVDN_PASS_CHUNK", chunk)
    rng = np.random.default_rng(int(chunk))
    for shape, dt in (((20, 3, 4, 16), np.dtype(np.float32)), ((17, 6, 8), np.dtype(np.float64)), ((10, 2, 3, 5), np.dtype(np.float32))):
        nd = len(shape)
        a = _rand(rng, shape, dt, 3.0)
        for bc in (0, 1, 2):
            b, d = _rand(rng, shape, dt, 0.7), _rand(rng, shape, dt, 0.7)
            b2, d2 = b.copy(), d.copy()
            getattr(tv, f"accumulator_update_{nd}D_FISTA")(a, b, d, dt.type(0.3), 0, dt.type(0.9), BC_mode=bc)
            oracle.acc_update(a, b2, d2, dt.type(0.3), 0, dt.type(0.9), bc)
            assert bits_equal(b, b2) and bits_equal(d, d2), (shape, bc)
        orig, recon = _rand(rng, shape, dt, 3.0), _rand(rng, shape, dt, 3.0)
        bs = [_rand(rng, shape, dt, 0.7) for _ in range(nd)]
        lm = np.array([1 / 32, 1 / 40, 1 / 64, 1 / 33][:nd]).astype(dt)
        r2 = recon.copy()
        getattr(tv, f"datacube_update_{nd}D")(orig, recon, *bs, lm)
        oracle.recon_update(orig, r2, bs, lm, 2)
        assert bits_equal(recon, r2), shape


class _DLPackOnly:
    """A device array that offers nothing but the DLPack protocol (what CuPy / JAX arrays look like to us)."""

    def __init__(self, t):
        self._t = t

    def __dlpack__(self, stream=None, **kw):
        return self._t.__dlpack__(stream=stream) if stream is not None else self._t.__dlpack__()

    def __dlpack_device__(self):
        return self._t.__dlpack_device__()


def test_dlpack_producers_are_updated_in_place(tv, oracle):
    import torch
    rng = np.random.default_rng(9)
    shape, dt = (6, 5, 8, 16), np.dtype(np.float32)
    a, b, d = (_rand(rng, shape, dt, s) for s in (3.0, 0.7, 0.7))
    ta, tb, td = (torch.from_numpy(v.copy()).cuda() for v in (a, b, d))
    ret = tv.accumulator_update_4D_FISTA(_DLPackOnly(ta), _DLPackOnly(tb), _DLPackOnly(td), 0.3, 1, np.float32(0.8))
    _, n64 = oracle.acc_update(a, b, d, np.float32(0.3), 1, np.float32(0.8), 2)
    assert bits_equal(tb.cpu().numpy(), b) and bits_equal(td.cpu().numpy(), d)      # the producer's memory changed
    assert ret == pytest.approx(n64, rel=1e-12)
    orig, recon = _rand(rng, shape, dt, 3.0), _rand(rng, shape, dt, 3.0)
    bs = [_rand(rng, shape, dt, 0.7) for _ in range(4)]
    lm = np.array([1 / 32, 1 / 40, 1 / 64, 1 / 33], dt)
    tr = torch.from_numpy(recon.copy()).cuda()
    tv.datacube_update_4D(_DLPackOnly(torch.from_numpy(orig).cuda()), _DLPackOnly(tr),
                          *[_DLPackOnly(torch.from_numpy(x).cuda()) for x in bs], lm)
    oracle.recon_update(orig, recon, bs, lm, 2)
    assert bits_equal(tr.cpu().numpy(), recon)
    got = tv.sum_square_error_4D(_DLPackOnly(tr), _DLPackOnly(ta))
    assert got == pytest.approx(oracle.sse(recon, a)[1], rel=1e-6)


def test_reduction_scratch_grows_and_folds_in_two_stages(tv, oracle):
    """More workgroups than the first-stage threshold (4096 partial rows) and than the initial scratch (2^18 rows):
    the sums must still match the oracle's f64 yardsticks to 1e-12."""
    import torch
    rng = np.random.default_rng(3)
    shape, dt = (300000 * 8, 1, 128), np.dtype(np.float32)      # 3-D with a unit middle axis: 32 units -> 1 tile per row
    a = _rand(rng, shape, dt, 1.0)
    b = _rand(rng, shape, dt, 1.0)
    got = tv.sum_square_error_3D(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())      # 3.7e4 workgroups
    want = float(np.sum(((a - b) * (a - b)).astype(np.float64)))       # squares in f32 as upstream, summed in f64
    assert got == pytest.approx(want, rel=1e-12)
    tb = torch.from_numpy(b).cuda()
    b2 = b.copy()
    ret = tv.accumulator_update_3D(torch.from_numpy(a).cuda(), tb, 0, np.float32(0.5))           # 3e5 marches of 8 rows
    _, n64 = oracle.acc_update(a, b2, None, 0.0, 0, np.float32(0.5), 2)
    assert bits_equal(tb.cpu().numpy(), b2)
    assert ret == pytest.approx(n64, rel=1e-12)
