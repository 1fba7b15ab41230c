"""Per-slab semantics of the fused HIP sweep: N logical slabs of one cube on ONE MI355X, halo rows
moved by device copies, must reproduce the single-slab result and the oracle bit for bit."""
import numpy as np
import pytest

from golden_util import bits_equal

pytestmark = pytest.mark.gpu

CASES = [
    # world, shape, dtype, bc, n_fista, n_plain
    (2, (7, 3, 4, 8), "float32", 2, 5, 0),
    (3, (9, 3, 4, 8), "float32", 2, 4, 2),
    (4, (9, 6, 16), "float64", 2, 0, 5),
    (2, (5, 3, 4, 4), "float32", 0, 4, 2),       # periodic ring of two
    (3, (8, 2, 3, 6), "float64", 0, 3, 2),       # periodic ring, scalar (VEC=1 for f64? 6 % 2 == 0 -> vector) path
    (8, (16, 2, 5, 7), "float32", 2, 6, 0),      # 8 slabs of 2 rows, scalar path
    (5, (5, 4, 8), "float32", 2, 3, 1),          # one row per slab
    (2, (40, 4, 4, 8), "float32", 2, 3, 0),      # several marching chunks per slab
    (1, (11, 3, 4, 8), "float32", 0, 3, 2),      # single slab, periodic, swept as edge rows + interior
    (1, (11, 5, 12), "float64", 2, 3, 2),
]


@pytest.mark.parametrize("world,shape,dtype,bc,n_f,n_p", CASES,
                         ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else str(v))
@pytest.mark.parametrize("split", [False, True], ids=["whole", "edge-rows-first"])
def test_logical_slabs_match_oracle(oracle, world, shape, dtype, bc, n_f, n_p, split):
    import torch
    from cytvdn_amd import synth
    from cytvdn_amd.engine import HipBackend, LocalSlabs, SlabLayout
    assert torch.cuda.is_available()
    dt = np.dtype(dtype)
    nd = len(shape)
    x = synth.cube(shape, seed=91, dtype=dt) + dt.type(0.25)
    mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
    lam = mu / dt.type(32.0 if nd == 4 else 16.0)
    bes = []
    for r in range(world):
        lay = SlabLayout(tuple(shape), r, world, bc)
        be = HipBackend(lay, dt, n_f > 0, device=0, max_iters=n_f + n_p)
        be.set_params(1.0 / lam, (lam / mu).astype(dt))
        be.set_input(x[lay.local_rows_global()])
        bes.append(be)
    grp = LocalSlabs(bes, split_sweeps=split)
    grp.run(n_f, n_p)
    recon = grp.gather_recon().cpu().numpy()
    its = [n_f, n_p] if (n_f and n_p) else (n_f or n_p)
    ref = oracle.denoise(x, mu, its, n_f > 0, BC_mode=bc)
    assert bits_equal(recon, ref["recon"])
    sums = grp.global_sums().cpu().numpy()
    np.testing.assert_allclose(sums[:, 0], ref["b_norm64"], rtol=1e-12)
    np.testing.assert_allclose(sums[:, 1], ref["delta64"], rtol=1e-12)
    np.testing.assert_allclose(sums[:, 2], ref["rnorm64"], rtol=1e-12)


def test_synth_device_matches_host():
    """tvdn_synth_fill restates cytvdn_amd/synth.py bit for bit, for whole cubes and row ranges."""
    import torch
    from cytvdn_amd import _lib, synth
    for shape, dt in (((6, 5, 16, 20), np.float32), ((7, 9, 33), np.float64), ((3, 4, 64, 64), np.float32)):
        nd = len(shape)
        seed = 1234567
        host = synth.cube(shape, seed=seed, dtype=dt)
        tdt = torch.float32 if dt == np.float32 else torch.float64
        out = torch.empty(shape, dtype=tdt, device="cuda")
        _lib.ctx(0)
        _lib.check(_lib.lib().tvdn_synth_fill(_lib.dtype_code(dt), nd, _lib.shape_arr(shape), seed, 0, shape[0],
                                              out.data_ptr(), _lib.current_stream(0)))
        assert bits_equal(out.cpu().numpy(), host)
        part = torch.empty((2,) + tuple(shape[1:]), dtype=tdt, device="cuda")
        _lib.check(_lib.lib().tvdn_synth_fill(_lib.dtype_code(dt), nd, _lib.shape_arr(shape), seed, 1, 2,
                                              part.data_ptr(), _lib.current_stream(0)))
        assert bits_equal(part.cpu().numpy(), host[1:3])


@pytest.mark.parametrize("shape,dtype,its,fista,stop,with_ref,bc", [
    ((17, 4, 8, 16), np.float32, [4, 3], True, None, True, 2),
    ((14, 6, 16), np.float64, 6, True, None, False, 0),
    ((19, 3, 7, 9), np.float32, 40, False, 0.02, False, 2),
])
def test_denoise_with_a_device_list(oracle, shape, dtype, its, fista, stop, with_ref, bc):
    """`denoise3D/4D(..., device=[...])`: several slabs inside one process (here all on device 0) through tvdn_run:
    the reference's arguments and return tuple, the oracle's bits."""
    import cytvdn_amd as tv
    from cytvdn_amd import synth
    dt = np.dtype(dtype)
    nd = len(shape)
    x = synth.cube(shape, seed=19, dtype=dt) + dt.type(0.25)
    refd = synth.cube(shape, seed=19, dtype=dt, kind="mean") if with_ref else None
    mu = np.array([1.0, 1.0, 0.5, 0.5][:nd] if nd == 4 else [1.0, 1.0, 0.5], dt)
    fn = tv.denoise4D if nd == 4 else tv.denoise3D
    one = fn(x, mu, its, FISTA=fista, stopping_relative_change=stop, reference_data=refd, BC_mode=bc, quiet=True)
    for devs in ([0, 0], [0, 0, 0, 0, 0]):
        got = fn(x, mu, its, FISTA=fista, stopping_relative_change=stop, reference_data=refd, BC_mode=bc, quiet=True,
                 device=devs)
        assert len(got) == len(one) and bits_equal(got[0], one[0])
        tol = 1e-6 if dt == np.float32 else 1e-12
        for u, v in zip(got[1:], one[1:]):
            assert u.dtype == v.dtype and np.array_equal(u == 0, v == 0)
            np.testing.assert_allclose(u, v, rtol=tol)
    ref = oracle.denoise(x, mu, its, fista, stopping_relative_change=stop, reference_data=refd, BC_mode=bc)
    assert bits_equal(one[0], ref["recon"])


def test_entry_points_leave_the_current_device_alone(oracle):
    """tvdn_run / tvdn_copy_* / tvdn_ctx_create select their device internally and put the caller's back (csrc/tvdn_common.hpp
    DeviceRestore): after a run on another GPU torch's next allocation still lands where it did.  Needs two GPUs to mean
    anything; on one it checks the call sequence only."""
    import torch
    import cytvdn_amd as tv
    from cytvdn_amd import synth
    dt = np.dtype(np.float32)
    x = synth.cube((12, 3, 4, 8), seed=3, dtype=dt) + dt.type(0.25)
    mu = np.array([1.0, 0.8, 0.5, 0.6], dt)
    other = 1 if torch.cuda.device_count() > 1 else 0
    torch.cuda.set_device(0)
    before = torch.cuda.current_device()
    got = tv.denoise4D(x, mu, 4, quiet=True, device=other)
    assert torch.cuda.current_device() == before == 0
    assert torch.empty(1, device="cuda").device.index == 0
    got2 = tv.denoise4D(x, mu, 4, quiet=True, device=[0, other])
    assert torch.cuda.current_device() == 0
    ref = oracle.denoise(x, mu, 4, True)["recon"]
    assert bits_equal(got[0], ref) and bits_equal(got2[0], ref)


@pytest.mark.parametrize("shape,dtype,its,with_ref,bc", [((23, 4, 8, 16), np.float32, [4, 3], True, 2), ((18, 6, 16), np.float64, 6, False, 0)])
def test_slabs_go_up_and_come_home_side_by_side(oracle, monkeypatch, shape, dtype, its, with_ref, bc):
    """Slabs on several devices are uploaded and downloaded by a host thread each (csrc/tvdn_run.hip each_slab; the staging lanes
    are per device).  On one GPU the threads are forced (TVDN_SLAB_IO_THREADS=1: the lanes of the one device take turns): same
    bits as the run that moves the slabs in turn, and the oracle's."""
    import cytvdn_amd as tv
    from cytvdn_amd import synth
    dt = np.dtype(dtype)
    nd = len(shape)
    x = synth.cube(shape, seed=23, dtype=dt) + dt.type(0.25)
    refd = synth.cube(shape, seed=23, dtype=dt, kind="mean") if with_ref else None
    mu = np.array([1.0, 1.0, 0.5, 0.5][:nd] if nd == 4 else [1.0, 1.0, 0.5], dt)
    fn = tv.denoise4D if nd == 4 else tv.denoise3D
    monkeypatch.setenv("TVDN_SLAB_IO_THREADS", "0")
    in_turn = fn(x, mu, its, FISTA=True, reference_data=refd, BC_mode=bc, quiet=True, device=[0, 0, 0, 0])
    monkeypatch.setenv("TVDN_SLAB_IO_THREADS", "1")
    for _ in range(3):
        got = fn(x, mu, its, FISTA=True, reference_data=refd, BC_mode=bc, quiet=True, device=[0, 0, 0, 0])
        assert len(got) == len(in_turn)
        for u, v in zip(got, in_turn):
            assert bits_equal(u, v)
    ref = oracle.denoise(x, mu, its, True, reference_data=refd, BC_mode=bc)
    assert bits_equal(got[0], ref["recon"])
