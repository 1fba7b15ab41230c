"""Inf / NaN in the FIRST row of the cube.  Upstream's Jia-Zhao accumulator of row 0 is clip((a[0] - a[0]) + b[0])
(anisotropic.pyx:65-73): NaN where a[0] is not finite, and the periodic wrap of the reconstruction update
(utils.pyx:98-101) carries that NaN into the LAST row.  The single-slab sweep wraps for real; every path that closes
the wrap across slabs or across streamed chunks must reproduce it too (TVDN_EDGE_WRAP): logical slabs on one GPU, the
slab API over two processes, tvdn_run's device list, streamed runs."""
import os
import socket
import tempfile

import numpy as np
import pytest

from golden_util import bits_equal

pytestmark = pytest.mark.gpu


def _cube(shape, dt, seed=3):
    from cytvdn_amd import synth
    x = synth.cube(shape, seed=seed, dtype=dt) + dt.type(0.25)
    x[(0, 1) + (2,) * (len(shape) - 2)] = np.inf        # first row: the case the wrap has to carry
    x[(0, 0) + (1,) * (len(shape) - 2)] = np.nan
    x[(shape[0] // 2, 1) + (0,) * (len(shape) - 2)] = -np.inf   # elsewhere: propagates on every path anyway
    return x


def _mu(nd, dt):
    return np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)


@pytest.mark.parametrize("shape,dtype,world,its", [
    ((12, 3, 4, 8), np.float32, 3, [4, 2]),
    ((9, 5, 16), np.float64, 2, 5),
    ((7, 2, 3, 4), np.float32, 7, 4),          # one row per slab
])
def test_logical_slabs_with_wrap_row(oracle, shape, dtype, world, its):
    import torch
    from cytvdn_amd.engine import HipBackend, LocalSlabs, SlabLayout
    dt = np.dtype(dtype)
    nd = len(shape)
    x = _cube(shape, dt)
    mu = _mu(nd, dt)
    lam = mu / dt.type(32.0 if nd == 4 else 16.0)
    n_f, n_p = its if isinstance(its, list) else (its, 0)
    ref = oracle.denoise(x, mu, its, True)
    assert np.isnan(ref["recon"][-1]).any()                       # the wrap did carry something into the last row
    for split in (False, True):
        bes = []
        for r in range(world):
            lay = SlabLayout(shape, r, world, 2, wrap_row=True)
            be = HipBackend(lay, dt, True, device=0, max_iters=n_f + n_p)
            be.set_params(1.0 / lam, (lam / mu).astype(dt))
            be.set_input(x[lay.local_rows_global()])
            bes.append(be)
        grp = LocalSlabs(bes, split_sweeps=split)
        grp.run(n_f, n_p)
        assert bits_equal(grp.gather_recon().cpu().numpy(), ref["recon"]), split
    # without the wrap row the last row stays finite where upstream has NaN: the constant is exact for finite data only
    bes = []
    for r in range(world):
        lay = SlabLayout(shape, r, world, 2)
        be = HipBackend(lay, dt, True, device=0, max_iters=n_f + n_p)
        be.set_params(1.0 / lam, (lam / mu).astype(dt))
        be.set_input(x[lay.local_rows_global()])
        bes.append(be)
    grp = LocalSlabs(bes)
    grp.run(n_f, n_p)
    assert not bits_equal(grp.gather_recon().cpu().numpy(), ref["recon"])
    torch.cuda.empty_cache()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, shape, dtype_name, its, outdir, staged=None, stop=None):
    import torch
    import torch.distributed as dist
    from cytvdn_amd.distributed import denoise_slabs, slab_rows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        dt = np.dtype(dtype_name)
        x = _cube(shape, dt)
        g0, g1 = slab_rows(shape, rank, world)
        try:
            own, bn, dl = denoise_slabs(x[g0:g1], shape, _mu(len(shape), dt), its, FISTA=True, staged=staged,
                                        stopping_relative_change=stop)
            np.savez(os.path.join(outdir, f"r{rank}.npz"), own=own)
        except NotImplementedError as e:
            np.savez(os.path.join(outdir, f"r{rank}.npz"), refused=str(e))
    finally:
        dist.destroy_process_group()


def test_denoise_slabs_switches_the_wrap_row_on_by_itself(oracle):
    import torch.multiprocessing as mp
    shape, dtype, world, its = (11, 3, 4, 8), "float32", 3, 5
    with tempfile.TemporaryDirectory() as tmp:
        mp.start_processes(_worker, args=(world, _free_port(), shape, dtype, its, tmp), nprocs=world, join=True,
                           start_method="spawn")
        recon = np.concatenate([np.load(os.path.join(tmp, f"r{r}.npz"))["own"] for r in range(world)], axis=0)
    dt = np.dtype(dtype)
    ref = oracle.denoise(_cube(shape, dt), _mu(4, dt), its, True)
    assert np.isnan(ref["recon"][-1]).any() and bits_equal(recon, ref["recon"])


@pytest.mark.parametrize("world,shape,dtype,its,staged", [
    (2, (20, 3, 4, 8), "float32", 9, (4, 3)),            # three passes of three levels, several chunks per rank
    (3, (19, 6, 16), "float64", [5, 4], (3, 4)),         # hybrid, the d -> b transition inside a pass, uneven slabs
    (2, (24, 2, 5, 7), "float32", 7, (2, 8)),            # more levels than chunk rows: row 0 of level j is final only in chunk j / R
    (3, (9, 3, 4, 8), "float32", 7, (2, 3)),             # the MIDDLE rank's 3-row halo reaches the cube's top face: it needs row 0 too
    (3, (12, 6, 16), "float64", [4, 3], (3, 4)),
])
def test_staged_slabs_carry_the_wrap_row_across_ranks(oracle, world, shape, dtype, its, staged):
    """Slabs in pinned host memory, wavefront schedule inside each (BASELINE config 5 in structure): rank 0 broadcasts row 0
    of every level of a pass; the ranks whose sweeps reach the cube's top face close the Jia-Zhao wrap with it (TVDN_EDGE_WRAP) -- the NaN upstream
    carries from a non-finite first row into the LAST row (anisotropic.pyx:65-73, utils.pyx:98-101) must appear."""
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as tmp:
        mp.start_processes(_worker, args=(world, _free_port(), shape, dtype, its, tmp, staged), nprocs=world, join=True,
                           start_method="spawn")
        recon = np.concatenate([np.load(os.path.join(tmp, f"r{r}.npz"))["own"] for r in range(world)], axis=0)
    dt = np.dtype(dtype)
    ref = oracle.denoise(_cube(shape, dt), _mu(len(shape), dt), its, True)
    assert np.isnan(ref["recon"][-1]).any() and bits_equal(recon, ref["recon"])


def test_staged_slabs_with_a_stopping_rule_carry_the_wrap_row_too(oracle):
    """With a stopping rule every pass is one iteration deep and the relay of row 0 happens once per iteration: the oracle's
    bits, NaNs in the last row included (round 3's trapezoid engine across ranks refused this cube)."""
    import torch.multiprocessing as mp
    shape, world, its = (12, 3, 4, 8), 2, 6
    with tempfile.TemporaryDirectory() as tmp:
        mp.start_processes(_worker, args=(world, _free_port(), shape, "float32", its, tmp, (3, 2), 1e-9), nprocs=world,
                           join=True, start_method="spawn")
        recon = np.concatenate([np.load(os.path.join(tmp, f"r{r}.npz"))["own"] for r in range(world)], axis=0)
    dt = np.dtype(np.float32)
    ref = oracle.denoise(_cube(shape, dt), _mu(4, dt), its, True, stopping_relative_change=1e-9)
    assert np.isnan(ref["recon"][-1]).any() and bits_equal(recon, ref["recon"])


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0, 0]])
def test_tvdn_run_device_list_nonfinite_first_row(oracle, devices):
    import ctypes as C
    from cytvdn_amd import _lib
    shape, dt, its = (13, 3, 4, 8), np.dtype(np.float32), 5
    x = _cube(shape, dt)
    mu = _mu(4, dt)
    lam = mu / dt.type(32.0)
    a = _lib.RunArgs(dtype=0, ndim=4, bc_mode=2, device=0, n_fista=its, n_plain=0, n_devices=len(devices))
    for i, d in enumerate(devices):
        a.devices[i] = d
    for i, s in enumerate(shape):
        a.shape[i] = s
    for q in range(4):
        a.clip[q] = float((1.0 / lam)[q])
        a.lambda_mu[q] = float((lam / mu).astype(dt)[q])
    recon, sums = np.empty_like(x), np.zeros((its, 3))
    a.data, a.recon_out, a.sums_out = x.ctypes.data, recon.ctypes.data, sums.ctypes.data
    _lib.check(_lib.lib().tvdn_run(C.byref(a)))
    ref = oracle.denoise(x, mu, its, True)
    assert np.isnan(ref["recon"][-1]).any() and bits_equal(recon, ref["recon"])


@pytest.mark.parametrize("env,val", [("TVDN_WAVEFRONT", "3,2"), ("TVDN_WAVEFRONT", "16,8"), ("TVDN_STAGED", "4,2"),
                                     ("TVDN_STAGED", "5,1"), ("TVDN_HBM_LIMIT", "2M")])
def test_streamed_runs_nonfinite_first_row(oracle, monkeypatch, env, val):
    import cytvdn_amd as tv
    for k in ("TVDN_WAVEFRONT", "TVDN_STAGED", "TVDN_HBM_LIMIT"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv(env, val)
    for shape, dtype in (((14, 3, 4, 8), np.float32), ((11, 6, 16), np.float64)):
        dt = np.dtype(dtype)
        nd = len(shape)
        x = _cube(shape, dt)
        mu = np.array([1.0, 1.0, 0.5, 0.5][:nd] if nd == 4 else [1.0, 1.0, 0.5], dt)
        fn = tv.denoise4D if nd == 4 else tv.denoise3D
        got = fn(x, mu, [4, 3], FISTA=True, quiet=True)
        ref = oracle.denoise(x, mu, [4, 3], True)
        assert np.isnan(ref["recon"][-1]).any()
        assert bits_equal(got[0], ref["recon"]), (env, val, shape)
