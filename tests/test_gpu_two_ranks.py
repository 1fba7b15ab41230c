"""Two PROCESSES, one MI355X: the full multi-rank code path (SlabLayout + HipBackend + SlabRunner +
torch.distributed) with the halo rows staged through the host over gloo.  RCCL needs one GPU per rank,
which a 1-GPU box cannot offer; everything except the device-to-device transport itself runs here."""
import os
import socket
import tempfile

import numpy as np
import pytest

from golden_util import bits_equal

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, shape, dtype_name, bc, n_f, n_p, outdir):
    import torch
    import torch.distributed as dist
    from cytvdn_amd import synth
    from cytvdn_amd.engine import HipBackend, SlabLayout, SlabRunner
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        dt = np.dtype(dtype_name)
        nd = len(shape)
        lay = SlabLayout(tuple(shape), rank, world, bc)
        be = HipBackend(lay, dt, n_f > 0, device=0, max_iters=n_f + n_p)
        mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
        lam = mu / dt.type(32.0 if nd == 4 else 16.0)
        be.set_params(1.0 / lam, (lam / mu).astype(dt))
        full = synth.cube(shape, seed=91, dtype=dt) + dt.type(0.25)
        be.set_input(full[lay.local_rows_global()])
        run = SlabRunner(be)
        run.run(n_f, n_p)
        own = be.recon_tensor()[lay.row_lo:lay.row_hi].cpu().numpy()
        sums = run.global_sums().cpu().numpy()
        np.savez(os.path.join(outdir, f"r{rank}.npz"), own=own, sums=sums)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,dtype,bc,n_f,n_p", [
    (2, (9, 3, 4, 8), "float32", 2, 5, 2),
    (2, (8, 6, 16), "float64", 0, 4, 0),
    (3, (10, 2, 4, 8), "float32", 2, 4, 0),
], ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else str(v))
def test_two_rank_processes_on_one_gpu(oracle, world, shape, dtype, bc, n_f, n_p):
    import torch.multiprocessing as mp
    from cytvdn_amd import synth
    dt = np.dtype(dtype)
    nd = len(shape)
    with tempfile.TemporaryDirectory() as tmp:
        mp.start_processes(_worker, args=(world, _free_port(), shape, dtype, bc, n_f, n_p, tmp), nprocs=world,
                           join=True, start_method="spawn")
        parts = [np.load(os.path.join(tmp, f"r{r}.npz")) for r in range(world)]
    recon = np.concatenate([p["own"] for p in parts], axis=0)
    x = synth.cube(shape, seed=91, dtype=dt) + dt.type(0.25)
    mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
    its = [n_f, n_p] if (n_f and n_p) else (n_f or n_p)
    ref = oracle.denoise(x, mu, its, n_f > 0, BC_mode=bc)
    assert bits_equal(recon, ref["recon"])
    np.testing.assert_allclose(parts[0]["sums"][:, 0], ref["b_norm64"], rtol=1e-12)
    np.testing.assert_allclose(parts[0]["sums"][:, 1], ref["delta64"], rtol=1e-12)


def _worker_api(rank, world, port, shape, dtype_name, its, fista, stop, outdir, staged=None):
    import torch
    import torch.distributed as dist
    from cytvdn_amd import synth
    from cytvdn_amd.distributed import denoise_slabs, slab_rows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        dt = np.dtype(dtype_name)
        nd = len(shape)
        full = synth.cube(shape, seed=14, dtype=dt)
        g0, g1 = slab_rows(shape, rank, world)
        mu = np.array([1.0, 1.0, 0.5, 0.5][:nd] if nd == 4 else [1.0, 1.0, 0.5], dt)
        own, bn, dl = denoise_slabs(full[g0:g1], shape, mu, its, FISTA=fista, stopping_relative_change=stop, device=0,
                                    staged=staged)
        np.savez(os.path.join(outdir, f"r{rank}.npz"), own=own, bn=bn, dl=dl)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,dtype,its,fista,stop", [
    (2, (6, 5, 8, 12), "float32", 40, False, 0.02),
    (2, (7, 6, 16), "float64", [30, 6], True, 0.03),
], ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else str(v))
def test_denoise_slabs_on_gpu_matches_single_process(oracle, world, shape, dtype, its, fista, stop):
    """The user-facing slab API (global traces, global stopping criterion) through the HIP sweep."""
    import torch.multiprocessing as mp
    from cytvdn_amd import synth
    dt = np.dtype(dtype)
    nd = len(shape)
    with tempfile.TemporaryDirectory() as tmp:
        mp.start_processes(_worker_api, args=(world, _free_port(), shape, dtype, its, fista, stop, tmp), nprocs=world,
                           join=True, start_method="spawn")
        parts = [np.load(os.path.join(tmp, f"r{r}.npz")) for r in range(world)]
    x = synth.cube(shape, seed=14, dtype=dt)
    mu = np.array([1.0, 1.0, 0.5, 0.5][:nd] if nd == 4 else [1.0, 1.0, 0.5], dt)
    ref = oracle.denoise(x, mu, its, fista, stopping_relative_change=stop)
    assert bits_equal(np.concatenate([p["own"] for p in parts]), ref["recon"])
    for p in parts:
        assert np.array_equal(p["dl"] == 0, ref["delta_recon"] == 0)
        np.testing.assert_allclose(p["dl"], ref["delta_recon"], rtol=1e-4 if dt == np.float32 else 1e-12)


@pytest.mark.parametrize("world,shape,dtype,its,fista,stop,staged", [
    (2, (20, 3, 4, 8), "float32", 9, True, None, (4, 3, "trapezoid")),   # several blocks per rank, 3 halo rows between ranks
    (3, (19, 6, 16), "float64", [5, 4], True, None, (3, 4, "trapezoid")), # hybrid schedule, uneven slabs
    (2, (20, 3, 4, 8), "float32", 9, True, None, (4, 3)),        # wavefront inside each slab, trapezoid at the slab faces
    (3, (19, 6, 16), "float64", [5, 4], True, None, (3, 4)),
    (2, (24, 2, 5, 7), "float32", 14, True, None, (2, 8)),       # more levels than chunk rows, scalar path
    (4, (21, 4, 8), "float32", 7, False, None, (5, 5)),
    (2, (12, 5, 8, 12), "float32", [30, 6], True, 0.03, (5, 8)),  # global stopping rule (forces k = 1)
], ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else str(v))
def test_staged_slabs_match_single_process(oracle, world, shape, dtype, its, fista, stop, staged):
    """BASELINE config 5 in miniature: every rank keeps its slab in host memory and streams it through the
    GPU with temporal blocking; ranks swap k rows of state per pass.  Bit-identical to one process."""
    import torch.multiprocessing as mp
    from cytvdn_amd import synth
    dt = np.dtype(dtype)
    nd = len(shape)
    with tempfile.TemporaryDirectory() as tmp:
        mp.start_processes(_worker_api, args=(world, _free_port(), shape, dtype, its, fista, stop, tmp, staged),
                           nprocs=world, join=True, start_method="spawn")
        parts = [np.load(os.path.join(tmp, f"r{r}.npz")) for r in range(world)]
    x = synth.cube(shape, seed=14, dtype=dt)
    mu = np.array([1.0, 1.0, 0.5, 0.5][:nd] if nd == 4 else [1.0, 1.0, 0.5], dt)
    ref = oracle.denoise(x, mu, its, fista, stopping_relative_change=stop)
    assert bits_equal(np.concatenate([p["own"] for p in parts]), ref["recon"])
    for p in parts:
        assert np.array_equal(p["dl"] == 0, ref["delta_recon"] == 0)
        np.testing.assert_allclose(p["dl"], ref["delta_recon"], rtol=1e-4 if dt == np.float32 else 1e-12)
        np.testing.assert_allclose(p["bn"], ref["b_norm"], rtol=1e-4 if dt == np.float32 else 1e-12)


@pytest.mark.parametrize("world,shape,dtype,its,fista,stop,hbm", [
    (2, (40, 8, 32, 64), "float32", 9, True, None, "12M"),      # a slab's state (22 MB) exceeds what a rank may count on: streamed
    (3, (45, 8, 32, 64), "float32", [5, 3], True, None, "12M"),  # ... three ranks, hybrid schedule
    (2, (40, 8, 32, 64), "float32", 30, True, 0.05, "12M"),      # ... with a stopping rule: one iteration per pass
    (2, (40, 8, 32, 64), "float32", 9, True, None, "1G"),        # room to spare: the slabs stay resident, halo rows over the wire
], ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else str(v))
def test_staged_auto_lets_the_planner_decide(oracle, monkeypatch, world, shape, dtype, its, fista, stop, hbm):
    """denoise_slabs(staged="auto"): rank 0 plans for everybody (planner.plan_run with the HBM each rank may count on and the
    host memory the ranks share) -- resident slabs when they fit, else every rank streams its slab with the planned chunk
    height, depth and resident rows.  The oracle's bits either way."""
    import torch.multiprocessing as mp
    from cytvdn_amd import synth
    monkeypatch.setenv("TVDN_HBM_LIMIT", hbm)                   # (the spawned ranks inherit it)
    dt = np.dtype(dtype)
    nd = len(shape)
    with tempfile.TemporaryDirectory() as tmp:
        mp.start_processes(_worker_api, args=(world, _free_port(), shape, dtype, its, fista, stop, tmp, "auto"),
                           nprocs=world, join=True, start_method="spawn")
        parts = [np.load(os.path.join(tmp, f"r{r}.npz")) for r in range(world)]
    x = synth.cube(shape, seed=14, dtype=dt)
    mu = np.array([1.0, 1.0, 0.5, 0.5][:nd] if nd == 4 else [1.0, 1.0, 0.5], dt)
    ref = oracle.denoise(x, mu, its, fista, stopping_relative_change=stop)
    assert bits_equal(np.concatenate([p["own"] for p in parts]), ref["recon"])
    for p in parts:
        assert np.array_equal(p["dl"] == 0, ref["delta_recon"] == 0)
