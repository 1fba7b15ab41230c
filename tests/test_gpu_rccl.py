"""The RCCL data path itself: one process per GPU, halo rows sent device-to-device over xGMI
(`SlabRunner.step_overlapped`: edge rows first, transfer on a side HIP stream under the interior sweep; and the
blocking exchange), chain (Jia-Zhao) and ring (periodic BC, where with two ranks both neighbours are the same peer),
bit for bit against the oracle.  Needs >= 2 GPUs (>= 3 for the three-rank cases): collected and SKIPPED on the
one-GPU box, so that the first multi-GPU machine that runs the suite checks the path before any number is taken
from it.  `bench.py --gpus N` runs the same check as a pre-flight (cytvdn_amd.distributed.selfcheck_exchange)."""
import os
import socket
import tempfile

import numpy as np
import pytest

from golden_util import bits_equal

pytestmark = pytest.mark.gpu


def _ngpu():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, shape, dtype_name, bc, n_f, n_p, overlap, outdir):
    import torch
    import torch.distributed as dist
    from cytvdn_amd import synth
    from cytvdn_amd.engine import HipBackend, SlabLayout, SlabRunner
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        dt = np.dtype(dtype_name)
        nd = len(shape)
        lay = SlabLayout(tuple(shape), rank, world, bc)
        be = HipBackend(lay, dt, n_f > 0, device=rank, max_iters=n_f + n_p)
        mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
        lam = mu / dt.type(32.0 if nd == 4 else 16.0)
        be.set_params(1.0 / lam, (lam / mu).astype(dt))
        full = synth.cube(shape, seed=91, dtype=dt) + dt.type(0.25)
        be.set_input(full[lay.local_rows_global()])
        run = SlabRunner(be)
        run.overlap = bool(overlap)
        assert run.transport == "rccl"
        run.run(n_f, n_p)
        own = be.recon_tensor()[lay.row_lo:lay.row_hi].cpu().numpy()
        sums = run.global_sums().cpu().numpy()
        np.savez(os.path.join(outdir, f"r{rank}.npz"), own=own, sums=sums)
    finally:
        dist.destroy_process_group()


CASES = [
    # world, shape, dtype, bc, n_fista, n_plain
    (2, (12, 3, 8, 16), "float32", 2, 6, 2),     # chain
    (2, (10, 6, 16), "float64", 0, 5, 0),        # ring of two: left and right neighbour are the same peer
    (3, (13, 2, 4, 8), "float32", 2, 5, 0),      # uneven slabs
    (3, (12, 4, 8), "float32", 0, 4, 2),         # ring of three, hybrid schedule
]


@pytest.mark.parametrize("overlap", [True, False], ids=["overlapped", "blocking"])
@pytest.mark.parametrize("world,shape,dtype,bc,n_f,n_p", CASES,
                         ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else str(v))
def test_rccl_halo_exchange_matches_oracle(oracle, world, shape, dtype, bc, n_f, n_p, overlap):
    if _ngpu() < world:
        pytest.skip(f"needs {world} GPUs for an RCCL run (one rank per GPU); this box has {_ngpu()}")
    import torch.multiprocessing as mp
    from cytvdn_amd import synth
    dt = np.dtype(dtype)
    nd = len(shape)
    with tempfile.TemporaryDirectory() as tmp:
        mp.start_processes(_worker, args=(world, _free_port(), shape, dtype, bc, n_f, n_p, overlap, tmp), nprocs=world,
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


def _selfcheck_worker(rank, world, port, backend, outdir, on_granules=False):
    import json
    import torch
    import torch.distributed as dist
    from cytvdn_amd.distributed import selfcheck_exchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = selfcheck_exchange(device=dev, on_granules=on_granules)
        json.dump(res, open(os.path.join(outdir, f"r{rank}.json"), "w"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_exchange_selfcheck(backend):
    """The pre-flight that bench.py runs before a multi-GPU measurement: over gloo with two ranks sharing this GPU
    (always runs), over RCCL when the box has two GPUs."""
    import json
    import torch.multiprocessing as mp
    if backend == "nccl" and _ngpu() < 2:
        pytest.skip("needs 2 GPUs")
    with tempfile.TemporaryDirectory() as tmp:
        mp.start_processes(_selfcheck_worker, args=(2, _free_port(), backend, tmp), nprocs=2, join=True,
                           start_method="spawn")
        res = [json.load(open(os.path.join(tmp, f"r{r}.json"))) for r in range(2)]
    assert res[0] == res[1]
    assert res[0]["blocking"] and res[0]["overlap"] and res[0]["error"] is None
    assert res[0]["transport"] == ("rccl" if backend == "nccl" else "gloo")
    assert res[0]["state_mem"] == "plain"            # a few hundred KiB of state: below the granule threshold


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_exchange_selfcheck_on_granules(backend, monkeypatch):
    """the pre-flight on granules must really run on granules -- its states forced there (threshold 0), `state_mem`
    reported, anything else counted as a failure -- because that is the memory a slab of 2 GiB or more hands to the transport.
    Over gloo (two ranks sharing this GPU) always; over RCCL with two GPUs.  With granules switched off the same call FAILS."""
    import json
    import torch.multiprocessing as mp
    if backend == "nccl" and _ngpu() < 2:
        pytest.skip("needs 2 GPUs")
    monkeypatch.setenv("TVDN_GRANULE_MIB", "2")
    for vmm, want in (("1", True), ("0", False)):
        monkeypatch.setenv("TVDN_VMM", vmm)          # (the spawned ranks inherit it)
        with tempfile.TemporaryDirectory() as tmp:
            mp.start_processes(_selfcheck_worker, args=(2, _free_port(), backend, tmp, True), nprocs=2, join=True, start_method="spawn")
            res = [json.load(open(os.path.join(tmp, f"r{r}.json"))) for r in range(2)]
        assert res[0]["blocking"] == res[1]["blocking"] == want and res[0]["overlap"] == want, res
        assert res[0]["state_mem"] == ("granules" if want else "plain")
        assert (res[0]["error"] is None) == want


def _mixed_groups_worker(rank, world, port, outdir):
    """What bench.py's init_groups does: gloo control group first, RCCL data group beside it."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = dist.new_group(backend="nccl", device_id=torch.device("cuda", rank))
        dist.barrier(group=g)
        t = torch.full((3,), float(rank + 1), dtype=torch.float64, device=f"cuda:{rank}")
        dist.all_reduce(t, group=g)
        c = torch.tensor([float(rank + 1)])
        dist.all_reduce(c)                                  # default group: gloo, host memory
        ok = dist.get_backend(g) == "nccl" and dist.get_backend() == "gloo"
        open(os.path.join(outdir, f"r{rank}.txt"), "w").write(f"{ok} {t[0].item()} {c[0].item()}")
    finally:
        dist.destroy_process_group()


def test_rccl_data_group_beside_gloo_control_group():
    """bench.py --gpus N keeps its decisions and timing reductions on a gloo group and moves halo rows over an RCCL
    group created beside it (`init_groups`).  One rank is enough to check that this torch/RCCL build accepts the
    arrangement; with two GPUs the all-reduces really cross."""
    import torch.multiprocessing as mp
    world = 2 if _ngpu() >= 2 else 1
    with tempfile.TemporaryDirectory() as tmp:
        mp.start_processes(_mixed_groups_worker, args=(world, _free_port(), tmp), nprocs=world, join=True, start_method="spawn")
        res = [open(os.path.join(tmp, f"r{r}.txt")).read().split() for r in range(world)]
    want = float(sum(range(1, world + 1)))
    for ok, t, c in res:
        assert ok == "True" and float(t) == want and float(c) == want


def _staged_worker(rank, world, port, shape, dtype_name, bc, its, staged, stop, bad, hbm, outdir):
    import torch
    import torch.distributed as dist
    from cytvdn_amd import synth
    from cytvdn_amd.distributed import denoise_slabs, slab_rows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if hbm:
        os.environ["TVDN_HBM_LIMIT"] = hbm
    backend = os.environ.get("TVDN_TEST_BACKEND", "nccl")        # "gloo": the same cases with every rank on GPU 0 (a rehearsal of
    dev = rank if backend == "nccl" else 0                       # this file's code on a one-GPU box; nothing of RCCL runs then)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        dt = np.dtype(dtype_name)
        nd = len(shape)
        full = synth.cube(shape, seed=91, dtype=dt) + dt.type(0.25)
        if bad:
            full[0, ..., 1] = np.inf
        g0, g1 = slab_rows(shape, rank, world, bc)
        mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
        own, bn, dl = denoise_slabs(full[g0:g1], shape, mu, its, FISTA=True, stopping_relative_change=stop, BC_mode=bc,
                                    device=dev, staged=staged)
        np.savez(os.path.join(outdir, f"r{rank}.npz"), own=own, bn=bn, dl=dl)
    finally:
        dist.destroy_process_group()


STAGED_CASES = [
    # world, shape, dtype, bc, iterations, staged, stop, non-finite first row, TVDN_HBM_LIMIT
    (2, (20, 3, 4, 8), "float32", 2, 9, (4, 3), None, False, None),            # k-row swaps over RCCL (rows staged through HBM)
    (2, (20, 3, 4, 8), "float32", 2, 9, (4, 3, 0), None, False, None),         # ... with no row resident
    (3, (19, 6, 16), "float64", 2, [5, 4], (3, 4), None, False, None),         # uneven slabs, hybrid schedule
    (3, (9, 3, 4, 8), "float32", 2, 7, (2, 3), None, True, None),              # row-0 broadcast: the middle rank needs it too
    (2, (18, 3, 4, 8), "float32", 0, 6, (2, 4), None, False, None),            # periodic: a ring of two
    (2, (12, 5, 8, 12), "float32", 2, [30, 6], (5, 8), 0.03, False, None),     # stopping rule: an all-reduce per iteration
    (2, (40, 8, 32, 64), "float32", 2, 9, "auto", None, False, "12M"),         # the planner decides: streamed
    (2, (40, 8, 32, 64), "float32", 2, 9, "auto", None, False, "1G"),          # ... resident slabs
]


@pytest.mark.parametrize("world,shape,dtype,bc,its,staged,stop,bad,hbm", STAGED_CASES,
                         ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else str(v))
def test_rccl_staged_slabs_match_oracle(oracle, world, shape, dtype, bc, its, staged, stop, bad, hbm):
    """BASELINE configs[4] in structure over RCCL: every rank streams its slab through ITS GPU (tvdn_run with a tvdn_slab_io), the
    hooks moving the k halo rows, the sums and the wrap planes device to device.  The gloo form of the same runs on one GPU in
    tests/test_gpu_two_ranks.py; this is the one that needs the GPUs (TVDN_TEST_BACKEND=gloo rehearses this file's own code with
    every rank on GPU 0)."""
    if os.environ.get("TVDN_TEST_BACKEND", "nccl") == "nccl" and _ngpu() < world:
        pytest.skip(f"needs {world} GPUs for an RCCL run (one rank per GPU); this box has {_ngpu()}")
    import torch.multiprocessing as mp
    from cytvdn_amd import synth
    dt = np.dtype(dtype)
    nd = len(shape)
    with tempfile.TemporaryDirectory() as tmp:
        mp.start_processes(_staged_worker, args=(world, _free_port(), shape, dtype, bc, its, staged, stop, bad, hbm, tmp),
                           nprocs=world, join=True, start_method="spawn")
        parts = [np.load(os.path.join(tmp, f"r{r}.npz")) for r in range(world)]
    x = synth.cube(shape, seed=91, dtype=dt) + dt.type(0.25)
    if bad:
        x[0, ..., 1] = np.inf
    mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
    ref = oracle.denoise(x, mu, its, True, BC_mode=bc, stopping_relative_change=stop)
    assert bits_equal(np.concatenate([p["own"] for p in parts], axis=0), ref["recon"])
    for p in parts:
        assert np.array_equal(p["dl"] == 0, ref["delta_recon"] == 0)
