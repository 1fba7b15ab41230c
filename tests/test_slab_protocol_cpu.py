"""The N>1 path without a GPU: SlabLayout bookkeeping, and SlabRunner's halo exchange rehearsed
with the gloo backend (world_size 2 and 3) on a CPU test double of the HIP backend.

What is proven here: cutting axis 0 into slabs, advancing each slab with the per-slab semantics of
tvdn_iterate_fused and exchanging ONE recon row per neighbour per iteration reproduces the
single-process result bit for bit, for Jia-Zhao (chain) and periodic (ring) BCs.
The HIP kernel's own per-slab semantics are checked on the GPU in test_gpu_slabs.py.
"""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch.multiprocessing as mp

from cytvdn_amd import _lib, synth
from cytvdn_amd.engine import SlabLayout, SlabRunner, fista_ratios


def test_layout_chain_and_ring():
    lay = [SlabLayout((10, 3, 4, 5), r, 3, 2) for r in range(3)]
    assert [(l.g0, l.g1) for l in lay] == [(0, 3), (3, 6), (6, 10)]
    assert [l.halo_lo for l in lay] == [0, 1, 1] and [l.halo_hi for l in lay] == [1, 1, 0]
    assert [l.lo_mode for l in lay] == [_lib.EDGE_BC, _lib.EDGE_HALO, _lib.EDGE_HALO]
    assert [l.hi_mode for l in lay] == [_lib.EDGE_HALO, _lib.EDGE_HALO, _lib.EDGE_ZERO]
    assert [l.left for l in lay] == [None, 0, 1] and [l.right for l in lay] == [1, 2, None]
    assert lay[1].local_shape == (5, 3, 4, 5) and (lay[1].row_lo, lay[1].row_hi) == (1, 4)
    assert list(lay[2].local_rows_global()) == [5, 6, 7, 8, 9]
    ring = [SlabLayout((10, 3, 4), r, 3, 0) for r in range(3)]
    assert all(l.halo_lo == 1 and l.halo_hi == 1 for l in ring)
    assert [l.left for l in ring] == [2, 0, 1] and [l.right for l in ring] == [1, 2, 0]
    assert list(ring[0].local_rows_global()) == [9, 0, 1, 2, 3]
    assert all(l.lo_mode == _lib.EDGE_HALO and l.hi_mode == _lib.EDGE_HALO for l in ring)
    one = SlabLayout((4, 2, 2), 0, 1, 0)
    assert one.local_shape == (4, 2, 2) and one.lo_mode == _lib.EDGE_BC and one.hi_mode == _lib.EDGE_BC
    with pytest.raises(ValueError):
        SlabLayout((2, 4, 4), 0, 3, 2)
    with pytest.raises(NotImplementedError):
        SlabLayout((8, 4, 4), 0, 2, 1)


def test_layout_explicit_bounds():
    """Uneven slabs given as explicit row bounds (thin neighbours around one big slab: how a multi-GPU slab shape is
    rehearsed on one GPU, tests/test_gpu_fullsize.py)."""
    b = (0, 3, 67, 70)
    lay = [SlabLayout((70, 2, 2, 4), r, 3, 2, bounds=b) for r in range(3)]
    assert [(l.g0, l.g1) for l in lay] == [(0, 3), (3, 67), (67, 70)]
    assert lay[1].local_shape == (66, 2, 2, 4) and lay[1].lo_mode == lay[1].hi_mode == _lib.EDGE_HALO
    assert list(lay[1].local_rows_global()) == list(range(2, 68))
    assert lay[2].hi_mode == _lib.EDGE_ZERO and lay[0].lo_mode == _lib.EDGE_BC
    for bad in ((0, 3, 70), (0, 3, 3, 70), (1, 3, 67, 70), (0, 3, 67, 71)):
        with pytest.raises(ValueError):
            SlabLayout((70, 2, 2, 4), 0, 3, 2, bounds=bad)


def test_fista_schedule_matches_reference_recurrence():
    r = fista_ratios(5)
    assert r[0] == 0.0
    tk = 1.0
    for i in range(5):
        tk_new = (1 + np.sqrt(1 + 4 * tk ** 2)) / 2
        assert r[i] == (tk - 1.0) / tk_new
        tk = tk_new


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, shape, dtype_name, bc, n_f, n_p, outdir, bounds=None):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    from slab_cpu_backend import OracleSlabBackend
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dtype = np.dtype(dtype_name)
        nd = len(shape)
        lay = SlabLayout(tuple(shape), rank, world, bc, bounds=bounds)
        be = OracleSlabBackend(lay, dtype, n_f > 0, max_iters=n_f + n_p)
        mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dtype)
        lam = mu / dtype.type(32.0 if nd == 4 else 16.0)
        be.set_params(1.0 / lam, (lam / mu).astype(dtype))
        full = synth.cube(shape, seed=91, dtype=dtype) + dtype.type(0.25)
        be.set_input(full[lay.local_rows_global()])          # own rows + halo rows (periodic wrap applied)
        runner = SlabRunner(be)
        runner.run(n_f, n_p)
        own = be.recon_tensor().numpy()[lay.row_lo:lay.row_hi]
        sums = runner.global_sums().numpy()
        np.savez(os.path.join(outdir, f"r{rank}.npz"), own=own, g0=lay.g0, g1=lay.g1, sums=sums)
    finally:
        dist.destroy_process_group()


CASES = [
    # world, shape, dtype, bc, n_fista, n_plain
    (2, (7, 3, 4, 8), "float32", 2, 5, 0),
    (2, (6, 5, 8), "float64", 2, 0, 5),
    (2, (5, 3, 4, 4), "float32", 0, 4, 2),      # periodic ring of two: one peer is both neighbours
    (3, (8, 2, 3, 4), "float64", 2, 3, 2),
    (3, (7, 4, 6), "float32", 0, 4, 0),
    (2, (2, 3, 3, 4), "float32", 2, 3, 0),      # one row per slab
    (3, (9, 2, 3, 4), "float32", 2, 4, 1, (0, 1, 8, 9)),   # explicit bounds: thin slabs around a big one
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join("x".join(map(str, v)) if isinstance(v, tuple) else str(v) for v in c))
def test_slabs_reproduce_single_process(oracle, case):
    world, shape, dtype, bc, n_f, n_p = case[:6]
    bounds = case[6] if len(case) > 6 else None
    dt = np.dtype(dtype)
    nd = len(shape)
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(world, _free_port(), shape, dtype, bc, n_f, n_p, tmp, bounds), nprocs=world, join=True)
        parts = [np.load(os.path.join(tmp, f"r{r}.npz")) for r in range(world)]
    recon = np.concatenate([p["own"] for p in parts], axis=0)
    x = synth.cube(shape, seed=91, dtype=dt) + dt.type(0.25)
    mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
    its = [n_f, n_p] if (n_f and n_p) else (n_f or n_p)
    ref = oracle.denoise(x, mu, its, n_f > 0, BC_mode=bc)
    assert recon.tobytes() == ref["recon"].tobytes()
    sums = parts[0]["sums"]
    for p in parts[1:]:
        assert np.array_equal(p["sums"], sums)           # all-reduced: every rank holds the global sums
    np.testing.assert_allclose(sums[:, 0], ref["b_norm64"], rtol=1e-12)
    np.testing.assert_allclose(sums[:, 1], ref["delta64"], rtol=1e-12)
    np.testing.assert_allclose(sums[:, 2], ref["rnorm64"], rtol=1e-12)


def _worker_api(rank, world, port, shape, dtype_name, its, fista, stop, outdir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    from slab_cpu_backend import OracleSlabBackend
    from cytvdn_amd.distributed import denoise_slabs, slab_rows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dtype = np.dtype(dtype_name)
        nd = len(shape)
        full = synth.cube(shape, seed=14, dtype=dtype)
        g0, g1 = slab_rows(shape, rank, world)
        mu = np.array([1.0, 1.0, 0.5, 0.5][:nd] if nd == 4 else [1.0, 1.0, 0.5], dtype)
        n = sum(its) if isinstance(its, list) else its
        fac = lambda lay: OracleSlabBackend(lay, dtype, fista or isinstance(its, list), max_iters=n)
        own, bn, dl = denoise_slabs(full[g0:g1], shape, mu, its, FISTA=fista, stopping_relative_change=stop,
                                    backend_factory=fac)
        np.savez(os.path.join(outdir, f"r{rank}.npz"), own=own, bn=bn, dl=dl)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,dtype,its,fista,stop", [
    (2, (6, 5, 8, 12), "float32", 40, False, 0.02),       # global stopping criterion hit mid-run
    (3, (7, 6, 16), "float64", [30, 6], True, 0.03),       # hybrid: FISTA phase stops early, plain phase still runs
    (2, (6, 5, 8, 12), "float64", 9, True, None),
], ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else str(v))
def test_denoise_slabs_api_matches_single_process(oracle, world, shape, dtype, its, fista, stop):
    dt = np.dtype(dtype)
    nd = len(shape)
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker_api, args=(world, _free_port(), shape, dtype, its, fista, stop, tmp), nprocs=world, join=True)
        parts = [np.load(os.path.join(tmp, f"r{r}.npz")) for r in range(world)]
    x = synth.cube(shape, seed=14, dtype=dt)
    mu = np.array([1.0, 1.0, 0.5, 0.5][:nd] if nd == 4 else [1.0, 1.0, 0.5], dt)
    ref = oracle.denoise(x, mu, its, fista, stopping_relative_change=stop)
    assert np.concatenate([p["own"] for p in parts]).tobytes() == ref["recon"].tobytes()
    for p in parts:
        assert np.array_equal(p["dl"] == 0, ref["delta_recon"] == 0)          # same iteration count on every rank
        np.testing.assert_allclose(p["dl"], ref["delta_recon"], rtol=1e-4 if dt == np.float32 else 1e-12)
        np.testing.assert_allclose(p["bn"], ref["b_norm"], rtol=1e-4 if dt == np.float32 else 1e-12)


from hypothesis import given, settings, strategies as st


@settings(max_examples=200, deadline=None, derandomize=True)
@given(n0=st.integers(1, 300), world=st.integers(1, 16), bc=st.sampled_from([0, 2]))
def test_layout_invariants(n0, world, bc):
    """Slabs tile axis 0 exactly, neighbours are mutual, halo rows hold the neighbour's edge rows."""
    if n0 < world:
        with pytest.raises(ValueError):
            SlabLayout((n0, 2, 3), 0, world, bc)
        return
    lays = [SlabLayout((n0, 2, 3), r, world, bc) for r in range(world)]
    assert lays[0].g0 == 0 and lays[-1].g1 == n0
    for a, b in zip(lays, lays[1:]):
        assert a.g1 == b.g0 and a.own_rows >= 1
    assert sum(l.own_rows for l in lays) == n0
    for l in lays:
        rows = l.local_rows_global()
        assert len(rows) == l.local_shape[0] == l.halo_lo + l.own_rows + l.halo_hi
        assert list(rows[l.row_lo:l.row_hi]) == list(range(l.g0, l.g1))
        if l.left is not None:
            assert lays[l.left].right == l.rank and rows[0] == (l.g0 - 1) % n0 == lays[l.left].g1 - 1
        if l.right is not None:
            assert lays[l.right].left == l.rank and rows[-1] == l.g1 % n0 == lays[l.right].g0
        if world == 1:
            assert l.halo_lo == l.halo_hi == 0 and l.lo_mode == _lib.EDGE_BC and l.hi_mode == _lib.EDGE_BC
        elif bc == 2:
            assert (l.lo_mode == _lib.EDGE_BC) == (l.rank == 0)
            assert l.hi_mode == (_lib.EDGE_ZERO if l.rank == world - 1 else _lib.EDGE_HALO)
        else:
            assert l.lo_mode == l.hi_mode == _lib.EDGE_HALO


def _hooks_worker(rank, world, port, periodic, outdir):
    """What the library's streamed loop asks of a rank between passes (tvdn.h tvdn_slab_io), driven here by hand over gloo:
    every rank holds arrays of halo + own + halo rows whose own rows carry (rank, array, row) tags."""
    import ctypes as C
    import torch.distributed as dist
    from cytvdn_amd.distributed import _RankHooks
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        hooks = _RankHooks(dist, None, rank, world, 0, periodic)
        depth, own, row_bytes, n_arr = 2, 3 + rank, 24, 3
        lo, hi = depth, depth + own
        arrs = [np.full((own + 2 * depth, row_bytes), 255, np.uint8) for _ in range(n_arr)]
        for i, a in enumerate(arrs):
            for r in range(own):
                a[lo + r] = 16 * rank + 4 * i + (r % 4)            # a tag per (rank, array, own row mod 4)
        ptrs = (C.c_void_p * n_arr)(*[a.ctypes.data for a in arrs])
        assert hooks.exchange(None, n_arr, ptrs, own + 2 * depth, lo, hi, depth, row_bytes) == 0, hooks.error
        s3 = (C.c_double * 3)(1.0 + rank, 10.0, 0.5 * rank)
        assert hooks.allreduce(None, s3) == 0, hooks.error
        planes = np.full((4, row_bytes), 7 + rank, np.uint8)
        assert hooks.relay_row0(None, 1 if rank == 0 else 0, planes.ctypes.data, 4, row_bytes) == 0, hooks.error   # a broadcast
        hooks.finish()
        np.savez(os.path.join(outdir, f"h{rank}.npz"), arrs=np.stack(arrs), sums=np.array(list(s3)), planes=planes, own=own)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,periodic", [(2, False), (3, False), (3, True), (2, True)])
def test_streamed_slab_hooks_over_gloo(world, periodic):
    """The multi-process streamed run's exchange (distributed._RankHooks): after it the low halo rows of every array are the
    left neighbour's highest own rows and the high halo rows the right neighbour's lowest, rows at the cube's two ends
    untouched unless the boundary is periodic; the sums are all-reduced; rank 0's wrap planes arrive at every other rank."""
    with tempfile.TemporaryDirectory() as tmp:
        mp.start_processes(_hooks_worker, args=(world, _free_port(), periodic, tmp), nprocs=world, join=True, start_method="spawn")
        out = [np.load(os.path.join(tmp, f"h{r}.npz")) for r in range(world)]
    d = 2
    for r in range(world):
        a, own = out[r]["arrs"], int(out[r]["own"])
        left = (r - 1) % world if (r > 0 or periodic) else None
        right = (r + 1) % world if (r < world - 1 or periodic) else None
        for i in range(3):
            if left is None:
                assert (a[i, :d] == 255).all()
            else:
                lown = int(out[left]["own"])
                assert np.array_equal(a[i, :d], out[left]["arrs"][i, d + lown - d:d + lown])
            if right is None:
                assert (a[i, d + own:] == 255).all()
            else:
                assert np.array_equal(a[i, d + own:], out[right]["arrs"][i, d:2 * d])
            assert (a[i, d:d + own] // 16 == r).all()                # own rows never written
        assert np.allclose(out[r]["sums"], [sum(1.0 + q for q in range(world)), 10.0 * world, sum(0.5 * q for q in range(world))])
    assert all((out[r]["planes"] == 7).all() for r in range(world))     # rank 0's planes, on every rank


def _host_check_worker(rank, world, port, needs, hosts, available, outdir):
    import torch.distributed as dist
    from cytvdn_amd.distributed import _check_hosts_hold_the_slabs
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        try:
            _check_hosts_hold_the_slabs(dist, None, world, needs[rank], "no device" if needs[rank] < 0 else None,
                                        host=hosts[rank], available=available)
            verdict = "ok"
        except MemoryError as e:
            verdict = "memory: " + str(e)
        except RuntimeError as e:
            verdict = "error: " + str(e)
        with open(os.path.join(outdir, f"v{rank}.txt"), "w") as f:
            f.write(verdict)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("needs,hosts,want", [
    ((30, 30, 30), ("a", "a", "b"), "ok"),               # 60 on host a, 30 on host b, 80 of 100 allowed each
    ((50, 50, 30), ("a", "a", "b"), "memory"),           # two ranks that would each pass overdraw host a together
    ((50, 50, 50), ("a", "b", "c"), "ok"),               # ... and do not when every rank has a host of its own
    ((30, -1, 30), ("a", "a", "b"), "error"),            # a rank that cannot size its slab: every rank hears of it
])
def test_the_ranks_of_a_host_add_up_what_they_will_page_lock(needs, hosts, want):
    """denoise_slabs(staged=...): before any rank page-locks its slab the ranks gather what each will take
    (tvdn_slab_host_need) and every rank reaches the SAME verdict per host -- nobody waits in a collective for a rank that has
    bailed out, and no host (or control group) is driven out of memory by ranks that each looked at their own share only."""
    world = len(needs)
    with tempfile.TemporaryDirectory() as tmp:
        mp.start_processes(_host_check_worker, args=(world, _free_port(), needs, hosts, 100, tmp), nprocs=world, join=True,
                           start_method="spawn")
        verdicts = [open(os.path.join(tmp, f"v{r}.txt")).read() for r in range(world)]
    assert all(v.startswith(want) for v in verdicts), verdicts
    assert len(set(verdicts)) == 1                                   # the same text on every rank
    if want == "memory":
        assert "a: 2 ranks" in verdicts[0]
