"""The multi-process streamed path at the plane size of BASELINE config 5 (1024 x 256 x 256 f32: 256 MiB per row), on one GPU.

Two PROCESSES stream an 8-row slab each from page-locked host memory, halo rows over gloo, the plan left to `staged="auto"`.
Round 4 lost a box on a run like this one (two ranks page-locking 100-120 GiB each under one control group; each rank's guard
saw only itself).  What keeps this one safe, and what it checks on hardware:
  * TVDN_HOST_LIMIT=110G: the ranks together may hold 0.8 x 110 = 88 GiB, and the plan they follow (k = 2) holds 86: 30 GiB
    page-locked per rank (10 arrays x (8 + 2 k) rows) = 60 GiB pinned, plus each rank's slab, its result and the rows of one swap
    in flight over gloo;
  * the same plan under TVDN_HOST_LIMIT=40G is REFUSED by `distributed._check_hosts_hold_the_slabs` on both ranks, before
    either has pinned a byte, and the process group is usable afterwards: the real run follows in the same processes;
  * skipped when the host offers less than 110 GiB.
"""
import json
import os
import tempfile

import numpy as np
import pytest

from golden_util import bits_equal
from test_gpu_two_ranks import _free_port

pytestmark = pytest.mark.gpu

PLANE = (1024, 256, 256)
OWN = 8
ITS = 4
HOST_LIMIT = "110G"
# windows of the result (global start, extent), each inside one rank's rows: the cube's first rows, both sides of the seam
# between the ranks, the cube's last rows
WINDOWS = [((0, 0, 0, 0), (2, 6, 8, 16)),
           ((OWN - 2, PLANE[0] - 6, PLANE[1] - 9, PLANE[2] - 16), (2, 6, 9, 16)),
           ((OWN, 500, 100, 64), (2, 5, 6, 12)),
           ((2 * OWN - 2, PLANE[0] - 7, 0, PLANE[2] - 12), (2, 7, 8, 12))]


def _rank(rank, world, port, outdir):
    import torch
    import torch.distributed as dist
    from cytvdn_amd import _lib, synth
    from cytvdn_amd.distributed import denoise_slabs, slab_rows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    report = {}
    try:
        torch.cuda.set_device(0)
        shape = (world * OWN,) + PLANE
        g0, g1 = slab_rows(shape, rank, world)
        buf = torch.empty((g1 - g0,) + PLANE, dtype=torch.float32, device="cuda")
        _lib.check(_lib.lib().tvdn_synth_fill(0, 4, _lib.shape_arr(shape), synth.SEED_4D, g0, g1 - g0, buf.data_ptr(),
                                              _lib.current_stream(0)))
        mine = buf.cpu().numpy()
        del buf
        torch.cuda.empty_cache()
        mu = np.array([1.0, 1.0, 0.5, 0.5], np.float32)
        # (1) over budget: 30 GiB page-locked + 13 GiB beside it per rank against 0.8 x 40 GiB -- every rank must refuse, nobody pins
        os.environ["TVDN_HOST_LIMIT"] = "40G"
        try:
            denoise_slabs(mine, shape, mu, ITS, device=0, staged=(1, 2, 0))
            report["refused"] = None
        except MemoryError as e:
            report["refused"] = str(e)
        os.environ["TVDN_HOST_LIMIT"] = HOST_LIMIT
        # (2) the plan the host holds, in the same process group
        own, bn, dl = denoise_slabs(mine, shape, mu, ITS, device=0, staged="auto")
        report["b_norm"] = [float(v) for v in bn]
        report["delta"] = [float(v) for v in dl]
        wins = {}
        for i, (start, ext) in enumerate(WINDOWS):
            if g0 <= start[0] and start[0] + ext[0] <= g1:
                wins[f"w{i}"] = own[(slice(start[0] - g0, start[0] - g0 + ext[0]),)
                                    + tuple(slice(s, s + e) for s, e in zip(start[1:], ext[1:]))].copy()
        np.savez(os.path.join(outdir, f"r{rank}.npz"), **wins)
    except BaseException as e:                       # the parent reports it; never leave the other rank waiting on a file
        report["error"] = f"{type(e).__name__}: {e}"
        raise
    finally:
        with open(os.path.join(outdir, f"r{rank}.json"), "w") as f:
            json.dump(report, f)
        dist.destroy_process_group()


def test_two_ranks_stream_config5_planes_within_a_shared_host_budget(oracle, monkeypatch):
    import torch
    import torch.multiprocessing as mp
    from cytvdn_amd import _lib, synth
    from cytvdn_amd.planner import HOST_FRACTION, host_available, plan_run
    monkeypatch.delenv("TVDN_HOST_LIMIT", raising=False)
    offered = host_available()
    if offered is None or offered < 110 * 2 ** 30:
        pytest.skip(f"the host offers {0 if offered is None else offered / 2 ** 30:.0f} GiB; this rehearsal wants 110")
    free, _ = torch.cuda.mem_get_info(0)
    if free < 100 * 2 ** 30:
        pytest.skip("needs 100 GiB of free HBM (two ranks x 40 GiB and the parent's copy of the cube)")
    monkeypatch.setenv("TVDN_HOST_LIMIT", HOST_LIMIT)     # (the spawned ranks inherit both)
    monkeypatch.setenv("TVDN_HBM_LIMIT", "40G")
    world = 2
    shape = (world * OWN,) + PLANE
    plan = plan_run(shape, np.dtype(np.float32), True, world, stop=False, device=0, swap_through_host=True)
    assert plan["mode"] == "slabs+wavefront" and plan["k"] == 2 and plan["resident_rows_per_rank"] == 0, plan
    assert world * plan["host_bytes_per_rank"] == 60 * 2 ** 30, plan
    assert world * (plan["host_bytes_per_rank"] + plan["host_beside_per_rank"]) == 86 * 2 ** 30 <= HOST_FRACTION * 110 * 2 ** 30
    with tempfile.TemporaryDirectory() as tmp:
        try:
            mp.start_processes(_rank, args=(world, _free_port(), tmp), nprocs=world, join=True, start_method="spawn")
        finally:
            reports = []
            for r in range(world):
                p = os.path.join(tmp, f"r{r}.json")
                reports.append(json.load(open(p)) if os.path.exists(p) else {"error": "no report"})
        assert not any("error" in rep for rep in reports), reports
        parts = [dict(np.load(os.path.join(tmp, f"r{r}.npz"))) for r in range(world)]
    # the over-budget variant: both ranks refused, with the same account of the host
    assert all(rep["refused"] for rep in reports), reports
    assert reports[0]["refused"] == reports[1]["refused"]
    assert "2 ranks x their slabs = 86.0 GiB" in reports[0]["refused"], reports[0]["refused"]
    # the traces are global: identical on the ranks, finite
    assert reports[0]["b_norm"] == reports[1]["b_norm"] and reports[0]["delta"] == reports[1]["delta"]
    assert len(reports[0]["b_norm"]) == ITS and np.all(np.isfinite(reports[0]["b_norm"])) and np.all(np.array(reports[0]["delta"]) > 0)
    # windows of the result against the oracle on the enlarged windows of the input
    got = {}
    for p in parts:
        got.update(p)
    assert sorted(got) == [f"w{i}" for i in range(len(WINDOWS))]
    buf = torch.empty(shape, dtype=torch.float32, device="cuda")
    _lib.check(_lib.lib().tvdn_synth_fill(0, 4, _lib.shape_arr(shape), synth.SEED_4D, 0, shape[0], buf.data_ptr(),
                                          _lib.current_stream(0)))
    mu = np.array([1.0, 1.0, 0.5, 0.5], np.float32)
    halo = 2 * ITS
    for i, (start, ext) in enumerate(WINDOWS):
        lo = [max(0, s - halo) for s in start]
        hi = [min(n, s + e + halo) for s, e, n in zip(start, ext, shape)]
        x = buf[tuple(slice(a, b) for a, b in zip(lo, hi))].cpu().numpy().copy()
        ref = oracle.denoise(x, mu, ITS, True)["recon"]
        inner = tuple(slice(s - a, s - a + e) for s, a, e in zip(start, lo, ext))
        assert bits_equal(got[f"w{i}"], ref[inner]), (start, ext)
    del buf
    torch.cuda.empty_cache()
