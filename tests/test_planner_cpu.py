"""Capacity planner: engine and slab count for BASELINE.json's five configurations on 288 GB
devices, and the TVDN_HBM_LIMIT knob.  Pure arithmetic: runs without a GPU."""
import pytest

from cytvdn_amd.planner import _parse_bytes, plan_run, state_arrays, wavefront_windows

GIB = 2 ** 30
HBM = 268 * GIB       # what an MI355X reports free (288 GB)


def test_state_arrays():
    assert state_arrays(4, True) == 15 and state_arrays(4, False) == 11
    assert state_arrays(3, True) == 12 and state_arrays(3, False) == 9


def test_baseline_configs():
    c1 = plan_run((128, 128, 512), "float32", True, 1, hbm_bytes=HBM)
    assert c1["mode"] == "in-core" and c1["bytes_per_gpu"] == 12 * 32 * 2 ** 20
    c2 = plan_run((256, 256, 128, 128), "float32", True, 1, hbm_bytes=HBM)
    assert c2["mode"] == "in-core" and c2["bytes_per_gpu"] == 60 * GIB and c2["n_slabs"] == 1
    c3 = plan_run((256, 256, 128, 128), "float64", False, 1, hbm_bytes=HBM)
    assert c3["mode"] == "in-core" and c3["bytes_per_gpu"] == 88 * GIB
    c4 = plan_run((512, 512, 256, 256), "float32", True, 8, hbm_bytes=HBM)
    assert c4["mode"] == "slabs" and c4["n_slabs"] == 8 and c4["min_slabs_in_core"] == 5
    assert c4["bytes_per_gpu"] == 15 * 66 * 128 * 2 ** 20            # 64 own rows + 2 halo rows of 128 MiB
    # the same cube on fewer GPUs than it needs, or on one: streamed from pinned host memory
    assert plan_run((512, 512, 256, 256), "float32", True, 4, hbm_bytes=HBM)["mode"] == "slabs+wavefront"
    one = plan_run((512, 512, 256, 256), "float32", True, 1, hbm_bytes=HBM)
    assert one["mode"] == "wavefront" and one["k"] >= 2 and one["bytes_per_gpu"] <= 0.85 * HBM
    assert one["bytes_per_gpu"] == wavefront_windows(4, one["chunk_rows"], one["k"]) * 128 * 2 ** 20
    # BASELINE config 5 on the 8 GPUs of ONE node with 3 TB of host memory: every rank streams its 128-row slab, keeps the
    # interior rows that fit beside the rings resident in HBM, and page-locks 10 arrays of (own + 2 k halo - resident) rows
    c5 = plan_run((1024, 1024, 256, 256), "float32", True, 8, hbm_bytes=HBM, host_bytes=3 * 10 ** 12)
    assert c5["mode"] == "slabs+wavefront" and c5["n_slabs"] == 8 and c5["min_slabs_in_core"] > 8
    assert c5["state_bytes"] == 15 * 256 * GIB
    res = c5["resident_rows_per_rank"]
    assert 0 < res <= 128 - 2 * c5["k"] and c5["bytes_per_gpu"] <= 0.85 * HBM
    assert c5["host_bytes_per_rank"] == 10 * (128 + 2 * c5["k"] - res) * 256 * 2 ** 20
    assert 8 * c5["host_bytes_per_rank"] <= 0.8 * 3 * 10 ** 12 and "WARNING" not in c5["why"]
    # with host memory to spare the same cube streams every row at the deepest k the rings allow (faster by the model)
    big = plan_run((1024, 1024, 256, 256), "float32", True, 8, hbm_bytes=HBM, host_bytes=8 * 10 ** 12)
    assert big["resident_rows_per_rank"] == 0 and big["k"] > c5["k"] and big["seconds_per_iteration_model"] < c5["seconds_per_iteration_model"]
    assert big["host_bytes_per_rank"] == 10 * (128 + 2 * big["k"]) * 256 * 2 ** 20
    # ... and on a host that cannot hold even the plan that page-locks least, the plan says so
    assert "WARNING" in plan_run((1024, 1024, 256, 256), "float32", True, 8, hbm_bytes=HBM, host_bytes=2 ** 38)["why"]


def test_stop_rule_selects_per_iteration_engine():
    p = plan_run((512, 512, 256, 256), "float32", True, 1, hbm_bytes=HBM, stop=True)
    assert p["mode"] == "wavefront" and p["k"] == 1 and p["chunk_rows"] >= 1
    assert plan_run((256, 256, 128, 128), "float32", True, 1, hbm_bytes=HBM, stop=True)["mode"] == "in-core"


def test_limit_knob_and_misfits(monkeypatch):
    assert _parse_bytes("48G") == 48 * GIB and _parse_bytes("24M") == 24 * 2 ** 20 and _parse_bytes("1.5GiB") == int(1.5 * GIB)
    assert _parse_bytes("1000") == 1000
    monkeypatch.setenv("TVDN_HBM_LIMIT", "24M")
    p = plan_run((40, 8, 32, 64), "float32", True, 1)      # 39 MB of state against 24 MB
    # (the shape of a streamed run is the library's own: tvdn_stream_plan with this much HBM, for a long run)
    assert p["hbm_bytes"] == 24 * 2 ** 20 and p["mode"] == "wavefront" and p["chunk_rows"] >= 1 and p["k"] >= 8
    assert p["bytes_per_gpu"] <= 0.85 * 24 * 2 ** 20 and 0 <= p["resident_rows_per_rank"] < 40
    ps = plan_run((40, 8, 32, 64), "float32", True, 1, stop=True)
    assert ps["mode"] == "wavefront" and ps["k"] == 1
    monkeypatch.setenv("TVDN_HBM_LIMIT", "1G")
    assert plan_run((40, 8, 32, 64), "float32", True, 1)["mode"] == "in-core"
    monkeypatch.setenv("TVDN_HBM_LIMIT", "64K")
    assert plan_run((40, 8, 32, 64), "float32", True, 1)["mode"] == "does-not-fit"
    with pytest.raises(TypeError):
        plan_run((4, 4), "float32")


def test_host_available_and_the_streamed_run_guard(monkeypatch):
    """A streamed plan whose pinned host state exceeds 80 % of what the host has available is refused before anything is
    allocated (page-locked memory cannot swap; a host out of memory takes every process on it down)."""
    from cytvdn_amd import planner
    real = planner.host_available()
    assert real is not None and 2 ** 28 < real < 2 ** 46              # this machine has between 256 MiB and 64 TiB
    monkeypatch.setenv("TVDN_HOST_LIMIT", "1G")
    assert planner.host_available() == 2 ** 30
    plan = planner.plan_run((64, 64, 128, 128), "float32", True, 1, hbm_bytes=2 ** 30)   # 256 MiB per array, 15 arrays
    assert plan["mode"] == "wavefront" and plan["host_bytes_per_rank"] > 2 ** 30
    with pytest.raises(MemoryError, match="page-locked host memory"):
        planner.check_host_fits(plan)
    monkeypatch.setenv("TVDN_HOST_LIMIT", "64G")
    if real > 8 * 2 ** 30:
        planner.check_host_fits(plan)                                  # 2.5-3.5 GiB of state: fine
    planner.check_host_fits(planner.plan_run((8, 8, 16, 16), "float32", True, 1, hbm_bytes=2 ** 30))   # in-core: nothing to check
    # eight ranks of BASELINE config 5 planned for a host with memory to spare (every row streamed, deep halos: 4.3 TiB) do not
    # fit one 3 TB node, and the check says so; planned for that node (rows resident in HBM, shallower halos) they do
    monkeypatch.delenv("TVDN_HOST_LIMIT")
    roomy = planner.plan_run((1024, 1024, 256, 256), "float32", True, 8, hbm_bytes=288 * 2 ** 30, host_bytes=8 * 10 ** 12)
    assert roomy["mode"] == "slabs+wavefront"
    monkeypatch.setattr(planner, "host_available", lambda: 3 * 10 ** 12)
    with pytest.raises(MemoryError):
        planner.check_host_fits(roomy, ranks_on_host=8)
    planner.check_host_fits(roomy, ranks_on_host=2)                    # four such nodes hold it
    c5 = planner.plan_run((1024, 1024, 256, 256), "float32", True, 8, hbm_bytes=288 * 2 ** 30)     # (host_available() = 3 TB here)
    assert c5["resident_rows_per_rank"] > 0
    planner.check_host_fits(c5, ranks_on_host=8)
