"""bench.py's host-side pieces that need no GPU: byte accounting, the launcher-less --gpus N path, the workload names
that key profiles/traffic.json."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_pass_counts_match_survey_8d():
    # algorithmic: every array of the reference's state read once and written once
    assert bench.passes_algorithmic(4, True) == 19 and bench.passes_algorithmic(4, False) == 11
    assert bench.passes_algorithmic(3, True) == 15 and bench.passes_algorithmic(3, False) == 9
    # moved: the compact state
    assert bench.passes_moved(4, True, "compact") == 15 and bench.passes_moved(4, True, "reference") == 19
    assert bench.passes_moved(3, True, "compact") == 12 and bench.passes_moved(4, False, "compact") == 11


def test_traffic_table_is_keyed_by_the_names_bench_prints():
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    for shape, dn, fista, slab in (((256, 256, 128, 128), "f32", True, 0), ((256, 256, 128, 128), "f64", False, 0),
                                   ((128, 128, 512), "f32", True, 0), ((512, 512, 256, 256), "f32", True, 8)):
        key = bench.workload_name(shape, dn, fista, 1, slab) + "|compact"
        assert key in t and t[key]["traffic_bytes"] > 0, key


def test_gpus_without_a_launcher_starts_the_ranks_itself(monkeypatch):
    calls = []
    monkeypatch.setattr(bench.subprocess, "call", lambda cmd: calls.append(cmd) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and len(calls) == 1
    cmd = calls[0]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    assert cmd[-4:] == ["--gpus", "4", "--steps", "7"] and cmd[-5].endswith("bench.py")


def test_world_size_must_match(monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.setenv("WORLD_SIZE", "3")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=3" in str(e.value)


def test_host_probes():
    assert bench.host_cores() >= 1 and bench.mem_available_gib() > 0


def test_traffic_json_is_what_the_script_makes_of_the_committed_csvs(tmp_path):
    """profiles/traffic.json (what bench.py copies into roofline.traffic) is generated, not edited: tools/make_traffic_json.py
    rebuilds it byte for byte from the committed per-workload PMC CSVs."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_traffic_json
    out = tmp_path / "traffic.json"
    make_traffic_json.write_traffic("r06", str(out))
    assert out.read_bytes() == open(os.path.join(ROOT, "profiles", "traffic.json"), "rb").read()
