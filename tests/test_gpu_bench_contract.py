"""bench.py's output contract on a real GPU (small shapes, seconds): exactly one JSON line on stdout with the fields the
driver and the judge read, for the single-GPU path, the one-slab path and a two-rank gloo rehearsal."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f"stdout must carry ONE line, got {len(lines)}: {p.stdout[:500]}"
    return json.loads(lines[0])


def _check_common(d, n_gpus, steps, warmup):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "Gvoxel-iters/s" and d["n_gpus"] == n_gpus and d["steps"] == steps and d["warmup"] == warmup
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["value"] > 0 and d["ms_per_step"] > 0 and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-3)
    assert r["moved_frac"] <= r["frac"] + 1e-9 and r["kernel_ms"] > 0
    assert r["algorithmic_bytes_per_launch"] >= r["moved_bytes_per_launch"] > 0


def test_single_gpu_line():
    d = _run(["--shape", "24x16x32x64", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"])
    _check_common(d, 1, 4, 2)
    assert d["dtype"] == "f32" and d["cpu_baseline"] is None and "also" not in d           # also[] belongs to the headline shape
    assert d["config"]["global_shape"] == [24, 16, 32, 64] and d["config"]["parallelism"] == "single"
    d64 = _run(["--shape", "24x16x32x64", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--dtype", "f64", "--plain"])
    _check_common(d64, 1, 3, 1)
    assert d64["dtype"] == "f64" and d64["roofline"]["moved_frac"] == pytest.approx(d64["roofline"]["frac"], rel=1e-3)


def test_one_slab_line():
    d = _run(["--shape", "64x16x32x64", "--slab-of", "4", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    _check_common(d, 1, 3, 1)
    assert d["config"]["local_block"][0] == 18 and d["roofline"]["launches_per_step"] == 3
    assert d["config"]["parallelism"] == "one slab of 4"


def test_two_rank_rehearsal_line():
    d = _run(["--gpus", "2", "--shape", "16x16x32x64", "--steps", "3", "--warmup", "1"], env={"TVDN_DIST_BACKEND": "gloo"})
    _check_common(d, 2, 3, 1)
    assert d["config"]["transport"] == "gloo" and d["transport_fallback"] is False
    assert d["preflight"]["blocking"] is True and d["preflight"]["overlap"] is True and d["preflight"]["error"] is None
    assert d["config"]["parallelism"] == "slab2" and d["cpu_baseline"] is None


def test_a_stuck_multi_rank_run_leaves_a_line_behind():
    """RCCL has not met a second GPU in any round.  Should a collective never return there, bench.py's watchdog (a timer thread:
    it runs while the main thread waits in C++) prints ONE JSON line from rank 0 -- value null, the stage the run was in -- and
    ends every rank, instead of hanging until the driver kills it with nothing on stdout.  Rehearsed over gloo with a watchdog
    far shorter than the run: the line must come, well-formed, and name a stage."""
    e = dict(os.environ)
    e["TVDN_DIST_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shape", "16x16x32x64", "--steps", "3", "--warmup", "1",
                        "--watchdog-s", "0.05"], capture_output=True, text=True, env=e, timeout=600)
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, (p.stdout[:500], p.stderr[-800:])
    d = json.loads(lines[0])
    assert d["value"] is None and d["n_gpus"] == 2 and "watchdog" in d["error"] and "stage" in d["error"]
    assert p.returncode != 0
