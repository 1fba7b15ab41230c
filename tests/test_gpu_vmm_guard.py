"""The allocator of state blocks proves at run time what it leans on (csrc/tvdn_devmem.hip, "Trust").

The reference allocates its state with NumPy and can trust it (cyTVDN/cyTVDN.py:131-145).  The state of a run of 2 GiB or more
lives on granules of HIP virtual memory, whose correctness on ROCm 7.2 depends on a work-around for stale GPU translations (a
hipFree after every remap).  These tests hold the guard to its word: the canary runs before the first granule block and passes
on this runtime; with the flush disabled (TVDN_VMM_SKIP_FLUSH=1, a test knob) it TRIPS, the process falls back to plain blocks
and still computes the oracle's bits; failures of the allocator's HIP calls are reported, not swallowed; the pool of spare
granules honours TVDN_HBM_LIMIT; and the sequence in which round 5 saw one native abort is looped."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from golden_util import bits_equal

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_canary_has_passed_before_the_first_granule_block():
    import torch
    from cytvdn_amd import _lib
    b = _lib.DeviceBlock(3 << 30, 0)
    st = _lib.mem_status(0)
    assert b.kind == _lib.MEM_GRANULES and st["vmm_state"] == 1
    assert st["canary"] == _lib.CANARY_PASSED and st["canary_runs"] >= 1
    assert st["faults"] == 0 and st["first_fault"] == "", st
    assert st["blocks"] >= 1 and st["granules"] >= 3 and st["bytes"] >= 3 << 30
    assert st["flushes"] >= 4                      # three remaps of the canary + the block's own map
    assert st["last_granules"] >= 3 and st["last_pool"] >= st["last_granules"]
    t = b.tensor(torch.uint8)
    t[-1] = 9
    assert int(t[-1]) == 9
    del t
    b.free()
    assert _lib.mem_status(0)["faults"] == 0       # the release went through call by call
    assert _lib.mem_selftest(0) is True            # ... and the canary can be asked for at any time
    assert _lib.mem_status(0)["canary_runs"] >= st["canary_runs"] + 1


_CHILD = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, {root!r})
sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch
import cytvdn_amd as tv
from cytvdn_amd import _lib, synth
from oracle import oracle
oracle.build()
out = {{}}
b = _lib.DeviceBlock(64 << 20, 0)
out["kind"] = b.kind
t = b.tensor(torch.uint8); t.fill_(5); out["sum_ok"] = int(t.sum(dtype=torch.int64)) == 5 * (64 << 20); del t
b.free()
out["status"] = _lib.mem_status(0)
x = synth.cube((12, 6, 16, 32), seed=3, dtype=np.float32) + np.float32(0.25)
mu = np.array([1.0, 0.8, 0.5, 0.6], np.float32)
os.environ["TVDN_WORKSPACE"] = "library"          # the library allocates the state itself: the path that would take granules
recon, bn, dl = tv.denoise4D(x, mu, 6, FISTA=True, quiet=True)
ref = oracle.denoise(x, mu, 6, True)
out["bits"] = recon.tobytes() == ref["recon"].tobytes()
out["status_after"] = _lib.mem_status(0)
print("RESULT " + json.dumps(out))
"""


def _child(env_extra):
    env = dict(os.environ)
    env.update({"TVDN_VMM_MIN_MIB": "1", "TVDN_GRANULE_MIB": "8"})
    env.update(env_extra)
    p = subprocess.run([sys.executable, "-c", _CHILD.format(root=ROOT)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:]), p.stderr


def test_the_canary_trips_when_the_flush_is_skipped_and_the_run_falls_back_to_plain_blocks():
    """TVDN_VMM_SKIP_FLUSH=1 takes away the hipFree that makes the GPU forget replaced translations.  On a runtime with the
    defect (ROCm 7.2) the canary must see the old mapping's words, mark the device, say so -- and everything after it must
    still be right, on plain blocks.  (On a runtime WITHOUT the defect it passes either way: then the skip knob changes
    nothing and the test says which world it ran in.)"""
    from cytvdn_amd import _lib
    good, err_good = _child({})
    assert good["kind"] == _lib.MEM_GRANULES and good["sum_ok"] and good["bits"]
    assert good["status"]["canary"] == _lib.CANARY_PASSED and good["status"]["faults"] == 0
    assert "STALE TRANSLATION" not in err_good
    bad, err_bad = _child({"TVDN_VMM_SKIP_FLUSH": "1"})
    assert bad["sum_ok"] and bad["bits"], "results must not depend on what the state is made of"
    if bad["status"]["canary"] == _lib.CANARY_PASSED:
        pytest.skip("this runtime forgets replaced translations by itself: the work-around is no longer needed here")
    assert bad["status"]["canary"] == _lib.CANARY_STALE and bad["status"]["vmm_state"] == -1
    assert bad["kind"] == _lib.MEM_PLAIN and bad["status"]["blocks"] == 0
    assert bad["status_after"]["blocks"] == 0 and bad["status_after"]["canary_runs"] == 1     # marked once, not asked again
    assert "STALE TRANSLATION" in err_bad and "Granules are off" in err_bad


def test_a_release_that_fails_is_reported():
    """Round 5 threw every hipMemUnmap / hipMemRelease / hipMemAddressFree return away.  Now a release of something that is no
    block of ours comes back as an error with the pointer in it (tvdn_mem_free -> hipFree refuses), and the allocator's own
    calls count their failures (`faults`, `first_fault`) -- zero on a healthy run."""
    from cytvdn_amd import _lib
    L = _lib.lib()
    rc = L.tvdn_mem_free(C.c_void_p(0x7f0000001000))
    assert rc == -3 and "0x7f0000001000" in L.tvdn_last_error().decode()
    st = _lib.mem_status(0)
    assert st["faults"] == 0, st


def test_the_pool_honours_the_hbm_limit(monkeypatch):
    """Spare granules (up to 3 x the block) are drawn from what this process may take: never beyond TVDN_HBM_LIMIT, never into
    the last max(4 GiB, 5 %) of the device (pool_room)."""
    import torch
    from cytvdn_amd import _lib
    monkeypatch.setenv("TVDN_VMM_MIN_MIB", "1")
    monkeypatch.setenv("TVDN_GRANULE_MIB", "64")
    torch.cuda.empty_cache()
    monkeypatch.delenv("TVDN_HBM_LIMIT", raising=False)
    b = _lib.DeviceBlock(1 << 30, 0)                 # 16 granules; cleared memory: the pool reaches 3 x at once
    free_pool = _lib.mem_status(0)["last_pool"]
    b.free()
    monkeypatch.setenv("TVDN_HBM_LIMIT", "1536M")
    b = _lib.DeviceBlock(1 << 30, 0)
    st = _lib.mem_status(0)
    b.free()
    assert st["last_granules"] == 16 and 16 <= st["last_pool"] <= 24, st
    assert free_pool >= st["last_pool"]
    monkeypatch.setenv("TVDN_HBM_LIMIT", "256M")     # below the block itself: the block is what was asked for, the pool is nothing more
    b = _lib.DeviceBlock(1 << 30, 0)
    st = _lib.mem_status(0)
    b.free()
    assert st["last_granules"] == 16 and st["last_pool"] == 16, st


def test_a_peer_copy_out_of_a_granule_block_is_checked_before_a_device_list_runs(oracle, monkeypatch):
    """ABI 9: slabs of a device list sit on granules also when neighbours on other devices read them (every device of the list is
    granted access), and the run first proves that a peer copy out of such a block arrives intact (tvdn_run_stats.peer_check).
    One GPU here: the 'peers' are the same device (TVDN_PEER_CHECK=1 asks for the check anyway) -- the descriptor list, the
    check's copies and the fallback switch are what this exercises; two real devices: tools/first_node_run.sh."""
    from cytvdn_amd import _lib, synth
    monkeypatch.setenv("TVDN_VMM_MIN_MIB", "1")
    monkeypatch.setenv("TVDN_GRANULE_MIB", "2")
    monkeypatch.setenv("TVDN_PEER_CHECK", "1")
    shape, n_f = (24, 6, 16, 32), 5
    x = synth.cube(shape, seed=43, dtype=np.float32) + np.float32(0.25)
    mu = np.array([1.0, 0.8, 0.5, 0.6], np.float32)
    lam = mu / np.float32(32.0)
    ref = oracle.denoise(x, mu, n_f, True)
    b = _lib.DeviceBlock(8 << 20, 0, peers=[0, 0])             # the shared form of the allocation: owner listed among the peers
    assert b.kind == _lib.MEM_GRANULES
    b.free()
    for peer_env, want in (("1", 1), ("0", 0)):
        monkeypatch.setenv("TVDN_VMM_PEER", peer_env)
        a = _lib.RunArgs(dtype=0, ndim=4, bc_mode=2, device=0, n_fista=n_f, n_plain=0, n_devices=3)
        for i, v in enumerate(shape):
            a.shape[i] = v
        for q in range(4):
            a.clip[q], a.lambda_mu[q] = float((1.0 / lam)[q]), float((lam / mu).astype(np.float32)[q])
        recon, sums, st = np.empty_like(x), np.zeros((n_f, 3)), _lib.RunStats()
        a.data, a.recon_out, a.sums_out, a.stats = x.ctypes.data, recon.ctypes.data, sums.ctypes.data, C.addressof(st)
        _lib.check(_lib.lib().tvdn_run(C.byref(a)))
        assert bits_equal(recon, ref["recon"])
        assert st.peer_check == 1, (peer_env, st.peer_check)    # (slabs of ONE device are on granules either way: checked both times)
    assert _lib.mem_status(0)["faults"] == 0


@pytest.mark.parametrize("cycles", [12])
def test_the_sequence_of_round_5s_abort_in_a_loop(oracle, monkeypatch, cycles):
    """NOTES r5 "Open": one SIGABRT in nine full suites, inside the in-core denoise4D on a torch workspace that followed a streamed
    one and ended by RELEASING A KEPT GRANULE BLOCK of several GiB.  The same sequence, looped (tools/vmm_stress.py runs it 50 +
    200 times for profiles/r06_vmm_stress.txt): a multi-GiB granule block left as the kept state -> streamed call -> in-core call
    on a torch workspace (tvdn_run releases the kept block) -- the oracle's bits every time, no fault recorded."""
    from cytvdn_amd import _lib
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import vmm_stress
    monkeypatch.delenv("TVDN_WAVEFRONT", raising=False)
    monkeypatch.delenv("TVDN_STAGED", raising=False)
    res = vmm_stress.abort_sequence(oracle, cycles, log=None)
    assert res["mismatches"] == 0 and res["faults"] == 0, res
    res = vmm_stress.alloc_resize_release(oracle, 3 * cycles, log=None)
    assert res["mismatches"] == 0 and res["faults"] == 0, res
    _lib.lib().tvdn_release_cache()
    assert _lib.mem_status(0)["blocks"] == 0


_PIN_CHILD = r"""
import os, sys
import numpy as np
sys.path.insert(0, {root!r})
import torch
import cytvdn_amd as tv
from cytvdn_amd import synth
x = synth.cube((40, 8, 32, 64), seed=5, dtype=np.float32) + np.float32(0.25)       # 2.6 MB: the size of the abort
ref = synth.cube((40, 8, 32, 64), seed=6, dtype=np.float32)
mu = np.array([1.0, 1.0, 0.5, 0.5], np.float32)
tv.denoise4D(x, mu, 2, quiet=True)                                                # (first use: module load, lanes)
print("PRODUCT BEGIN", file=sys.stderr, flush=True)
tv.denoise4D(x, mu, [5, 3], FISTA=True, quiet=True)                               # tvdn_run, resident, torch workspace
tv.denoise4D(x, mu, 200, FISTA=True, quiet=True)                                  # traces of 200 x 3 doubles come home
tv.denoise4D(x, mu, 6, FISTA=True, reference_data=ref, quiet=True)                # MSE trace: the reference cube goes up too
os.environ["TVDN_LOOP"] = "python"
tv.denoise4D(x, mu, 6, FISTA=True, reference_data=ref, stopping_relative_change=1e-9, quiet=True)   # the loop in Python
os.environ.pop("TVDN_LOOP")
os.environ["TVDN_HBM_LIMIT"] = "24M"
tv.denoise4D(x, mu, [5, 3], FISTA=True, quiet=True)                               # streamed
tv.denoise3D(x[0].copy(), mu[:3], 5, quiet=True)
print("PRODUCT END", file=sys.stderr, flush=True)
t = torch.from_numpy(x).cuda()                                                    # the runtime's own path, for comparison
print("RUNTIME END", file=sys.stderr, flush=True)
"""


def test_the_library_never_lets_the_runtime_pin_the_callers_memory():
    """The abort of profiles/r06_abort_found.txt: for a copy from / to pageable memory above a few KiB the ROCm runtime pins the
    caller's pages IN PLACE and caches the pin by address and size; the cache outlives the memory, and a later copy from a new
    array on the same address faults on the GPU.  The library must never make that kind of call: under AMD_LOG_LEVEL=4 the
    runtime says "HSA Copy Using Pinned resource size N" whenever it does, and between the markers -- resident, streamed, MSE,
    Python-loop and 3-D calls on a 2.6 MB cube, the very size of the abort -- that line must not appear (torch's own copy
    after the markers shows that the check can see it)."""
    env = dict(os.environ)
    env["AMD_LOG_LEVEL"] = "4"
    p = subprocess.run([sys.executable, "-c", _PIN_CHILD.format(root=ROOT)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       errors="replace", timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    err = p.stderr
    a, b, c = err.index("PRODUCT BEGIN"), err.index("PRODUCT END"), err.index("RUNTIME END")
    inside = [l for l in err[a:b].splitlines() if "Copy Using Pinned resource" in l]
    after = [l for l in err[b:c].splitlines() if "Copy Using Pinned resource" in l]
    assert after, "the runtime no longer reports its pinned-in-place copies at this log level: the check above proves nothing"
    assert not inside, inside[:3]


def test_warm_up_is_idempotent_and_leaves_a_healthy_allocator():
    """cytvdn_amd.warm_up (tvdn.h tvdn_warm_up): the one-time set-up of a process's first call, on demand, on a helper thread or
    not, as often as one likes; afterwards the canary has run and passed and a denoise call gives the oracle's bits as ever."""
    import cytvdn_amd as tv
    from cytvdn_amd import _lib
    tv.warm_up(0)
    t = tv.warm_up(0, background=True)
    t.join(60)
    assert not t.is_alive()
    st = _lib.mem_status(0)
    assert st["canary"] == _lib.CANARY_PASSED and st["faults"] == 0
